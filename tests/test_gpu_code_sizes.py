"""Every code size the reference accepts: `pq_L2sqr` steps four bytes at a time (IndexIVF_HNSW.cpp:802-814), so
code_size is any multiple of 4.  4 / 8 / 16 / 32 have scan kernels of their own; everything else runs the run-time
form (table in dynamic LDS, same m order of the sum).  Labels and distance bits of the oracle for k = 1, k > 1
(ascending and faiss heap order), Grouping + pruning, one query per call, and the construction side's code bytes."""
import numpy as np
import pytest

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu

CASES = [
    dict(seed=81, nc=128, d=96, M=12, n_base=9000, nq=48, efConstruction=60),    # dsub 8
    dict(seed=82, nc=128, d=96, M=24, n_base=9000, nq=48, efConstruction=60),    # dsub 4
    dict(seed=83, nc=96, d=160, M=20, n_base=6000, nq=32, efConstruction=60),    # dsub 8
    dict(seed=84, nc=96, d=128, M=64, n_base=6000, nq=32, efConstruction=60),    # dsub 2, 64 KB table
    dict(seed=85, nc=64, d=112, M=28, n_base=4000, nq=32, efConstruction=60, opq=True),
    dict(seed=86, nc=64, d=128, M=128, n_base=3000, nq=16, efConstruction=60),   # dsub 1: the LDS limit (128 KB)
]


def _upload(g, c):
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)


@pytest.mark.parametrize("kw", CASES, ids=lambda kw: "M%d" % kw["M"])
def test_any_multiple_of_four(gpu, kw):
    c = corpus(**kw)
    nprobe, max_codes, ef = 12, 1500, 32
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, _, _, st = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef)
    assert np.array_equal(lab, ref_l) and np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
    assert g.last_scan_counts()[0] == st.ncode
    for i in range(8):   # one query per call: the split scan with atomicMin
        d1, l1 = g.search(c["queries"][i], 1, nprobe, max_codes, efSearch=ef)
        assert l1[0, 0] == ref_l[i, 0] and d1[0, 0] == ref_d[i, 0]
    # k > 1: the heap array faiss leaves, and the same set ascending
    ref_dk, ref_lk, _, _, _ = ox.search_batch(c["queries"], k=7)
    dk, lk = g.search(c["queries"], 7, nprobe, max_codes, efSearch=ef, heap_order=True)
    assert np.array_equal(lk, ref_lk) and np.array_equal(dk.view(np.uint32), ref_dk.view(np.uint32))
    da, la = g.search(c["queries"], 7, nprobe, max_codes, efSearch=ef)
    assert (np.diff(da, axis=1) >= 0).all()
    assert np.array_equal(np.sort(la, axis=1), np.sort(ref_lk, axis=1))


def test_grouping_with_code_size_12(gpu):
    c = corpus(seed=87, nc=128, d=96, M=12, n_base=9000, nq=48, efConstruction=60, nsubc=8, opq=True)
    ox = synth.oracle_index(c)
    g = gpu()
    _upload(g, c)
    g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    for pruning in (False, True):
        ox.set_params(10, 1200, 40, do_pruning=pruning)
        ref_d, ref_l, _, _, st = ox.search_batch(c["queries"], k=1)
        dist, lab = g.search(c["queries"], 1, 10, 1200, efSearch=40, do_pruning=pruning)
        assert np.array_equal(lab, ref_l) and np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
        assert g.last_scan_counts()[0] == st.ncode


@pytest.mark.parametrize("d,M", [(96, 12), (96, 24), (160, 20)])
def test_encode_with_other_code_sizes(gpu, d, M):
    e = synth.make_encode_case(seed=90 + M, nc=96, d=d, M=M, opq=False, n=3000)
    g = gpu()
    gr = e["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(d, M, e["cb"], e["nt"])
    idx, codes, ncodes = g.encode(e["x"], efSearch=40)
    o_idx, o_codes, o_nc, _ = e["ox"].add_batch_encode(e["x"])
    assert np.array_equal(idx, o_idx) and np.array_equal(codes, o_codes) and np.array_equal(ncodes, o_nc)
