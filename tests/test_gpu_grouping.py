"""GPU parity: IndexIVF_HNSW_Grouping::search (IndexIVF_HNSW_Grouping.cpp:188-363) through the C ABI vs the
CPU oracle, with and without pruning, with and without OPQ: identical labels, bit-identical distances,
identical number of scored codes / sub-groups (the reference's `ncode`, :334).
"""
import numpy as np
import pytest

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu


def _gpu_index(gpu, c):
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    return g


CASES = [
    # corpus, nprobe, max_codes, efSearch
    (dict(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16), 16, 800, 40),     # max_codes bites in both passes
    (dict(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16), 16, 10 ** 9, 40),
    (dict(seed=42, nc=128, d=128, M=16, n_base=20000, nq=64, nsubc=64), 8, 600, 80),      # nsubc = one wavefront
    (dict(seed=43, nc=256, d=96, M=16, n_base=20000, nq=64, nsubc=8), 32, 1500, 64),      # DEEP shape
    (dict(seed=44, nc=256, d=128, M=8, n_base=20000, nq=64, nsubc=80, efConstruction=120), 4, 10 ** 9, 100),  # nsubc > 64
    (dict(seed=45, nc=128, d=128, M=16, n_base=12000, nq=64, nsubc=32, opq=True), 16, 700, 48),
]


@pytest.mark.parametrize("do_pruning", [False, True])
@pytest.mark.parametrize("kw,nprobe,max_codes,ef", CASES)
def test_grouping_top1_matches_oracle(gpu, kw, nprobe, max_codes, ef, do_pruning):
    c = corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=do_pruning)
    ref_d, ref_l, cid, cd, st = ox.search_batch(c["queries"], k=1)
    g = _gpu_index(gpu, c)
    # host-supplied coarse stage
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, coarse_ids=cid, coarse_dists=cd, do_pruning=do_pruning)
    assert np.array_equal(lab, ref_l), "labels differ at %s" % np.nonzero((lab != ref_l).any(1))[0][:10]
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
    assert g.last_scan_counts() == (st.ncode, st.nseg)
    # whole path on the device
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef, do_pruning=do_pruning)
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))


def test_pruning_changes_the_scanned_set(gpu):
    """Sanity of the test itself: with pruning on, fewer sub-groups are scored, so the two modes really
    exercise different plans."""
    c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    g = _gpu_index(gpu, c)
    g.search(c["queries"], 1, 16, 10 ** 9, efSearch=40, do_pruning=False)
    full = g.last_scan_counts()
    g.search(c["queries"], 1, 16, 10 ** 9, efSearch=40, do_pruning=True)
    pruned = g.last_scan_counts()
    assert pruned[0] < full[0] and pruned[1] < full[1]


def test_grouping_requires_quantizer(gpu, pkg):
    c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    with pytest.raises(pkg.IvfHnswError) as e:
        g.search(c["queries"][:2], 1, 4, 100, coarse_ids=np.zeros((2, 4), np.uint32),
                 coarse_dists=np.zeros((2, 4), np.float32))
    assert e.value.code == pkg.ERR_STATE


BIG_CASES = [
    # larger batches (>= 1024 queries), more probes than a wavefront has lanes, nsubc > 64:
    # corpus, nprobe, max_codes, efSearch
    (dict(seed=46, nc=256, d=128, M=16, n_base=30000, nq=1500, nsubc=16), 16, 800, 40),
    (dict(seed=47, nc=128, d=96, M=16, n_base=20000, nq=1100, nsubc=80, opq=True, efConstruction=120), 24, 2500, 64),
    (dict(seed=48, nc=256, d=128, M=8, n_base=20000, nq=1300, nsubc=8), 200, 10 ** 9, 220),  # more probes than a wavefront
]


@pytest.mark.parametrize("do_pruning", [False, True])
@pytest.mark.parametrize("kw,nprobe,max_codes,ef", BIG_CASES)
def test_grouping_large_batches_match_oracle(gpu, kw, nprobe, max_codes, ef, do_pruning):
    """Same contract as above on batches of more than a thousand queries, with up to 200 probes per query."""
    c = corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=do_pruning)
    ref_d, ref_l, _, _, st = ox.search_batch(c["queries"], k=1)
    g = _gpu_index(gpu, c)
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef, do_pruning=do_pruning)
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
    assert g.last_scan_counts()[0] == st.ncode
