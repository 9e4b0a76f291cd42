"""GPU parity at the edges of what the device path supports: efSearch 1024 with nprobe 1000 (sixteen registers of
result set per lane), k = 1024, a list longer than the reference's 65536-entry norm scratch
(IndexIVF_HNSW.cpp:17, a6 in SURVEY.md 8 -- the reference overflows there, the oracle and the device do not),
every list but one empty, a single query, a single centroid.
"""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _upload(g, c):
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)


def _check(g, c, k, nprobe, max_codes, ef, heap=True):
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, _, _, st = ox.search_batch(c["queries"], k=k)
    dist, lab = g.search(c["queries"], k, nprobe, max_codes, efSearch=ef, heap_order=heap)
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
    assert g.last_scan_counts()[0] == st.ncode


def test_widest_walk_and_most_probes(gpu):
    c = synth.make_corpus(seed=301, nc=2500, d=32, M=4, n_base=30000, nq=24, efConstruction=60)
    g = gpu()
    _upload(g, c)
    _check(g, c, 1, 1000, 10 ** 9, 1024)
    _check(g, c, 1, 700, 9000, 1000)
    ids, dist = g.coarse(c["queries"], 1000, 1024)          # the coarse stage alone at full width
    for i, q in enumerate(c["queries"]):
        rid, rd = c["graph"].search_knn(q, 1024, 1000)
        assert np.array_equal(ids[i, :len(rid)], rid) and np.array_equal(dist[i, :len(rid)].view(np.uint32), rd.view(np.uint32))


def test_largest_k(gpu):
    c = synth.make_corpus(seed=302, nc=64, d=32, M=4, n_base=20000, nq=12, efConstruction=40)
    g = gpu()
    _upload(g, c)
    _check(g, c, 1024, 16, 10 ** 9, 32)


def test_list_longer_than_the_reference_scratch(gpu):
    """One list of 70 000 codes (the reference's `norms` scratch holds 65536)."""
    c = synth.make_corpus(seed=303, nc=16, d=32, M=4, n_base=90000, nq=8, efConstruction=20, empty_frac=0.0)
    sizes = np.diff(c["offsets"].astype(np.int64))
    if sizes.max() <= 65536:    # force it: move everything into list 0 in file order
        n = int(c["offsets"][-1])
        off = np.zeros_like(c["offsets"])
        off[1:] = n
        c = dict(c, offsets=off)
    assert np.diff(c["offsets"].astype(np.int64)).max() > 65536
    g = gpu()
    _upload(g, c)
    _check(g, c, 1, 4, 10 ** 9, 8)
    _check(g, c, 10, 4, 10 ** 9, 8)


def test_single_centroid_and_single_query(gpu):
    rng = np.random.default_rng(304)
    c = synth.make_corpus(seed=304, nc=1, d=16, M=4, n_base=500, nq=1, efConstruction=10, empty_frac=0.0)
    g = gpu()
    _upload(g, c)
    _check(g, c, 1, 1, 10 ** 9, 1)
    _check(g, c, 5, 1, 100, 1)
