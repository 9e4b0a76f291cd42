"""GPU parity: the on-device HNSW walk (hnswlib/hnswalg.cpp:48-109,227-234 + the unload loop of
IndexIVF_HNSW.cpp:249-259) vs the CPU oracle: same centroid ids in the same order with bit-identical
distances, for every query.  An approximate walk has to be reproduced decision by decision; a
brute-force coarse search would not be parity.
"""
import numpy as np
import pytest

from conftest import corpus
import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _walk_both(g, graph, queries, k, ef):
    g.upload_quantizer(graph.counts, graph.links, graph.vectors, graph.enterpoint)
    ids, dist = g.coarse(queries, k, ef)
    for i, q in enumerate(queries):
        rid, rd = graph.search_knn(q, ef, k)
        n = len(rid)
        assert np.array_equal(ids[i, :n], rid), "query %d: ids differ\n%s\n%s" % (i, ids[i], rid)
        assert np.array_equal(dist[i, :n].view(np.uint32), rd.view(np.uint32)), "query %d: distances differ" % i
        assert (ids[i, n:] == 0xffffffff).all()


@pytest.mark.parametrize("k,ef", [(1, 1), (8, 8), (16, 40), (32, 80), (64, 100), (128, 130), (210, 210), (40, 300)])
def test_walk_matches_oracle(gpu, k, ef):
    c = corpus(seed=31, nc=2048, d=128, M=16, n_base=20000, nq=96, efConstruction=100)
    _walk_both(gpu(), c["graph"], c["queries"], k, ef)


def test_walk_d96(gpu):
    c = corpus(seed=13, nc=128, d=96, M=16, n_base=10000, nq=64)
    _walk_both(gpu(), c["graph"], c["queries"], 32, 64)


def test_walk_with_duplicate_centroids(gpu):
    """Exact distance ties: every centroid appears four times, so (dist, id) ordering of both heaps
    (pair<float,idx_t>, hnswalg.cpp:53-54) decides which copies are returned."""
    rng = np.random.default_rng(77)
    base = synth.sift_like(rng, 256, 128)
    cents = np.concatenate([base, base, base, base])[rng.permutation(1024)]
    graph = orc.Hnsw.build(cents, M=8, efConstruction=64)
    q = (base[rng.choice(256, 64)] + rng.normal(0, 5, size=(64, 128))).astype(np.float32)
    q[:8] = base[:8]  # and queries that coincide with centroids (distance 0)
    for k, ef in [(4, 4), (16, 16), (8, 30), (64, 64)]:
        _walk_both(gpu(), graph, q, k, ef)


def test_tiny_graph_fewer_nodes_than_nprobe(gpu):
    """The reference pops an empty queue here (IndexIVF_HNSW.cpp:249-258, undefined); the oracle and the
    device both return what was found and pad."""
    rng = np.random.default_rng(5)
    cents = synth.sift_like(rng, 5, 128)
    graph = orc.Hnsw.build(cents, M=4, efConstruction=10)
    q = synth.sift_like(rng, 4, 128)
    _walk_both(gpu(), graph, q, 8, 16)


def test_search_with_device_coarse_equals_host_coarse(gpu):
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    ox = synth.oracle_index(c)
    ox.set_params(16, 3000, 40)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    dist, lab = g.search(c["queries"], 1, 16, 3000, efSearch=40)
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))


def test_opq_search_with_device_coarse(gpu):
    """OPQ: the query is rotated on the device, then walks the rotated graph (IndexIVF_HNSW.cpp:240,248)."""
    c = corpus(seed=21, nc=128, d=128, M=16, n_base=10000, nq=64, opq=True)
    ox = synth.oracle_index(c)
    ox.set_params(8, 4000, 40)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    dist, lab = g.search(c["queries"], 1, 8, 4000, efSearch=40)
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
