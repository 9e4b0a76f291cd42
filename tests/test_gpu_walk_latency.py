"""The latency form of the coarse walk (kernels_hnsw_lat.hip: one workgroup per query on the fat graph, taken for
batches of at most 256 queries after ivfhnsw_gpu_prepare_latency) against the oracle: same ids in the same order with
bit-identical distances, ties included -- and identical to the throughput walk, which larger batches keep using."""
import numpy as np
import pytest

from conftest import corpus
import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _check(g, graph, queries, k, ef, chunk):
    for a in range(0, len(queries), chunk):
        qs = queries[a:a + chunk]
        ids, dist = g.coarse(qs, k, ef)
        for i, q in enumerate(qs):
            rid, rd = graph.search_knn(q, ef, k)
            n = len(rid)
            assert np.array_equal(ids[i, :n], rid), "query %d: ids differ\n%s\n%s" % (a + i, ids[i], rid)
            assert np.array_equal(dist[i, :n].view(np.uint32), rd.view(np.uint32)), "query %d: distances differ" % (a + i)
            assert (ids[i, n:] == 0xffffffff).all()


@pytest.mark.parametrize("k,ef", [(1, 1), (8, 8), (16, 40), (32, 80), (64, 100), (128, 130), (210, 210), (32, 256)])
def test_latency_walk_matches_oracle(gpu, k, ef):
    c = corpus(seed=31, nc=2048, d=128, M=16, n_base=20000, nq=96, efConstruction=100)
    g = gpu()
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.prepare_latency()
    _check(g, gr, c["queries"], k, ef, 1)       # one query per call
    _check(g, gr, c["queries"], k, ef, 32)      # small batches: a workgroup each


def test_latency_walk_d96_and_ties(gpu):
    c = corpus(seed=13, nc=128, d=96, M=16, n_base=10000, nq=64)
    g = gpu()
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.prepare_latency()
    _check(g, gr, c["queries"], 32, 64, 1)
    # exact distance ties: every centroid four times (the (dist, id) order of both heaps decides)
    rng = np.random.default_rng(77)
    base = synth.sift_like(rng, 256, 128)
    cents = np.concatenate([base, base, base, base])[rng.permutation(1024)]
    graph = orc.Hnsw.build(cents, M=8, efConstruction=64)
    q = (base[rng.choice(256, 64)] + rng.normal(0, 5, size=(64, 128))).astype(np.float32)
    q[:8] = base[:8]
    g2 = gpu()
    g2.upload_quantizer(graph.counts, graph.links, graph.vectors, graph.enterpoint)
    g2.prepare_latency()
    for k, ef in [(4, 4), (16, 16), (8, 30), (64, 64)]:
        _check(g2, graph, q, k, ef, 1)
        _check(g2, graph, q, k, ef, 64)


def test_search_per_call_with_latency_walk_equals_oracle(gpu):
    """The whole path one query per call (tests/test_ivfhnsw_sift1b.cpp:193-208) with the latency walk in front,
    IVFADC with OPQ and Grouping with pruning: labels and distance bits of the oracle."""
    for kw, pruning in ((dict(seed=21, nc=128, d=128, M=16, n_base=10000, nq=48, opq=True), False),
                        (dict(seed=41, nc=256, d=128, M=16, n_base=30000, nq=48, nsubc=16), True)):
        c = corpus(**kw)
        ox = synth.oracle_index(c)
        ox.set_params(8, 3000, 40, do_pruning=pruning)
        ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
        g = gpu()
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                     c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
        gr = c["graph"]
        g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
        if c["nsubc"]:
            g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                              c["inter_centroid_dists"])
        g.prepare_latency()
        for i, x in enumerate(c["queries"]):
            d1, l1 = g.search(x, 1, 8, 3000, efSearch=40, do_pruning=pruning)
            assert l1[0, 0] == ref_l[i, 0] and d1.view(np.uint32)[0, 0] == ref_d.view(np.uint32)[i, 0]


def test_prepare_latency_refuses_shapes_it_cannot_take(gpu, pkg):
    rng = np.random.default_rng(3)
    cents = synth.sift_like(rng, 64, 64)            # d = 64: not a shape of the latency form
    graph = orc.Hnsw.build(cents, M=4, efConstruction=20)
    g = gpu()
    g.upload_quantizer(graph.counts, graph.links, graph.vectors, graph.enterpoint)
    with pytest.raises(pkg.IvfHnswError) as e:
        g.prepare_latency()
    assert e.value.code == pkg.ERR_INVALID
    ids, _ = g.coarse(cents[:4], 4, 8)              # the throughput walk still serves the handle
    assert (ids[:, 0] == np.arange(4)).all()
