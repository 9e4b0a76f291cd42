"""The device walk on a graph of the reference's size (993 127 centroids, examples/run_sift1b.sh): ids need the
10-bit visited tags, the filter's survivors enter the visited set late (FMODE 3), the neighbour rows (4 GB) and the
vectors no longer fit the Infinity Cache.  Bar as everywhere: ids and distance bits of the CPU oracle's walk."""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def test_walk_at_993127_nodes_matches_oracle(gpu):
    nc, nq, ef, k = 993127, 1500, 80, 32
    rng = np.random.default_rng(77)
    cents = synth.sift_like(rng, nc, 128)
    counts, links = synth.knn_graph_torch(cents, 16, 32)
    q = (cents[rng.choice(nc, nq)] + rng.normal(0, 12.0, size=(nq, 128))).astype(np.float32)
    g = gpu()
    g.upload_quantizer(counts, links, cents, 0)
    ids, dist = g.coarse(q, k, ef)
    graph = orc.Hnsw.from_arrays(counts, links, cents, 16, 0)
    try:
        for i in range(nq):
            rid, rd = graph.search_knn(q[i], ef, k)
            n = len(rid)
            assert np.array_equal(ids[i, :n], rid), "query %d: ids differ" % i
            assert np.array_equal(dist[i, :n].view(np.uint32), rd.view(np.uint32)), "query %d: distances differ" % i
    finally:
        graph.free()
