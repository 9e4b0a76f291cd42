"""ivfhnsw_gpu_build_graph: hnswlib's addPoint loop (hnswlib/hnswalg.cpp:212-225) for all nodes at once, candidates = the
exact nearest earlier nodes from the MFMA neighbour-table kernel.  Checked against the oracle's SERIAL restatement of the
same contract (orc_hnsw_build_exact: the literal insertion loop over the reference's connect step) link for link, and for
what the graph is for: on CLUSTERED centroids -- where a plain k-NN graph falls apart into islands -- the walk from the
fixed enter point still finds the true nearest centroid."""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def clustered(rng, n, d, per=64, spread=18.0):
    """n rows in tight clusters of `per` around SIFT-like cluster centres (k-means centroids of clustered data look so)."""
    centres = synth.sift_like(rng, (n + per - 1) // per, d)
    x = centres[rng.integers(0, len(centres), n)] + rng.normal(0, spread, (n, d))
    return x.astype(np.float32)


@pytest.mark.parametrize("d,k", [(128, 16), (96, 64), (32, 5)])
def test_earlier_rows_table_matches_oracle(gpu, d, k):
    rng = np.random.default_rng(d + k)
    x = rng.normal(0, 1, (1500, d)).astype(np.float32)
    x[700:720] = x[10]
    g = gpu()
    ids, dist = g.knn(x, k, mode=g.KNN_EARLIER)
    rid, rdist = orc.knn(x, k, mode=2)
    assert np.array_equal(ids, rid) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    assert (ids[0] == 0xffffffff).all() and ids[1, 0] == 0 and (ids[1, 1:] == 0xffffffff).all()
    valid = ids != 0xffffffff
    assert (ids[valid] < np.nonzero(valid)[0]).all()


@pytest.mark.parametrize("kind,n,d,M,maxM,ncand", [("iid", 3000, 128, 16, 32, 64), ("clustered", 4000, 128, 16, 32, 64),
                                                    ("clustered", 2500, 96, 8, 16, 20), ("dups", 1200, 64, 16, 32, 48),
                                                    ("iid", 10, 128, 16, 32, 64), ("iid", 1, 128, 16, 32, 64)])
def test_graph_equals_the_serial_insertion_loop(gpu, kind, n, d, M, maxM, ncand):
    rng = np.random.default_rng(n + d)
    if kind == "clustered":
        x = clustered(rng, n, d)
    else:
        x = synth.sift_like(rng, n, d) if d % 16 == 0 and kind != "dups" else rng.normal(0, 1, (n, d)).astype(np.float32)
    if kind == "dups":
        x[100:180] = x[3]      # 81 identical rows: every tie rule of the heuristic and the shrink is exercised
    counts, links = gpu().build_graph(x, M, maxM, ncand)
    ref = orc.Hnsw.build_exact(x, M, maxM, ncand)
    assert np.array_equal(counts, ref.counts)
    # slots at and beyond a node's count are never read (the serial loop leaves stale ids there after a shrink,
    # hnswalg.cpp:199-207; the edges file holds only `count` entries, :252-265)
    live = np.arange(maxM)[None, :] < counts[:, None]
    assert np.array_equal(np.where(live, links, 0), np.where(live, ref.links.reshape(n, maxM), 0))
    ref.free()
    if n > 100:
        assert counts[1:].min() >= 1 and counts.max() <= maxM


def test_walk_on_clustered_centroids_reaches_the_true_nearest(gpu):
    """What the construction is for.  20 000 centroids in clusters of 64: a plain 16-NN graph with reverse links keeps every
    link inside a node's own cluster; the insertion-ordered graph keeps the long links early nodes made."""
    rng = np.random.default_rng(77)
    n, d = 20000, 128
    x = clustered(rng, n, d)
    q = (x[rng.integers(0, n, 2000)] + rng.normal(0, 6.0, (2000, d))).astype(np.float32)
    g = gpu()
    gt, _ = g.knn(x, 1, q)
    counts, links = g.build_graph(x, 16, 32, 64)
    g.upload_quantizer(counts, links, x, 0)
    ids, _ = g.coarse(q, 1, 80)
    recall = float((ids[:, 0] == gt[:, 0]).mean())
    kc, kl = synth.knn_graph(x, 16, 32)
    g2 = gpu()
    g2.upload_quantizer(kc, kl, x, 0)
    ids2, _ = g2.coarse(q, 1, 80)
    recall_knn = float((ids2[:, 0] == gt[:, 0]).mean())
    print("\n[graph] clustered centroids, efSearch 80: walk finds the true nearest centroid for %.3f of the queries on the "
          "insertion-ordered graph, %.3f on the plain k-NN graph" % (recall, recall_knn))
    assert recall >= 0.95
    # and the device walk on it equals the oracle's walk, as on any graph
    og = orc.Hnsw.from_arrays(counts, links, x, 16, 0)
    for i in range(0, 2000, 100):
        found, _ = og.search_knn(q[i], 80, 1)
        assert found[0] == ids[i, 0]
    og.free()
