"""ivfhnsw_gpu_knn (kernels_knn.hip): exact k-nearest-neighbour tables on the matrix cores against the oracle's
restatement of the same contract (orc_knn: fmaf-chain norms and dot products, dist = (nq + nx) - 2 dot, the k smallest
(dist, id)) -- ids AND distance bits.  Shapes: neighbour tables (a row's own entry left out) and query-vs-base tables,
d = 128 / 96 / 64 / 32 / 20, k on both sides of every buffer size the kernel instantiates, duplicate rows (exact ties:
the smaller id first), fewer rows than k, column splits (few queries against many rows), row counts that are not
multiples of the 128-row block or the 32-column tile."""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _check(g, base, k, queries=None):
    ids, dist = g.knn(base, k, queries)
    rid, rdist = orc.knn(base, k, queries)
    assert np.array_equal(ids, rid)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    return ids, dist


@pytest.mark.parametrize("d,k", [(128, 16), (128, 1), (96, 16), (96, 33), (64, 32), (32, 64), (20, 17), (128, 80)])
def test_neighbour_table_matches_oracle(gpu, d, k):
    rng = np.random.default_rng(d * 100 + k)
    n = 3000 + 37
    x = synth.sift_like(rng, n, d) if d in (128, 64) else rng.normal(0, 1, (n, d)).astype(np.float32)
    x[100:140] = x[50]          # 41 identical rows: exact ties, smaller id first
    ids, dist = _check(gpu(), x, k)
    assert (ids != np.arange(n)[:, None]).all()
    assert (np.diff(dist.astype(np.float64), axis=1) >= 0).all()


@pytest.mark.parametrize("nq,nx,d,k", [(1, 70000, 128, 10), (257, 50000, 96, 16), (1000, 4096, 128, 40), (33, 31, 128, 16),
                                        (5, 3, 64, 8), (130, 129, 32, 80)])
def test_query_tables_and_column_splits(gpu, nq, nx, d, k):
    rng = np.random.default_rng(nq + nx)
    x = rng.normal(0, 1, (nx, d)).astype(np.float32)
    q = (x[rng.integers(0, nx, nq)] + rng.normal(0, 0.3, (nq, d))).astype(np.float32)
    q[0] = x[nx // 2]                      # an exact hit: distance rounds to (about) zero, possibly negative
    ids, dist = _check(gpu(), x, k, q)
    if nx < k:
        assert (ids[:, nx:] == 0xffffffff).all() and (dist[:, nx:] == np.finfo(np.float32).max).all()


def test_knn_graph_helper_is_native_and_symmetrised(gpu):
    """synth.knn_graph (the bench's million-node graphs) = the library's table + reverse links on the host."""
    rng = np.random.default_rng(5)
    x = synth.sift_like(rng, 5000, 128)
    counts, links = synth.knn_graph(x, 16, 32)
    rid, _ = orc.knn(x, 16)
    assert np.array_equal(links[:, :16], rid)
    assert counts.min() >= 16 and counts.max() <= 32
    # every forward edge has its reverse unless the target was full
    for s in range(0, 5000, 97):
        for t in rid[s]:
            assert s in links[t][:counts[t]] or counts[t] == 32


def test_knn_refuses_what_it_cannot_do(gpu, pkg):
    g = gpu()
    x = np.zeros((10, 130), np.float32)
    with pytest.raises(pkg.IvfHnswError):
        g.knn(x, 4)                        # d not a multiple of 4 / above 128
    with pytest.raises(pkg.IvfHnswError):
        g.knn(np.zeros((10, 128), np.float32), 81)
