"""More exact distance ties at the efSearch boundary than the walk's LDS tail holds (kTailCap = 64, walk_set.h).

The reference has no limit there: a candidate evicted from topResults stays in candidateSet and is still popped while its
distance EQUALS the lower bound (hnswlib/hnswalg.cpp:67-68,93).  Round 2 gave up at 64 with an error status; since round 3
the tail moves into a per-wavefront global bitmap (TailSpill) and the walk goes on exactly as the reference's does.

The graph below forces it.  All vectors lie on one axis, the query at 0:
  hubs H0..H5  nearest (distance ~4), chained, each with 31 further links: expanded first, one after the other
  93 nodes A   all at the SAME distance 9: the first 76 fill the result set (efSearch 80), the rest are refused (9 < 9 fails)
  70 nodes B   distinct distances in (4.4, 8.5): every admission evicts the A of largest id, unexpanded, while other A's
               keep the maximum at 9 -> the tail grows to 72 entries
  node C       the nearest of all, linked ONLY from the A of smallest id -- which is evicted into the tail and popped
               last (largest id first).  A walk that drops the tail never finds C.
"""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu

D, MAXM, EF = 96, 32, 80   # d = 96: the latency walk (d = 128 or 96 only) takes the graph too


def build(n_b):
    """(counts, links, vectors, enterpoint): ids 0..5 hubs, then C, then A's, then B's."""
    hubs = list(range(6))
    c_id = 6
    a_ids = list(range(7, 7 + 93))
    b_ids = list(range(100, 100 + n_b))
    n = 100 + n_b
    x = np.zeros(n, np.float64)
    for j in hubs:
        x[j] = 2.0 - 0.01 * j
    x[c_id] = 0.5
    x[a_ids] = 3.0
    x[b_ids] = np.linspace(2.1, 2.9, n_b)
    vec = np.zeros((n, D), np.float32)
    vec[:, 0] = x
    links = np.zeros((n, MAXM), np.uint32)
    counts = np.zeros(n, np.uint8)

    def set_links(node, lst):
        assert len(lst) <= MAXM
        links[node, :len(lst)] = lst
        counts[node] = len(lst)

    # the A's are offered in DESCENDING id order, so that the 76 that enter the set are the large ids and the A of
    # smallest id ... is refused?  No: it must be IN the set to be evicted into the tail -- offer it first.
    a_order = [a_ids[0]] + a_ids[:0:-1]
    set_links(0, [1] + a_order[0:31])
    set_links(1, [2] + a_order[31:62])
    set_links(2, [3] + a_order[62:93])
    set_links(3, [4] + b_ids[0:31])
    set_links(4, [5] + b_ids[31:62])
    set_links(5, b_ids[62:n_b][:32])
    set_links(a_ids[0], [c_id])          # the only way to C
    set_links(c_id, [a_ids[0]])
    return counts, links, vec, 0


def queries(nq, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros((nq, D), np.float32)
    q[:, 1:] = rng.normal(0, 0.05, (nq, D - 1))   # off-axis components shift every distance by the same amount
    q[0, 1:] = 0
    return q


@pytest.mark.parametrize("n_b,finds_c", [(70, True), (95, False)])
def test_tail_beyond_its_lds_capacity_matches_the_reference(gpu, n_b, finds_c):
    """n_b 70: 72 tail entries stay alive and are popped one by one, the last one leads to C.  n_b 95: every A is
    evicted, the maximum drops below 9 while the tail holds > 64 entries -- it dies at once (hnswalg.cpp:67)."""
    counts, links, vec, ep = build(n_b)
    og = orc.Hnsw.from_arrays(counts, links, vec, 16, ep)
    k = 16
    g = gpu()
    g.upload_quantizer(counts, links, vec, ep)
    for nq in (300, 3):      # the throughput walk; after prepare_latency the workgroup-per-query walk (<= 256 queries)
        q = queries(nq, nq)
        ids, dist = g.coarse(q, k, EF)
        for i in range(nq):
            rid, rd = og.search_knn(q[i], EF, k)
            assert np.array_equal(ids[i, :len(rid)], rid), (nq, i)
            assert np.array_equal(dist[i, :len(rid)].view(np.uint32), np.asarray(rd, np.float32).view(np.uint32))
            assert (6 in rid) == finds_c
        if nq == 300:
            g.prepare_latency()
    # and again on the same handle: the bitmaps were handed back zero
    q = queries(300, 9)
    ids2, _ = g.coarse(q, k, EF)
    for i in range(0, 300, 37):
        rid, _ = og.search_knn(q[i], EF, k)
        assert np.array_equal(ids2[i, :len(rid)], rid)
    og.free()


def test_tail_spill_inside_a_full_search_and_a_split_batch(gpu):
    """The same graph as the coarse quantizer of an index: one-part, two-part (>= 8192 queries) and host-pointer calls
    report no error and agree with the oracle."""
    counts, links, vec, ep = build(70)
    n = len(counts)
    rng = np.random.default_rng(3)
    sizes = rng.integers(1, 6, n)
    offsets = np.zeros(n + 1, np.uint64)
    offsets[1:] = np.cumsum(sizes)
    tot = int(offsets[-1])
    M = 4
    codes = rng.integers(0, 256, (tot, M)).astype(np.uint8)
    ncodes = rng.integers(0, 256, tot).astype(np.uint8)
    ids = np.arange(tot, dtype=np.uint32)
    pq = rng.normal(0, 0.2, (M, 256, D // M)).astype(np.float32)
    ntab = np.sort(rng.normal(9, 1, 256)).astype(np.float32)
    cn = (vec.astype(np.float64) ** 2).sum(1).astype(np.float32)
    g = gpu()
    g.upload_ivf(D, M, offsets, ids, codes, ncodes, cn, pq, ntab)
    g.upload_quantizer(counts, links, vec, ep)
    og = orc.Hnsw.from_arrays(counts, links, vec, 16, ep)
    ox = orc.Index(D, M, og, pq, ntab, offsets, ids, codes, ncodes, cn)
    ox.set_params(16, 10 ** 9, EF)
    q = queries(9000, 5)
    rd, rl, _, _, _ = ox.search_batch(q, 1, 8)
    for split in (780, 0):
        g.set_batch_split(split)
        dist, lab = g.search(q, 1, 16, 10 ** 9, efSearch=EF)
        assert np.array_equal(lab, rl) and np.array_equal(dist.view(np.uint32), rd.view(np.uint32))
    og.free()
