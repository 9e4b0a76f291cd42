"""CPU tests of the oracle itself (no GPU): known-answer checks of every leaf against an independent
numpy float32 evaluation of the order the oracle documents, heap semantics against a brute-force model,
HNSW walk against an independent pure-Python restatement with heapq, and the on-disk formats.

PARITY UNPINNED: the reference ships no golden vectors for this path, so these tests pin
the oracle to its own documented contract and to hand-computable cases, not to reference outputs.
"""
import heapq
import os
import struct

import numpy as np
import pytest

import synth
from oracle import orc

F = np.float32


def np_l2_ref(x, y):
    lane = np.zeros(8, F)
    for b in range(len(x) // 16):
        for h in range(2):
            xs, ys = x[b * 16 + h * 8:][:8], y[b * 16 + h * 8:][:8]
            diff = (xs - ys).astype(F)
            lane = (lane + (diff * diff).astype(F)).astype(F)
    r = F(lane[0])
    for i in range(1, 8):
        r = F(r + lane[i])
    return r


def np_ip_sse(x, y):
    s = np.zeros(4, F)
    n = len(x)
    i = 0
    while i + 4 <= n:
        s = (s + (x[i:i + 4] * y[i:i + 4]).astype(F)).astype(F)
        i += 4
    for l in range(n - i):
        s[l] = F(s[l] + F(x[i + l] * y[i + l]))
    return F(F(s[0] + s[1]) + F(s[2] + s[3]))


@pytest.mark.parametrize("d", [16, 96, 128, 100])
def test_l2sqr_order(d):
    rng = np.random.default_rng(d)
    for _ in range(50):
        x = rng.normal(0, 50, d).astype(F)
        y = rng.normal(0, 50, d).astype(F)
        assert F(orc.l2sqr(x, y)) == np_l2_ref(x, y)


def test_l2sqr_ignores_tail_beyond_multiple_of_16():
    """hnswalg.cpp:330: qty16 = d >> 4, the remaining d % 16 dims never enter the sum."""
    x = np.arange(20, dtype=F)
    y = np.zeros(20, F)
    assert orc.l2sqr(x, y) == float((np.arange(16.0) ** 2).sum())


def test_l2sqr_known_answer():
    x = np.zeros(16, F)
    y = np.zeros(16, F)
    x[3] = 3.0
    y[11] = 4.0
    assert orc.l2sqr(x, y) == 25.0


@pytest.mark.parametrize("d,M", [(128, 16), (128, 8), (96, 16), (64, 4), (96, 8)])
def test_inner_prod_table_order(d, M):
    rng = np.random.default_rng(d * M)
    dsub = d // M
    cb = rng.normal(0, 10, (M, 256, dsub)).astype(F)
    x = rng.normal(0, 30, d).astype(F)
    tab = orc.inner_prod_table(x, cb, M)
    for m in range(M):
        for c in (0, 1, 17, 255):
            assert tab[m, c] == np_ip_sse(x[m * dsub:(m + 1) * dsub], cb[m, c])


def test_inner_prod_table_known_answer():
    d, M = 16, 4
    cb = np.zeros((M, 256, 4), F)
    cb[2, 7] = [1, 2, 3, 4]
    x = np.arange(16, dtype=F)
    tab = orc.inner_prod_table(x, cb, M)
    assert tab[2, 7] == 8 * 1 + 9 * 2 + 10 * 3 + 11 * 4
    assert np.count_nonzero(tab) == 1


def test_opq_apply_is_fma_chain():
    import math
    rng = np.random.default_rng(3)
    d = 32
    A = rng.normal(size=(d, d)).astype(F)
    x = rng.normal(size=d).astype(F)
    y = orc.opq_apply(A, x)
    for i in range(d):
        acc = 0.0
        for k in range(d):  # fma in double then round == fmaf for these magnitudes? no: emulate exactly
            acc = float(F(np.float64(A[i, k]) * np.float64(x[k]) + np.float64(acc)))
        # product of two f32 is exact in f64; the sum with an f32 accumulator rounds once in f64 then once
        # to f32 -- double rounding can differ from fmaf by 1 ulp in rare cases, so allow it
        assert abs(float(y[i]) - acc) <= abs(acc) * 2 ** -23
    # identity and permutation matrices are exact
    assert np.array_equal(orc.opq_apply(np.eye(d, dtype=F), x), x)
    P = np.eye(d, dtype=F)[rng.permutation(d)]
    assert np.array_equal(orc.opq_apply(P, x), P @ x)


def _heap_model_run(k, seq):
    """What any correct max-heap of size k does with the admit rule of IndexIVF_HNSW.cpp:285-288."""
    L = orc.lib()
    val = np.empty(k, F)
    ids = np.empty(k, np.int64)
    L.orc_maxheap_heapify(k, val.ctypes.data, ids.ctypes.data)
    model = []
    for i, v in enumerate(seq):
        if v < val[0]:
            L.orc_maxheap_pop(k, val.ctypes.data, ids.ctypes.data)
            L.orc_maxheap_push(k, val.ctypes.data, ids.ctypes.data, float(v), i)
        model.append((v, i))
        # heap property (1-based children 2i, 2i+1)
        for j in range(1, k):
            assert val[(j + 1) // 2 - 1] >= val[j]
    return val, ids


@pytest.mark.parametrize("k", [1, 2, 7, 64])
def test_maxheap_keeps_k_smallest(k):
    rng = np.random.default_rng(k)
    seq = rng.normal(size=500).astype(F)
    val, ids = _heap_model_run(k, seq)
    want = np.sort(seq)[:k]
    assert np.array_equal(np.sort(val), want)
    assert np.array_equal(np.sort(seq[ids]), want)


def test_maxheap_initial_state_and_strict_less():
    L = orc.lib()
    val = np.empty(3, F)
    ids = np.empty(3, np.int64)
    L.orc_maxheap_heapify(3, val.ctypes.data, ids.ctypes.data)
    assert (val == np.finfo(F).max).all() and (ids == -1).all()
    # equal distances: the first one scanned stays (strict '<')
    val1, ids1 = _heap_model_run(1, np.array([5.0, 5.0, 5.0], F))
    assert ids1[0] == 0


# ---- HNSW: independent pure-Python restatement of hnswalg.cpp:48-109 with heapq ------------------------
def py_search_knn(vectors, counts, links, ep, q, ef, k):
    dist = lambda i: float(np_l2_ref(q, vectors[i]))
    visited = {ep}
    d0 = dist(ep)
    top = [(-d0, -ep)]           # max-heap of (dist, id) via negation
    cand = [(d0, -ep)]           # pops smallest dist, then LARGEST id (pair<-dist,id> max-heap)
    lower = d0
    while cand:
        d, nid = cand[0]
        if d > lower:
            break
        heapq.heappop(cand)
        node = -nid
        for j in range(counts[node]):
            t = int(links[node, j])
            if t in visited:
                continue
            visited.add(t)
            dt = dist(t)
            if -top[0][0] > dt or len(top) < ef:
                heapq.heappush(cand, (dt, -t))
                heapq.heappush(top, (-dt, -t))
                if len(top) > ef:
                    heapq.heappop(top)
                lower = -top[0][0]
    while len(top) > k:
        heapq.heappop(top)
    res = sorted([(-a, -b) for a, b in top])
    return [r[1] for r in res], [r[0] for r in res]


@pytest.mark.parametrize("ef,k", [(1, 1), (10, 4), (40, 16), (64, 64)])
def test_hnsw_walk_matches_python_model(ef, k):
    rng = np.random.default_rng(9)
    base = synth.sift_like(rng, 100, 32)
    cents = np.concatenate([base, base[:50]])  # duplicates -> exact ties
    g = orc.Hnsw.build(cents, M=6, efConstruction=40)
    counts, links, vecs = g.counts.copy(), g.links.copy(), g.vectors.copy()
    for t in range(20):
        q = (base[rng.integers(100)] + rng.normal(0, 3, 32)).astype(F) if t % 4 else base[t].copy()
        ids, dist = g.search_knn(q, ef, k)
        pi, pd = py_search_knn(vecs, counts, links, g.enterpoint, q, ef, k)
        assert ids.tolist() == pi
        assert [float(x) for x in dist] == pd


def test_hnsw_graph_invariants():
    rng = np.random.default_rng(1)
    cents = synth.sift_like(rng, 400, 64)
    g = orc.Hnsw.build(cents, M=8, efConstruction=50)
    c, l = g.counts, g.links
    assert c.max() <= 16 and c[1:].min() >= 1
    for i in range(400):
        nb = l[i, :c[i]]
        assert len(set(nb.tolist())) == len(nb) and i not in nb
    assert g.enterpoint == 0


# ---- search: hand-checkable IVFADC case ------------------------------------------------------------------
def test_search_ivf_formula_by_hand():
    """dist = (||x-c||^2 - ||c||^2) + norm - 2 <x, y_R>  (IndexIVF_HNSW.cpp:206-233) on a 2-list index."""
    d, M = 16, 4
    cents = np.zeros((2, d), F)
    cents[1, 0] = 10.0
    g = orc.Hnsw.build(cents, M=2, efConstruction=4)
    cb = np.zeros((M, 256, 4), F)
    cb[0, 1, 0] = 1.0   # code word 1 of sub-space 0 = e0
    cb[0, 2, 0] = 2.0
    norm_table = np.arange(256, dtype=F)
    offsets = np.array([0, 2, 3], np.uint64)
    codes = np.zeros((3, M), np.uint8)
    codes[0, 0], codes[1, 0], codes[2, 0] = 1, 2, 1
    norm_codes = np.array([1, 4, 121], np.uint8)   # ||0+e0||^2 = 1, ||2 e0||^2 = 4, ||(10+1) e0||^2 = 121
    ids = np.array([100, 101, 102], np.uint32)
    ix = orc.Index(d, M, g, cb, norm_table, offsets, ids, codes, norm_codes, g.centroid_norms())
    ix.set_params(2, 10 ** 9, 4)
    x = np.zeros(d, F)
    x[0] = 3.0
    dist, lab, st = ix.search(x, k=3)
    got = dict(zip(lab.tolist(), dist.tolist()))
    # true squared distances to the reconstructed points e0, 2e0, 11e0 from 3e0: 4, 1, 64
    assert got == {100: 4.0, 101: 1.0, 102: 64.0}
    assert st.ncode == 3
    d1, l1, _ = ix.search(x, k=1)
    assert l1[0] == 101 and d1[0] == 1.0
    # max_codes: stop after the list that reaches it (IndexIVF_HNSW.cpp:290-292)
    ix.set_params(2, 2, 4)
    d2, l2, st2 = ix.search(x, k=3)
    assert st2.ncode == 2 and sorted(l2.tolist()) == [-1, 100, 101]


def test_empty_list_is_skipped_and_does_not_count():
    c = synth.make_corpus(seed=3, nc=64, d=32, M=4, n_base=2000, nq=8, efConstruction=40, empty_frac=0.3)
    sizes = np.diff(c["offsets"].astype(np.int64))
    assert (sizes == 0).any()
    ix = synth.oracle_index(c)
    ix.set_params(64, 10 ** 9, 64)
    d, l, cid, cd, st = ix.search_batch(c["queries"], k=1)
    assert st.ncode == 8 * 2000            # every list probed -> every code scored once per query
    assert st.nseg == 8 * int((sizes > 0).sum())


def test_grouping_without_pruning_scores_every_subgroup_and_matches_bruteforce_formula():
    c = synth.make_corpus(seed=5, nc=64, d=32, M=4, n_base=3000, nq=6, nsubc=4, efConstruction=60)
    ix = synth.oracle_index(c)
    ix.set_params(8, 10 ** 9, 32, do_pruning=False)
    dist, lab, cid, cd, st = ix.search_batch(c["queries"], k=1)
    off = c["offsets"].astype(np.int64)
    g = c["graph"]
    for qi in range(6):
        x = c["queries"][qi]
        tab = orc.inner_prod_table(x, c["pq_centroids"], 4)
        best = (np.inf, -1)
        for pi, cc in enumerate(cid[qi]):
            cc = int(cc)
            a = c["alphas"][cc]
            term1 = F(F(1 - a) * F(cd[qi, pi] - c["centroid_norms"][cc]))
            pos = off[cc]
            for s in range(4):
                sg = int(c["subgroup_sizes"][cc, s])
                if sg == 0:
                    continue
                nn = int(c["nn_centroid_idxs"][cc, s])
                qn = F(orc.l2sqr(x, g.vectors[nn]))
                term2 = F(a * F(qn - c["centroid_norms"][nn]))
                for j in range(pos, pos + sg):
                    sm = F(0)
                    for m in range(4):
                        sm = F(sm + tab[m, c["codes"][j, m]])
                    dd = F(F(F(term1 + term2) + c["norm_table"][c["norm_codes"][j]]) - F(2 * sm))
                    if dd < best[0]:
                        best = (dd, int(c["ids"][j]))
                pos += sg
        assert (F(best[0]), best[1]) == (dist[qi, 0], lab[qi, 0])


def test_batch_threads_agree_with_serial():
    c = synth.make_corpus(seed=6, nc=128, d=64, M=8, n_base=6000, nq=40, nsubc=8, efConstruction=60)
    ix = synth.oracle_index(c)
    ix.set_params(8, 900, 32, do_pruning=True)
    a = ix.search_batch(c["queries"], k=3, nthreads=1)
    b = ix.search_batch(c["queries"], k=3, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[4].ncode, a[4].nseg) == (b[4].ncode, b[4].nseg)
    for qi in range(5):
        d1, l1, _ = ix.search(c["queries"][qi], k=3)
        assert np.array_equal(d1, a[0][qi]) and np.array_equal(l1, a[1][qi])


# ---- on-disk formats ---------------------------------------------------------------------------------------
def test_index_file_layout_ivf(tmp_path):
    """IndexIVF_HNSW.cpp:637-663: size_t d, size_t nc, then ids / codes / norm codes as uint32-counted
    vectors (utils.h:59-81), then centroid_norms."""
    c = synth.make_corpus(seed=8, nc=16, d=32, M=4, n_base=300, nq=4, efConstruction=20)
    ix = synth.oracle_index(c)
    p = str(tmp_path / "ivf.index")
    ix.write(p)
    raw = open(p, "rb").read()
    d, nc = struct.unpack_from("<QQ", raw, 0)
    assert (d, nc) == (32, 16)
    pos = 16
    off = c["offsets"].astype(np.int64)
    for cc in range(16):
        (n,) = struct.unpack_from("<I", raw, pos)
        assert n == off[cc + 1] - off[cc]
        got = np.frombuffer(raw, np.uint32, n, pos + 4)
        assert np.array_equal(got, c["ids"][off[cc]:off[cc + 1]])
        pos += 4 + 4 * n
    for cc in range(16):
        (n,) = struct.unpack_from("<I", raw, pos)
        assert n == 4 * (off[cc + 1] - off[cc])
        pos += 4 + n
    for cc in range(16):
        (n,) = struct.unpack_from("<I", raw, pos)
        pos += 4 + n
    (n,) = struct.unpack_from("<I", raw, pos)
    assert n == 16 and pos + 4 + 64 == len(raw)
    back = orc.read_index(p, grouping=False)
    for key in ("offsets", "ids", "codes", "norm_codes", "centroid_norms"):
        assert np.array_equal(back[key], c[key]), key


def test_index_file_roundtrip_grouping(tmp_path):
    c = synth.make_corpus(seed=9, nc=32, d=32, M=4, n_base=600, nq=4, nsubc=4, efConstruction=30, empty_frac=0.2)
    ix = synth.oracle_index(c)
    p = str(tmp_path / "grp.index")
    ix.write(p)
    back = orc.read_index(p, grouping=True)
    assert back["nsubc"] == 4
    for key in ("offsets", "ids", "codes", "norm_codes", "centroid_norms", "alphas", "nn_centroid_idxs",
                "subgroup_sizes", "inter_centroid_dists"):
        assert np.array_equal(back[key], c[key]), key


def test_hnsw_files_layout(tmp_path):
    """hnswalg.cpp:236-265: info = 7 size_t + 1 uint32 (60 bytes, no padding); edges = per node uint32 n + n ids."""
    rng = np.random.default_rng(2)
    cents = synth.sift_like(rng, 50, 32)
    g = orc.Hnsw.build(cents, M=4, efConstruction=20)
    pi, pe, pd = (str(tmp_path / n) for n in ("info.bin", "edges.ivecs", "centroids.fvecs"))
    g.save(pi, pe)
    raw = open(pi, "rb").read()
    assert len(raw) == 60
    maxel, ep, data_size, offset_data, per_elem, M, maxM, links0 = struct.unpack("<QIQQQQQQ", raw)
    assert (maxel, ep, data_size, M, maxM) == (50, 0, 128, 4, 8)
    assert links0 == 8 * 4 + 1 and offset_data == links0 and per_elem == links0 + data_size
    with open(pd, "wb") as f:
        for v in cents:
            f.write(struct.pack("<I", 32))
            f.write(v.tobytes())
    g2 = orc.Hnsw.load(pi, pd, pe)
    assert np.array_equal(g2.counts, g.counts) and np.array_equal(g2.vectors, g.vectors)
    for i in range(50):
        assert np.array_equal(g2.links[i, :g.counts[i]], g.links[i, :g.counts[i]])
    assert os.path.getsize(pe) == 4 * 50 + 4 * int(g.counts.sum())


# ---------------------------------------------------------------------------------------------- construction side
def np_l2_sse(x, y):
    s = np.zeros(4, F)
    n, i = len(x), 0
    while i + 4 <= n:
        t = (x[i:i + 4] - y[i:i + 4]).astype(F)
        s = (s + (t * t).astype(F)).astype(F)
        i += 4
    for l in range(n - i):
        t = F(x[i + l] - y[i + l])
        s[l] = F(s[l] + F(t * t))
    return F(F(s[0] + s[1]) + F(s[2] + s[3]))


def np_add_batch_encode(x, idx, cents, cb, nt, A):
    """Independent float32 evaluation of IndexIVF_HNSW.cpp:86-121 in the order the oracle documents."""
    M, _, dsub = cb.shape
    d = cents.shape[1]
    codes = np.zeros((len(x), M), np.uint8)
    ncodes = np.zeros(len(x), np.uint8)
    norms = np.zeros(len(x), F)
    for i, xi in enumerate(x):
        cen = cents[idx[i]]
        res = (xi + (F(-1.0) * cen).astype(F)).astype(F)
        enc = res
        if A is not None:
            enc = np.array([_fma_chain(A[r], res) for r in range(d)], F)
        for m in range(M):
            best, arg = F(1e20), -1
            for c in range(256):
                dis = np_l2_sse(enc[m * dsub:(m + 1) * dsub], cb[m, c])
                if dis < best:
                    best, arg = dis, c
            codes[i, m] = arg & 0xff
        dec = np.concatenate([cb[m, codes[i, m]] for m in range(M)]).astype(F)
        if A is not None:
            dec = np.array([_fma_chain(A[:, k], dec) for k in range(d)], F)
        rec = (dec + (F(1.0) * cen).astype(F)).astype(F)
        norms[i] = np_ip_sse(rec, rec)
        best, arg = F(1e20), -1
        for c in range(256):
            t = F(norms[i] - nt[c])
            dis = F(t * t)
            if dis < best:
                best, arg = dis, c
        ncodes[i] = arg & 0xff
    return codes, ncodes, norms


def _fma_chain(a, x):
    """acc = fmaf(a[k], x[k], acc) for k = 0..: one rounding per step, evaluated in float64 (exact product of two
    float32 fits; the sum of a float32 and that product rounds to float32 correctly via float64 except in
    double-rounding corner cases, which the assertion below would expose)."""
    acc = F(0.0)
    for k in range(len(a)):
        acc = F(np.float64(a[k]) * np.float64(x[k]) + np.float64(acc))
    return acc


@pytest.mark.parametrize("d,M,opq", [(32, 4, False), (32, 8, True), (48, 8, False)])
def test_add_batch_encode_matches_numpy_model(d, M, opq):
    rng = np.random.default_rng(400 + d + M)
    nc, n = 40, 24
    cents = synth.sift_like(rng, nc, d)
    graph = orc.Hnsw.build(cents, M=4, efConstruction=20)
    cb = rng.normal(0, 9.0, size=(M, 256, d // M)).astype(F)
    nt = np.sort(rng.normal(d * 2100.0, d * 300.0, size=256)).astype(F)
    A = synth.random_rotation(rng, d) if opq else None
    ox = orc.Index(d, M, graph, cb, nt, np.zeros(nc + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, M), np.uint8),
                   np.zeros(0, np.uint8), np.zeros(nc, F), opq_A=A)
    ox.set_params(1, 0, 30)
    x = (cents[rng.choice(nc, n)] + rng.normal(0, 15.0, size=(n, d))).astype(F)
    idx, codes, ncodes, norms = ox.add_batch_encode(x)
    # assign = searchKnn(x, 1) with efSearch (IndexIVF_HNSW.cpp:68-72)
    for i in range(n):
        rid, _ = graph.search_knn(x[i], 30, 1)
        assert idx[i] == rid[0]
    mc, mn, mnorm = np_add_batch_encode(x, idx, cents, cb, nt, A)
    assert np.array_equal(codes, mc)
    assert np.array_equal(norms.view(np.uint32), mnorm.view(np.uint32))
    assert np.array_equal(ncodes, mn)
    # supplied assignments take the other branch (:79-80)
    pre = rng.integers(0, nc, size=n).astype(np.uint32)
    idx2, codes2, ncodes2, _ = ox.add_batch_encode(x, pre)
    mc2, mn2, _ = np_add_batch_encode(x, pre, cents, cb, nt, A)
    assert np.array_equal(idx2, pre) and np.array_equal(codes2, mc2) and np.array_equal(ncodes2, mn2)


def test_add_group_encode_matches_numpy_model():
    """IndexIVF_HNSW_Grouping.cpp:43-125 for one group against an independent float32 evaluation: neighbours from
    searchKnn(centroid, nsubc + 1) minus the nearest, alpha as the max-heap of (-dist, (numerator, denominator))
    picks it, first nearest sub-centroid, then the add_batch chain against the sub-centroids."""
    rng = np.random.default_rng(77)
    d, M, nsubc, nc, n = 32, 4, 5, 60, 30
    s = synth.make_encode_case(1, nc, d, M, True, n=8, hnsw_M=8)
    ox, cents, cb, nt, A = s["ox"], s["cents"], s["cb"], s["nt"], s["A"]
    ox.set_params(1, 0, 40)
    c = 7
    x = (cents[c] + rng.normal(0, 9, size=(n, d))).astype(F)
    nn, alpha, sub, codes, ncodes = ox.add_group_encode(nsubc, c, x)
    ids, dist = s["graph"].search_knn(cents[c], 40, nsubc + 1)
    assert np.array_equal(nn, ids[1:])
    cen = cents[c]
    cv = np.stack([(cents[i] + (F(-1) * cen).astype(F)).astype(F) for i in nn])
    num_s, den_s = F(0), F(0)
    for p in x:
        pv = (p + (F(-1) * cen).astype(F)).astype(F)
        best = None
        for k in range(nsubc):
            num = np_ip_sse(cv[k], pv)
            num = num if num > 0 else F(0)
            den = dist[k + 1]
            subc = (cen + (F(num / den) * cv[k]).astype(F)).astype(F)
            cand = (-np_l2_ref(p, subc), num, den)
            if best is None or best < cand:
                best = cand
        num_s, den_s = F(num_s + best[1]), F(den_s + best[2])
    model_alpha = F(num_s / den_s)
    assert np.float32(alpha).view(np.uint32) == model_alpha.view(np.uint32)
    subcents = np.stack([(cen + (model_alpha * cv[k]).astype(F)).astype(F) for k in range(nsubc)])
    msub = np.array([int(np.argmin([np_l2_ref(subcents[k], p) for k in range(nsubc)])) for p in x], np.uint32)
    assert np.array_equal(sub, msub)
    mc, mn, _ = np_add_batch_encode(x, msub, subcents, cb, nt, A)
    assert np.array_equal(codes, mc) and np.array_equal(ncodes, mn)
    # an empty group records the neighbours and nothing else
    nn0, alpha0, sub0, codes0, nc0 = ox.add_group_encode(nsubc, c, np.zeros((0, d), F))
    assert np.array_equal(nn0, nn) and alpha0 == 0 and len(sub0) == 0
