"""Seeded synthetic corpora for the parity tests and bench.py.

Two kinds:
  * make_corpus(): a small, *real* index built the way the reference builds one (IndexIVF_HNSW.cpp:75-138,
    IndexIVF_HNSW_Grouping.cpp:43-157): base = centroid + noise -> residual -> (OPQ) -> PQ encode ->
    reconstruct -> quantised norm; HNSW graph by the oracle's reference-identical serial construction.
  * make_throughput_corpus(): SIFT1B-shaped tables with list sizes only; codes are the counter-hash byte
    stream the device generates itself (ivfhnsw_gpu_upload_ivf_synthetic), reproduced here in numpy.

No reference data, no downloads: everything is generated from the seed.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import orc  # noqa: E402

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
NORM_SEED_XOR = np.uint64(0x6e6f726d6e6f726d)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_bytes(seed, first_byte, nbytes):
    """Bytes [first_byte, first_byte+nbytes) of the device's synthetic stream (kernels_search.hip fill_bytes)."""
    if nbytes == 0:
        return np.zeros(0, np.uint8)
    if nbytes > (1 << 30):  # 1B-vector corpora: in pieces, so that the temporaries stay small
        out = np.empty(nbytes, np.uint8)
        step = 1 << 29
        for a in range(0, nbytes, step):
            n = min(step, nbytes - a)
            out[a:a + n] = hash_bytes(seed, first_byte + a, n)
            if (a // step) % 8 == 7:   # minutes of work: keep a watcher of stderr informed
                print("[synth] host copy of the synthetic stream: %.0f / %.0f GiB" % ((a + n) / 2 ** 30, nbytes / 2 ** 30),
                      file=sys.stderr, flush=True)
        return out
    w0 = first_byte // 8
    w1 = (first_byte + nbytes + 7) // 8
    with np.errstate(over="ignore"):
        w = np.arange(w0, w1, dtype=np.uint64)
        v = _mix64(np.uint64(seed) + (w + np.uint64(1)) * GOLDEN)
    b = v.view(np.uint8)  # little endian: byte i of word = (v >> 8i) & 0xff
    s = first_byte - w0 * 8
    return b[s:s + nbytes].copy()


def sift_like(rng, n, d):
    """Non-negative, byte-ranged vectors (SIFT descriptors are uint8 histograms)."""
    x = rng.normal(30.0, 35.0, size=(n, d))
    return np.clip(np.rint(x), 0, 255).astype(np.float32)


def clustered_centroids(rng, n, d, per=64, spread=18.0):
    """n centroid-like rows in tight clusters of ~`per` around SIFT-like cluster centres: what the k-means centroids of
    clustered descriptors (SIFT, DEEP) look like -- a query's nearest centroids are each other's neighbours -- where
    sift_like() rows are iid and every row's neighbours are strangers to each other."""
    centres = sift_like(rng, (n + per - 1) // per, d)
    x = centres[rng.integers(0, len(centres), n)] + rng.normal(0, spread, (n, d))
    return np.maximum(x, 0).astype(np.float32)


def random_rotation(rng, d):
    q, r = np.linalg.qr(rng.normal(size=(d, d)))
    q = q * np.sign(np.diag(r))
    return np.ascontiguousarray(q.astype(np.float32))


def _pq_encode(res, cb):
    """Nearest code word per sub-space.  res [n,d], cb [M,256,dsub] -> codes [n,M] uint8."""
    n, d = res.shape
    M, _, dsub = cb.shape
    codes = np.empty((n, M), np.uint8)
    for m in range(M):
        sub = res[:, m * dsub:(m + 1) * dsub]
        c = cb[m]
        dist = (sub * sub).sum(1, keepdims=True) - 2.0 * sub @ c.T + (c * c).sum(1)[None, :]
        codes[:, m] = dist.argmin(1).astype(np.uint8)
    return codes


def _pq_decode(codes, cb):
    n, M = codes.shape
    dsub = cb.shape[2]
    out = np.empty((n, M * dsub), np.float32)
    for m in range(M):
        out[:, m * dsub:(m + 1) * dsub] = cb[m][codes[:, m]]
    return out


def make_corpus(seed=1234, nc=256, d=128, M=16, n_base=20000, opq=False, nsubc=0, hnsw_M=16, efConstruction=200,
                empty_frac=0.05, noise=12.0, nq=64, query_noise=10.0):
    """Small real index.  Returns a dict of numpy arrays + the oracle graph handle."""
    rng = np.random.default_rng(seed)
    dsub = d // M
    centroids = sift_like(rng, nc, d)
    graph = orc.Hnsw.build(centroids, M=hnsw_M, efConstruction=efConstruction)

    # list membership: skewed sizes, a few empty lists (reference skips them, IndexIVF_HNSW.cpp:271)
    w = rng.gamma(2.0, 1.0, size=nc)
    w[rng.random(nc) < empty_frac] = 0.0
    w /= w.sum()
    assign = rng.choice(nc, size=n_base, p=w).astype(np.uint32)
    base = centroids[assign] + rng.normal(0.0, noise, size=(n_base, d)).astype(np.float32)
    base = base.astype(np.float32)
    ids = np.arange(n_base, dtype=np.uint32)

    A = random_rotation(rng, d) if opq else None

    out = dict(seed=seed, d=d, nc=nc, code_size=M, centroids=centroids, opq_A=A, nsubc=nsubc)

    if nsubc:
        # IndexIVF_HNSW_Grouping.cpp:47-63: the nsubc nearest other centroids, ascending distance
        nn = np.zeros((nc, nsubc), np.uint32)
        for c in range(nc):
            found, _ = graph.search_knn(centroids[c], max(efConstruction, nsubc + 1), nsubc + 1)
            assert len(found) == nsubc + 1, "graph too small for nsubc"
            nn[c] = found[1:]
        alphas = rng.uniform(0.0, 0.5, size=nc).astype(np.float32)
        out.update(nn_centroid_idxs=nn, alphas=alphas)
        # sub-centroid of (c, s) = c + alpha * (nn - c); each point goes to its nearest sub-centroid
        subc_of = np.zeros(n_base, np.uint32)
        sub_centroid = np.empty((n_base, d), np.float32)
        for c in range(nc):
            idx = np.nonzero(assign == c)[0]
            if idx.size == 0:
                continue
            sc = centroids[c][None, :] + alphas[c] * (centroids[nn[c]] - centroids[c][None, :])
            dist = ((base[idx][:, None, :] - sc[None, :, :]) ** 2).sum(2)
            s = dist.argmin(1)
            subc_of[idx] = s
            sub_centroid[idx] = sc[s]
        anchor = sub_centroid
        order = np.lexsort((ids, subc_of, assign))  # list, then sub-group, then insertion order
    else:
        anchor = centroids[assign]
        order = np.argsort(assign, kind="stable")

    residual = (base - anchor).astype(np.float32)
    if opq:
        residual = (residual @ A.T).astype(np.float32)  # y = A x

    # code books: 256 sampled residual sub-vectors per sub-space, jittered
    cb = np.empty((M, 256, dsub), np.float32)
    for m in range(M):
        pick = rng.choice(n_base, size=256, replace=n_base < 256)
        cb[m] = residual[pick, m * dsub:(m + 1) * dsub] + rng.normal(0, 0.5, size=(256, dsub)).astype(np.float32)
    codes = _pq_encode(residual, cb)
    dec = _pq_decode(codes, cb)
    if opq:
        dec = (dec @ A).astype(np.float32)  # x = A^T y
    recon = (anchor + dec).astype(np.float32)
    norms = (recon.astype(np.float64) ** 2).sum(1).astype(np.float32)
    # norm quantiser: 256 sorted quantiles; nearest entry
    norm_table = np.quantile(norms, (np.arange(256) + 0.5) / 256).astype(np.float32)
    norm_codes = np.abs(norms[:, None] - norm_table[None, :]).argmin(1).astype(np.uint8)

    counts = np.bincount(assign, minlength=nc).astype(np.uint64)
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(counts)
    out.update(offsets=offsets, ids=ids[order].copy(), codes=codes[order].copy(),
               norm_codes=norm_codes[order].copy(), pq_centroids=cb, norm_table=norm_table,
               centroid_norms=graph.centroid_norms(), base=base)

    if nsubc:
        sg = np.zeros((nc, nsubc), np.uint32)
        np.add.at(sg, (assign, subc_of), 1)
        out.update(subgroup_sizes=sg, inter_centroid_dists=graph.inter_centroid_dists(out["nn_centroid_idxs"]))

    # the drivers rotate the graph's centroids after the index is complete (tests/test_ivfhnsw_sift1b.cpp:164-167)
    if opq:
        graph.rotate(A)
    out["graph"] = graph

    # queries near base points (so that the right list is usually probed), plus a few exact duplicates
    qsrc = rng.choice(n_base, size=nq, replace=False)
    queries = base[qsrc] + rng.normal(0.0, query_noise, size=(nq, d)).astype(np.float32)
    queries[: max(1, nq // 16)] = base[qsrc[: max(1, nq // 16)]]
    out["queries"] = np.ascontiguousarray(queries, np.float32)
    out["query_src"] = qsrc
    return out


def oracle_index(c):
    """orc.Index over a corpus dict."""
    return orc.Index(c["d"], c["code_size"], c["graph"], c["pq_centroids"], c["norm_table"], c["offsets"], c["ids"],
                     c["codes"], c["norm_codes"], c["centroid_norms"], opq_A=c["opq_A"], nsubc=c["nsubc"],
                     alphas=c.get("alphas"), nn_centroid_idxs=c.get("nn_centroid_idxs"),
                     subgroup_sizes=c.get("subgroup_sizes"), inter_centroid_dists=c.get("inter_centroid_dists"))


def make_recall_corpus(pkg, seed, nc, n_base, d=128, M=16, nq=10000, device=0, base_noise=10.0, query_noise=6.0,
                       train_n=65536, train_iters=8, ef_assign=220, batch=1 << 20, log=None):
    """A RECALL-BEARING index built by the library's own pipeline on the device, from clustered data (VERDICT round 2,
    item 4): what the reference's drivers do to SIFT1B (tests/test_ivfhnsw_sift1b.cpp:47-167), at a size that builds in
    seconds.
      centroids   clustered_centroids(): tight clusters of ~64 around SIFT-like centres (k-means centroids of clustered data)
      graph       ivfhnsw_gpu_build_graph (the insertion loop, exact candidates), M 16 / maxM 32
      base        centroid + N(0, base_noise): a mixture; ids in generation order
      code books  ProductQuantizer::train's Lloyd iterations on the device (ivfhnsw_gpu_pq_train) over the residuals of a
                  sample to ITS assigned centroids (IndexIVF_HNSW::train_pq, IndexIVF_HNSW.cpp:536-593); norm table = 256
                  quantiles of the sample's reconstructed norms
      lists       ivfhnsw_gpu_encode (assign by the walk at efSearch 220, residual, PQ code, norm code: add_batch,
                  IndexIVF_HNSW.cpp:75-121), appended in id order
      queries     base rows + N(0, query_noise); ground truth = the exact nearest base row (ivfhnsw_gpu_knn)
    Returns a dict like make_corpus() (graph as arrays: counts / links / centroids) + gt [nq]."""
    import time
    rng = np.random.default_rng(seed)
    t0 = time.time()
    say = log or (lambda *a: None)
    centroids = clustered_centroids(rng, nc, d)
    g = pkg.GpuIndex(device)
    counts, links = g.build_graph(centroids, 16, 32, 64)
    g.upload_quantizer(counts, links, centroids, 0)
    sizes = list_sizes(rng, nc, n_base).astype(np.int64)
    gen = np.repeat(np.arange(nc, dtype=np.uint32), sizes)
    rng.shuffle(gen)
    base = np.empty((n_base, d), np.float32)
    for a in range(0, n_base, batch):
        b = min(n_base, a + batch)
        base[a:b] = centroids[gen[a:b]] + rng.standard_normal((b - a, d), dtype=np.float32) * np.float32(base_noise)
    say("[recall corpus] %d centroids + graph, %d base vectors: %.1fs" % (nc, n_base, time.time() - t0))
    # code books from a sample's residuals
    t0 = time.time()
    pick = rng.choice(n_base, size=min(train_n, n_base), replace=False)
    xs = base[pick]
    idx_s, _ = g.coarse(xs, 1, ef_assign)
    res = (xs - centroids[idx_s[:, 0]]).astype(np.float32)
    dsub = d // M
    cb0 = np.stack([res[rng.choice(len(res), 256, replace=len(res) < 256), m * dsub:(m + 1) * dsub] for m in range(M)])
    cb, _ = g.pq_train(res, M, cb0, niter=train_iters)
    g.upload_codebooks(d, M, cb, np.arange(256, dtype=np.float32))
    _, codes_s, _ = g.encode(xs, precomputed_idx=idx_s[:, 0])
    recon = centroids[idx_s[:, 0]] + _pq_decode(codes_s, cb)
    norms = (recon.astype(np.float64) ** 2).sum(1)
    norm_table = np.quantile(norms, (np.arange(256) + 0.5) / 256).astype(np.float32)
    g.upload_codebooks(d, M, cb, norm_table)
    say("[recall corpus] code books (%d Lloyd iterations on %d residuals) + norm table: %.1fs" % (train_iters, len(res), time.time() - t0))
    # the lists
    t0 = time.time()
    idx = np.empty(n_base, np.uint32)
    codes = np.empty((n_base, M), np.uint8)
    ncodes = np.empty(n_base, np.uint8)
    for a in range(0, n_base, batch):
        b = min(n_base, a + batch)
        idx[a:b], codes[a:b], ncodes[a:b] = g.encode(base[a:b], efSearch=ef_assign)
    order = np.argsort(idx, kind="stable")          # list by list, insertion (= id) order inside a list
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(idx, minlength=nc))
    say("[recall corpus] %d vectors assigned (efSearch %d) and encoded on the device: %.1fs (%.2f M vectors/s)"
        % (n_base, ef_assign, time.time() - t0, n_base / (time.time() - t0) / 1e6))
    # queries and their exact nearest neighbours
    t0 = time.time()
    qsrc = rng.choice(n_base, size=nq, replace=False)
    queries = (base[qsrc] + rng.standard_normal((nq, d), dtype=np.float32) * np.float32(query_noise)).astype(np.float32)
    gt, _ = g.knn(base, 1, queries)
    g.close()
    say("[recall corpus] exact ground truth of %d queries against %d vectors: %.1fs" % (nq, n_base, time.time() - t0))
    return dict(seed=seed, d=d, nc=nc, code_size=M, centroids=centroids, counts=counts, links=links, opq_A=None, nsubc=0,
                offsets=offsets, ids=order.astype(np.uint32), codes=codes[order], norm_codes=ncodes[order],
                pq_centroids=cb, norm_table=norm_table, queries=queries, gt=gt[:, 0].astype(np.int64), query_src=qsrc,
                centroid_norms=(centroids.astype(np.float64) ** 2).sum(1).astype(np.float32), n_base=n_base,
                assign_agrees_with_generator=float((idx == gen).mean()))


def list_sizes(rng, nc, n_total, sigma=0.6, cap=65536):
    """Log-normal list sizes summing to n_total, every list <= 65536 (IndexIVF_HNSW.cpp:17 scratch size)."""
    w = rng.lognormal(0.0, sigma, size=nc)
    sizes = np.floor(w / w.sum() * n_total).astype(np.int64)
    sizes = np.minimum(sizes, cap)
    short = n_total - int(sizes.sum())
    if short > 0:
        bump = rng.choice(nc, size=short, replace=True)
        np.add.at(sizes, bump, 1)
        sizes = np.minimum(sizes, cap)
    return sizes.astype(np.uint64)


def make_throughput_tables(seed, nc, d, M, n_total, kind="sift"):
    """Tables of a throughput corpus (no codes): centroids, code books, norm table, list offsets.
    kind "sift": byte-ranged non-negative iid rows; "clustered": SIFT-ranged rows in tight clusters of ~64 (k-means
    centroids of clustered descriptors); "deep": unit vectors with components of both signs (DEEP1B)."""
    rng = np.random.default_rng(seed)
    dsub = d // M
    if kind == "deep":
        centroids = rng.normal(0.0, 1.0, size=(nc, d)).astype(np.float32)
        centroids /= np.linalg.norm(centroids, axis=1, keepdims=True)
    elif kind == "clustered":
        centroids = clustered_centroids(rng, nc, d)
    else:
        centroids = sift_like(rng, nc, d)
    sizes = list_sizes(rng, nc, n_total)
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(sizes)
    if kind == "deep":
        cb = rng.normal(0.0, 0.03, size=(M, 256, dsub)).astype(np.float32)
        norm_table = np.sort(rng.normal(1.0, 0.1, size=256)).astype(np.float32)
    else:
        cb = rng.normal(0.0, 12.0, size=(M, 256, dsub)).astype(np.float32)
        # a reconstructed SIFT-like vector has squared norm around d * (30^2 + 35^2)
        norm_table = np.sort(rng.normal(d * 2100.0, d * 300.0, size=256)).astype(np.float32)
    return dict(seed=seed, d=d, nc=nc, code_size=M, centroids=centroids, offsets=offsets, pq_centroids=cb,
                norm_table=norm_table, opq_A=None, nsubc=0)


def synthetic_codes(seed, offsets, code_size):
    """Host copy of what ivfhnsw_gpu_upload_ivf_synthetic generates on the device."""
    n = int(offsets[-1])
    codes = hash_bytes(seed, 0, n * code_size).reshape(n, code_size)
    norm_codes = hash_bytes(int(np.uint64(seed) ^ NORM_SEED_XOR), 0, n)
    ids = np.arange(n, dtype=np.uint32)
    return ids, codes, norm_codes


def knn_table(x, k, device=None):
    """ids of the k nearest OTHER rows of every row of x, ascending (distance, id): the library's exact neighbour
    table on the matrix cores (ivfhnsw_gpu_knn, kernels_knn.hip; tests/test_gpu_knn.py pins it to the oracle).
    Rounds 1-2 used torch's GEMM + top-k here; nothing of the corpus preparation runs on torch kernels any more."""
    import __graft_entry__ as ge
    g = ge.load_pkg().GpuIndex(getattr(device, "index", None) or 0)   # a torch.device, or None = GPU 0
    try:
        ids, _ = g.knn(np.ascontiguousarray(x, np.float32), k)
    finally:
        g.close()
    return ids


def knn_graph(centroids, M=16, maxM=32, device=None, chunk=None):
    """A navigable small-world graph for throughput runs: each node links to its M nearest neighbours, then
    reverse links are added up to maxM (the same degree bounds the reference's construction keeps,
    IndexIVF_HNSW.cpp:50, hnswalg.cpp:171-184).  The reference-identical serial construction (orc.Hnsw.build,
    used by every parity test) would take minutes at 10^5-10^6 nodes; graph construction is outside the
    search path so the throughput corpus uses this brute-force stand-in: the exact M-NN table from
    the library's MFMA kernel, the reverse links on the host.  Returns (counts u8 [n], links u32 [n, maxM])."""
    n = len(centroids)
    knn = knn_table(centroids, M, device)
    links = np.zeros((n, maxM), np.uint32)
    links[:, :M] = knn
    counts = np.full(n, M, np.int64)
    # reverse edges, in order of the forward edge's rank (closest first), while room remains
    # vectorised per rank: for rank r, add edge dst->src where dst has room and the edge is not already there
    for r in range(M):
        s_r = np.arange(n, dtype=np.uint32)
        d_r = knn[:, r]
        # skip when the reverse edge already exists as a forward edge of d_r
        exists = (knn[d_r] == s_r[:, None]).any(1)
        cand = np.nonzero(~exists)[0]
        # several sources may target the same node in this round: take them in source order
        tgt = d_r[cand]
        o = np.argsort(tgt, kind="stable")
        tgt, srcs = tgt[o], s_r[cand][o]
        first = np.r_[True, tgt[1:] != tgt[:-1]]
        rank_in_group = np.arange(len(tgt)) - np.maximum.accumulate(np.where(first, np.arange(len(tgt)), 0))
        slot = counts[tgt] + rank_in_group
        ok = slot < maxM
        links[tgt[ok], slot[ok]] = srcs[ok]
        np.add.at(counts, tgt[ok], 1)
    return counts.astype(np.uint8), links


knn_graph_torch = knn_graph   # the name rounds 1-2 used (callers in tests/ and tools/)


def synthetic_codes_shard(seed, offsets, code_size, rank, world, list_owner=None):
    """The lists a rank owns (list_owner[c] == rank, default c % world == rank) of the device's synthetic corpus
    (same bytes as the unsharded stream): (global ids, codes, norm_codes) in increasing list order."""
    nc = len(offsets) - 1
    if list_owner is None:
        owned = np.arange(rank, nc, world)
    else:
        owned = np.nonzero(np.asarray(list_owner) == rank)[0]
    return synthetic_codes_lists(seed, offsets, code_size, owned)


def synthetic_codes_lists(seed, offsets, code_size, lists):
    """The given lists (increasing) of the device's synthetic corpus: (global vector index, codes, norm_codes)."""
    off = offsets.astype(np.int64)
    owned = np.asarray(lists, np.int64)
    sizes = off[owned + 1] - off[owned]
    n = int(sizes.sum())
    # global vector index of every owned vector
    starts = np.repeat(off[owned], sizes)
    within = np.arange(n, dtype=np.int64) - np.repeat(np.cumsum(sizes) - sizes, sizes)
    gidx = (starts + within).astype(np.uint64)
    with np.errstate(over="ignore"):
        wpc = code_size // 8 if code_size % 8 == 0 else 0
        if wpc:
            w = (gidx[:, None] * np.uint64(wpc) + np.arange(wpc, dtype=np.uint64)[None, :]).reshape(-1)
            v = _mix64(np.uint64(seed) + (w + np.uint64(1)) * GOLDEN)
            codes = v.view(np.uint8).reshape(n, code_size)
        else:  # code_size 4: half a word per code
            w = gidx // np.uint64(2)
            v = _mix64(np.uint64(seed) + (w + np.uint64(1)) * GOLDEN)
            sh = (gidx % np.uint64(2)) * np.uint64(32)
            codes = ((v >> sh) & np.uint64(0xffffffff)).astype(np.uint32).view(np.uint8).reshape(n, 4)
        nseed = np.uint64(seed) ^ NORM_SEED_XOR
        w = gidx // np.uint64(8)
        v = _mix64(nseed + (w + np.uint64(1)) * GOLDEN)
        norm_codes = ((v >> ((gidx % np.uint64(8)) * np.uint64(8))) & np.uint64(0xff)).astype(np.uint8)
    return gidx.astype(np.uint32), np.ascontiguousarray(codes), norm_codes


def synthetic_codes_sparse(seed, offsets, code_size, lists=None, into=None):
    """A full-size host view of the device's synthetic corpus in which only `lists` hold their real bytes -- for an
    oracle sample at sizes (1B vectors = 21 GB) whose full host copy would take most of a minute.  The arrays come from
    calloc, so the lists never touched stay uncommitted zero pages.  `into` = a previous result to fill further."""
    n = int(offsets[-1])
    if into is None:
        into = (np.zeros(n, np.uint32), np.zeros((n, code_size), np.uint8), np.zeros(n, np.uint8))
    ids, codes, norm_codes = into
    if lists is not None and len(lists):
        lists = np.unique(np.asarray(lists, np.int64))
        for a in range(0, len(lists), 8192):  # bounded temporaries
            gidx, c, nc_ = synthetic_codes_lists(seed, offsets, code_size, lists[a:a + 8192])
            g = gidx.astype(np.int64)
            ids[g] = gidx
            codes[g] = c
            norm_codes[g] = nc_
    return into


def knn_ids_torch(centroids, k, device=None, chunk=None):
    """ids of the k nearest OTHER centroids of every centroid, ascending distance (the library's exact table)."""
    return knn_table(centroids, k, device)


def make_grouping_tables(seed, tb, nsubc, device=None):
    """Grouping tables for a throughput corpus (IndexIVF_HNSW_Grouping.h:17-22,61): the nsubc nearest centroids
    of every centroid, alpha in [0, 0.5], each list split at random into nsubc sub-groups, and the exact
    inter-centroid distances in the reference's float order (through the oracle)."""
    rng = np.random.default_rng(seed)
    nc = tb["nc"]
    nn = knn_ids_torch(tb["centroids"], nsubc, device=device)
    alphas = rng.uniform(0.0, 0.5, size=nc).astype(np.float32)
    sizes = np.diff(tb["offsets"].astype(np.int64))
    # multinomial split of every list into nsubc parts (vectorised: sorted uniform cut points)
    cuts = np.sort(rng.integers(0, sizes[:, None] + 1, size=(nc, nsubc - 1)), axis=1)
    edges = np.concatenate([np.zeros((nc, 1), np.int64), cuts, sizes[:, None]], axis=1)
    sg = np.diff(edges, axis=1).astype(np.uint32)
    assert (sg.sum(1) == sizes).all()
    g = orc.Hnsw.from_arrays(np.zeros(nc, np.uint8), np.zeros((nc, 1), np.uint32), tb["centroids"], 1)
    icd = g.inter_centroid_dists(nn)
    g.free()
    return dict(nsubc=nsubc, nn_centroid_idxs=nn, alphas=alphas, subgroup_sizes=sg, inter_centroid_dists=icd)


def rotated_vectors(vectors, A):
    """The quantizer's vectors after rotate_quantizer (IndexIVF_HNSW.cpp:789-800), in the reference's float order
    (through the oracle: synthetic-data preparation for the Grouping workloads, never part of what is measured)."""
    n = len(vectors)
    g = orc.Hnsw.from_arrays(np.zeros(n, np.uint8), np.zeros((n, 1), np.uint32), vectors, 1)
    g.rotate(A)
    out = g.vectors.copy()
    g.free()
    return out


def make_encode_case(seed, nc, d, M, opq, n, hnsw_M=8, kind="sift"):
    """Inputs of an add_batch parity case (construction side): centroids, their reference-identical graph, code
    books, optional OPQ matrix, base vectors, and the oracle index that encodes them."""
    from oracle import orc
    rng = np.random.default_rng(seed)
    if kind == "sift":
        cents = sift_like(rng, nc, d)
        x = (cents[rng.choice(nc, n)] + rng.normal(0, 15.0, size=(n, d))).astype(np.float32)
        cb = rng.normal(0.0, 9.0, size=(M, 256, d // M)).astype(np.float32)
        nt = np.sort(rng.normal(d * 2100.0, d * 300.0, size=256)).astype(np.float32)
    else:
        cents = rng.normal(0, 1, size=(nc, d))
        cents = (cents / np.linalg.norm(cents, axis=1, keepdims=True)).astype(np.float32)
        x = (cents[rng.choice(nc, n)] + rng.normal(0, 0.05, size=(n, d))).astype(np.float32)
        cb = rng.normal(0.0, 0.03, size=(M, 256, d // M)).astype(np.float32)
        nt = np.sort(rng.normal(1.0, 0.1, size=256)).astype(np.float32)
    graph = orc.Hnsw.build(cents, M=hnsw_M, efConstruction=60)
    A = random_rotation(rng, d) if opq else None
    ox = orc.Index(d, M, graph, cb, nt, np.zeros(nc + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, M), np.uint8),
                   np.zeros(0, np.uint8), np.zeros(nc, np.float32), opq_A=A)
    ox.set_params(1, 0, 40)
    return dict(cents=cents, x=x, cb=cb, nt=nt, A=A, graph=graph, ox=ox, d=d, M=M, nc=nc)
