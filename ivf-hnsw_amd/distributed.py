"""Multi-GPU search: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

SURVEY.md 8e: every rank holds the replicated tables (graph, centroid norms, code books, list offsets);
the inverted lists are sharded list-wise (list c lives on rank c % world).  The scan plan (probe order,
max_codes rule, pruning) is derived identically on every rank from the replicated tables, each rank
scores only the lists it owns, and ONE tiny exchange per query batch merges the per-shard winners:

    keys   : int64 MIN all-reduce of the packed (orderable distance << 32 | scan position) keys -- the scan
             position reproduces the reference's strict-'<' first-scanned-wins rule (IndexIVF_HNSW.cpp:285)
             across shards;
    labels : the owner of the winning scan position resolves the label, the others contribute -1, MAX
             all-reduce.

The coarse HNSW walk is split over the ranks by query and all-gathered, so no stage is replicated work.
Payload per 10 k-query batch: 2 x 80 KB (keys, labels) + 2.5 MB (coarse stage): latency-bound, one
collective each, never per query.
"""
import numpy as np

SIGN_FLIP = np.uint64(0x8000000000000000)
KEY_INIT = (np.uint64(0x7f7fffff | 0x80000000) << np.uint64(32))


def pack_keys(dist, vpos):
    """numpy mirror of the device key (device_common.h f32_orderable + kernels_search.hip scan): the signed
    int64 whose order is (distance, scan position).  Used by the CPU tests of the merge protocol."""
    d = np.ascontiguousarray(dist, np.float32) + np.float32(0.0)
    u = d.view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    o = np.where(neg, (~u) & np.uint64(0xffffffff), u | np.uint64(0x80000000))
    key = (o << np.uint64(32)) | np.ascontiguousarray(vpos, np.uint32).astype(np.uint64)
    return (key ^ SIGN_FLIP).view(np.int64)


def unpack_keys(skey):
    key = np.ascontiguousarray(skey, np.int64).view(np.uint64) ^ SIGN_FLIP
    o = (key >> np.uint64(32)).astype(np.uint32)
    vpos = (key & np.uint64(0xffffffff)).astype(np.uint32)
    u = np.where((o & np.uint32(0x80000000)) != 0, o & np.uint32(0x7fffffff), ~o)
    return u.astype(np.uint32).view(np.float32), vpos


def query_slice(nq, rank, world):
    """(lo, hi, per): the queries whose coarse stage `rank` computes; `per` rows per rank in the padded
    all-gather buffer."""
    per = (nq + world - 1) // world
    return min(rank * per, nq), min((rank + 1) * per, nq), per


def expected_list_load(list_sizes, link_counts, links):
    """Expected codes scored per query on every list, up to a constant: its size times how often it is probed.  A list
    is probed when its centroid is among the nearest of a query, and queries fall where the data is, so the in-degree
    of the centroid in the quantizer's own neighbour graph (the HNSW links the index already holds) estimates the
    probe frequency -- small-norm "hub" centroids of non-negative descriptors are probed several times as often as
    the average one, which a partition balanced by size alone turns into a 1.8x hot rank."""
    n = len(list_sizes)
    cnt = np.asarray(link_counts, np.int64)
    lk = np.asarray(links).reshape(n, -1)
    valid = np.arange(lk.shape[1])[None, :] < cnt[:, None]
    indeg = np.bincount(lk[valid].astype(np.int64), minlength=n).astype(np.float64)
    return np.asarray(list_sizes, np.float64) * (1.0 + indeg)


def partition_lists(centroids, list_sizes, world, method="spatial", load=None):
    """Owner table for list-wise sharding (ivfhnsw_ivf_desc.list_owner): rank of every inverted list.

    "spatial": recursive bisection of the centroids along their principal direction, each cut placed so that both
    sides carry the same LOAD per rank -- `load` (expected_list_load: size x probe frequency) when given, else the
    number of codes (any world size: a node of w ranks splits w//2 : w - w//2).  Lists
    whose centroids are close end up on the same rank, so a query's probes -- the centroids nearest to it -- fall
    on few ranks; only those build and stage the query's table (the plan of every other rank is empty for it).
    How much that buys depends on the data: clustered descriptors (SIFT, DEEP) concentrate a query's probes,
    iid-Gaussian synthetic centroids in 128 dimensions hardly do (DESIGN.md 7 has the measured figures).
    "mod": c % world, the layout of the C ABI's default."""
    nc = len(list_sizes)
    if world == 1:
        return np.zeros(nc, np.uint32)
    if method == "mod":
        return (np.arange(nc, dtype=np.uint64) % np.uint64(world)).astype(np.uint32)
    if method != "spatial":
        raise ValueError("unknown partition method %r" % (method,))
    x_all = np.ascontiguousarray(centroids, np.float32)
    sizes = np.asarray(list_sizes if load is None else load, np.float64)
    owner = np.zeros(nc, np.uint32)
    work = [(np.arange(nc, dtype=np.int64), 0, world)]
    while work:
        idx, r0, w = work.pop()
        if w == 1 or len(idx) == 0:
            owner[idx] = r0
            continue
        wl = w // 2
        x = x_all[idx].astype(np.float64)
        x -= x.mean(0)
        v = np.ones(x.shape[1]) / np.sqrt(x.shape[1])     # power iteration from a fixed start: deterministic
        for _ in range(12):
            v = x.T @ (x @ v)
            n = np.linalg.norm(v)
            if n == 0:
                v = np.ones(x.shape[1]) / np.sqrt(x.shape[1])
                break
            v /= n
        order = np.argsort(x @ v, kind="stable")
        cs = np.cumsum(sizes[idx][order] + 1e-9)          # the epsilon orders empty lists too
        cut = int(np.searchsorted(cs, cs[-1] * wl / w))
        cut = min(max(cut, 1), len(idx) - 1) if len(idx) > 1 else 0
        work.append((idx[order[:cut]], r0, wl))
        work.append((idx[order[cut:]], r0 + wl, w - wl))
    return owner


def _all_gather_rows(buf, rank, per, group):
    """All-gather of `per` rows per rank into buf (RCCL: ncclAllGather).  The rank's own rows are sent from a
    copy: whether an input that aliases the output is accepted is the backend's business, and the copy is 1.3 MB."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(buf, buf[rank * per:(rank + 1) * per].clone(), group=group)
    else:  # gloo (CPU tests / single-GPU rehearsal): stage through the host
        host = buf.cpu()
        parts = [host[r * per:(r + 1) * per].clone() for r in range(dist.get_world_size(group))]
        dist.all_gather(parts, host[rank * per:(rank + 1) * per].clone(), group=group)
        for r, t in enumerate(parts):
            buf[r * per:(r + 1) * per].copy_(t)


def _all_reduce(t, op, group):
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=op, group=group)
    else:
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)


def _all_gather_stack(t, group):
    """[world, *t.shape]: every rank's tensor (RCCL: ncclAllGather; gloo: staged through the host)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return out
    host = t.cpu().contiguous()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return torch.stack(parts).to(t.device)


def merge_topk_keys(stacked_keys, k):
    """[world, nq, k] signed keys of the shards -> [nq, k], the k smallest per query in ascending order: the
    reference's result SET (smallest distance, earliest scan position on ties); its order is merge_heap_order's."""
    import torch
    world, nq, kk = stacked_keys.shape
    allk = stacked_keys.permute(1, 0, 2).reshape(nq, world * kk)
    return torch.sort(allk, dim=1).values[:, :k].contiguous()


def merge_streams(streams, lens, cap):
    """Candidate streams of the shards ([world, nq, L] unsigned keys held in int64, [world, nq] lengths) -> one
    stream per query in GLOBAL scan order ([nq, world * L], [nq] lengths).  Scan positions are unique across shards
    (a position belongs to exactly one list, a list to one shard), so sorting by position interleaves the shards
    exactly as the unsharded scan meets the codes."""
    import torch
    world, nq, L = streams.shape
    idx = torch.arange(L, device=streams.device).view(1, 1, L)
    valid = idx < lens.view(world, nq, 1).to(torch.int64)
    pos = streams & 0xffffffff                                   # low word = scan position
    pos = torch.where(valid, pos, torch.full_like(pos, 1 << 40))  # padding after everything real
    pos = pos.permute(1, 0, 2).reshape(nq, world * L)
    keys = streams.permute(1, 0, 2).reshape(nq, world * L)
    order = torch.argsort(pos, dim=1)
    merged = torch.gather(keys, 1, order).contiguous()
    total = lens.to(torch.int64).sum(0).to(torch.int32).contiguous()
    return merged, total


class ShardedSearcher:
    """Drives one GpuIndex shard per rank.  All tensors are torch CUDA tensors owned by the caller.

    Stream contract: the collectives run on torch's current stream, the shard's kernels on the handle's stream, and
    nothing else orders the two -- so the constructor BINDS the handle to torch's current stream of `device`
    (ivfhnsw_gpu_set_stream) and step() checks that it is still the current one.  A caller that wants another stream
    enters `torch.cuda.stream(s)` before constructing the searcher and around every step."""

    def __init__(self, gpu_index, rank, world, nq, nprobe, device, group=None, k=1, force_collectives=False,
                 stream_cap=None, split=None, split_min=8192):
        import torch
        self.g, self.rank, self.world, self.nq, self.nprobe, self.group = gpu_index, rank, world, nq, nprobe, group
        self.k = k
        self.device = torch.device(device)
        self.stream_cap = stream_cap  # tests: a smaller candidate-stream capacity than the library's (overflow path)
        self.bound_stream = None
        if self.device.type == "cuda":
            self.bound_stream = torch.cuda.current_stream(self.device).cuda_stream
            gpu_index.set_stream(self.bound_stream)
        # run every collective even in a world of one (a single-rank RCCL group accepts them all): lets a one-GPU box
        # execute the very calls the 8-GPU node makes (tests/test_gpu_rccl_single_rank.py)
        self.collectives = world > 1 or force_collectives
        self.lo, self.hi, self.per = query_slice(nq, rank, world)
        self.cid = torch.empty((self.per * world, nprobe), dtype=torch.int32, device=device)
        self.cd = torch.empty((self.per * world, nprobe), dtype=torch.float32, device=device)
        self.keys = torch.empty((nq, k), dtype=torch.int64, device=device)
        self.xrot = None  # own slice of the batch, OPQ-rotated for the coarse walk (allocated on first use)
        # Two overlapping parts (round 3): what the one-GPU call does inside ivfhnsw_gpu_search_dev -- the scan of the
        # first part beside the walk of the second -- for the sharded step, where the library cannot do it (the coarse
        # results cross the ranks in between).  The rank's slice of `per` queries is cut n1 | n2; part p's scan covers the
        # world * n_p queries every rank walked in its part p; part 2 runs on a VIEW of the handle and a side stream.
        # k = 1 only, equal slices only.  Measured per rank (tools/rank_emulation.py --split, 1B shape, collectives
        # aside): 2 shards 1.92 -> 1.81 ms, 4 shards 2.06 -> 1.94, 8 shards 2.22 -> 2.07 (600 permille, two-kernel scan).
        self.parts = None
        if split is None:
            split = 780 if world < 8 else 600
        if (split and k == 1 and self.device.type == "cuda" and self.collectives and nq % world == 0 and self.per >= split_min
                and hasattr(gpu_index, "view")):
            per = self.per
            n2 = ((per * (1000 - split) // 1000 + 1024) // 2048) * 2048
            n2 = max(min(2048, per // 2), min(n2, per // 2))
            n1 = per - n2
            self.view = gpu_index.view()
            self.side = torch.cuda.Stream(device=self.device)
            self.view.set_stream(self.side.cuda_stream)
            for h in (gpu_index, self.view):   # the pipelined table + scan kernel holds most of a CU's LDS: not beside a walk
                h.set_option("scan_pipe", 0)
            ar = torch.arange(world, device=device).view(world, 1) * per
            self.parts = []
            for h, n, off in ((gpu_index, n1, 0), (self.view, n2, n1)):
                idx = (ar + off + torch.arange(n, device=device).view(1, n)).reshape(-1)   # rank-major rows of this part
                self.parts.append(dict(h=h, n=n, off=off, idx=idx,
                                       cid=torch.empty((n * world, nprobe), dtype=torch.int32, device=device),
                                       cd=torch.empty((n * world, nprobe), dtype=torch.float32, device=device),
                                       xrot=None))
            self.keys_all = torch.empty((nq, 1), dtype=torch.int64, device=device)
            self.dist_all = torch.empty((nq, 1), dtype=torch.float32, device=device)
            self.lab_all = torch.empty((nq, 1), dtype=torch.int64, device=device)
            self.idx_all = torch.cat([p["idx"] for p in self.parts])
            a1 = n1 * world
            for p, sl in zip(self.parts, (slice(0, a1), slice(a1, nq))):
                p["keys"], p["dist"], p["lab"] = self.keys_all[sl], self.dist_all[sl], self.lab_all[sl]

    def last_scan_counts(self):
        """(codes, segments) this rank scored in the last step (both parts of a two-part step)."""
        a, b = self.g.last_scan_counts()
        if self.parts is not None:
            a2, b2 = self.view.last_scan_counts()
            a, b = a + a2, b + b2
        return a, b

    def close(self):
        """Release the view the two-part step runs its second part on (the shard's own handle stays the caller's)."""
        if self.parts is not None:
            self.view.close()
            self.parts = None

    def handles(self):
        """The shard's handle and, in two-part mode, the view its second part runs on."""
        return [self.g] + ([self.view] if self.parts is not None else [])

    def coarse(self, d_q, efSearch):
        """This rank's slice of the coarse stage, then the all-gather: self.cid / self.cd hold the whole batch."""
        import torch
        g, r, per = self.g, self.rank, self.per
        if self.hi > self.lo:
            if self.xrot is None:
                self.xrot = torch.empty((per, d_q.shape[1]), dtype=torch.float32, device=d_q.device)
            g.rotate_dev(self.hi - self.lo, d_q[self.lo:self.hi], self.xrot)
            g.coarse_dev(self.hi - self.lo, self.xrot, self.nprobe, efSearch,
                         self.cid[r * per:], self.cd[r * per:])
        if self.collectives:
            _all_gather_rows(self.cid, r, per, self.group)
            _all_gather_rows(self.cd, r, per, self.group)

    def step(self, d_q, d_dist, d_lab, max_codes, efSearch, do_pruning=False, heap_order=False):
        """One batch: coarse slice -> all-gather -> scan own shard -> merge keys -> resolve -> MAX labels.
        With OPQ the walk runs on the rotated slice (IndexIVF_HNSW.cpp:240,248); search_dev rotates the whole batch
        again for its tables, from the unrotated d_q.

        k = 1: ONE int64 MIN all-reduce of the packed keys.  k > 1: all-gather of the shards' nq*k keys and a local
        k-way merge (ascending), or -- heap_order, the array the reference's faiss heap leaves,
        IndexIVF_HNSW.cpp:285-288 -- all-gather of the shards' candidate streams, merged in scan order and replayed."""
        import torch
        import torch.distributed as dist
        g, k, nq = self.g, self.k, self.nq
        if self.bound_stream is not None and torch.cuda.current_stream(self.device).cuda_stream != self.bound_stream:
            raise RuntimeError("ShardedSearcher.step on another torch stream than the one its handle is bound to: the "
                               "collectives would not be ordered behind the shard's kernels")
        if self.parts is not None and not heap_order:
            return self._step_two_parts(d_q, d_dist, d_lab, max_codes, efSearch, do_pruning)
        self.coarse(d_q, efSearch)
        heap = heap_order and k > 1
        g.search_dev(nq, k, d_q, d_dist, d_lab, self.nprobe, max_codes, d_coarse_ids=self.cid,
                     d_coarse_dists=self.cd, do_pruning=do_pruning, d_out_keys=self.keys, heap_order=heap)
        if not self.collectives and not heap:
            return
        if k == 1:
            _all_reduce(self.keys, dist.ReduceOp.MIN, self.group)
        elif not heap:
            self.keys.copy_(merge_topk_keys(_all_gather_stack(self.keys, self.group), k))
        else:
            lens = torch.empty((nq,), dtype=torch.int32, device=d_q.device)
            cap = g.last_stream_dev(nq, d_len=lens)
            if self.stream_cap is not None:
                cap = min(cap, self.stream_cap)
            g.sync()  # torch reads below: order them after the handle's stream
            lmax = lens.max().to(torch.int64).view(1)
            # MAX over the shards BEFORE the overflow test: every rank must take the same branch, or the ranks that do
            # not overflow would wait in the all-gathers below for one that has left the step
            if self.collectives:
                _all_reduce(lmax, dist.ReduceOp.MAX, self.group)
            if int(lmax.item()) > cap:
                raise RuntimeError("candidate stream of a query exceeded %d entries on some shard: use ascending "
                                   "order" % cap)
            L = max(1, int(lmax.item()))
            mine = torch.empty((nq, L), dtype=torch.int64, device=d_q.device)
            g.last_stream_dev(nq, L, d_keys=mine)
            g.sync()
            if self.collectives:
                streams = _all_gather_stack(mine, self.group)
                all_lens = _all_gather_stack(lens, self.group)
            else:
                streams, all_lens = mine.unsqueeze(0), lens.unsqueeze(0)
            merged, total = merge_streams(streams, all_lens, cap)
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            g.replay_stream_dev(nq, k, merged, total, merged.shape[1], self.keys)
        g.resolve_keys_dev(nq, k, self.keys, d_dist, d_lab)
        if self.collectives:
            _all_reduce(d_lab, dist.ReduceOp.MAX, self.group)

    def _step_two_parts(self, d_q, d_dist, d_lab, max_codes, efSearch, do_pruning):
        """The k = 1 step as two overlapping parts (see __init__).  Collectives are issued in one order on every rank:
        part 1's all-gathers, part 2's, one MIN over all keys, one MAX over all labels."""
        import torch
        import torch.distributed as dist
        S = torch.cuda.current_stream(self.device)
        r, world = self.rank, self.world
        fork = torch.cuda.Event()
        fork.record(S)
        self.side.wait_event(fork)
        for p, stream in zip(self.parts, (S, self.side)):
            with torch.cuda.stream(stream):
                h, n = p["h"], p["n"]
                q_p = d_q.index_select(0, p["idx"])                       # this part's queries of every rank, rank-major
                if p["xrot"] is None:
                    p["xrot"] = torch.empty((n, d_q.shape[1]), dtype=torch.float32, device=self.device)
                own = q_p[r * n:(r + 1) * n]
                h.rotate_dev(n, own, p["xrot"])
                h.coarse_dev(n, p["xrot"], self.nprobe, efSearch, p["cid"][r * n:], p["cd"][r * n:])
                _all_gather_rows(p["cid"], r, n, self.group)
                _all_gather_rows(p["cd"], r, n, self.group)
                h.search_dev(n * world, 1, q_p, p["dist"], p["lab"], self.nprobe, max_codes, d_coarse_ids=p["cid"],
                             d_coarse_dists=p["cd"], do_pruning=do_pruning, d_out_keys=p["keys"])
                p["q"] = q_p   # keep the gathered queries alive until the streams have met again
        join = torch.cuda.Event()
        join.record(self.side)
        S.wait_event(join)
        _all_reduce(self.keys_all, dist.ReduceOp.MIN, self.group)
        merged = torch.cuda.Event()
        merged.record(S)
        self.side.wait_event(merged)
        for p, stream in zip(self.parts, (S, self.side)):
            with torch.cuda.stream(stream):
                p["h"].resolve_keys_dev(p["n"] * world, 1, p["keys"], p["dist"], p["lab"])
        join2 = torch.cuda.Event()
        join2.record(self.side)
        S.wait_event(join2)
        _all_reduce(self.lab_all, dist.ReduceOp.MAX, self.group)
        d_dist.index_copy_(0, self.idx_all, self.dist_all)
        d_lab.index_copy_(0, self.idx_all, self.lab_all)
