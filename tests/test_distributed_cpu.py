"""CPU test of the multi-GPU exchange step with real torch.distributed ranks (gloo, world 2/3).

Each rank emulates its shard in numpy (the scan plan of IndexIVF_HNSW.cpp:267-292 with global scan positions,
ADC in the oracle's float order, only lists c % world == rank), then the ranks run exactly the collectives the
GPU path runs (ivf-hnsw_amd/distributed.py): int64 MIN all-reduce of the packed keys, MAX all-reduce of the
owner's labels.  The merged result must equal the unsharded oracle bit for bit -- including which of several
equal distances wins (first scanned).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_best(c, q, cid, cd, max_codes, rank, world, pack_keys):
    """Best (key, label) over the lists owned by `rank`, numpy restatement of plan + scan."""
    from oracle import orc
    F = np.float32
    off = c["offsets"].astype(np.int64)
    M = c["code_size"]
    nq = len(q)
    keys = np.empty(nq, np.int64)
    labels = np.full(nq, -1, np.int64)
    for i in range(nq):
        tab = orc.inner_prod_table(q[i], c["pq_centroids"], M)
        ncode = 0
        best_d, best_v, best_l = None, None, -1
        for p, cc in enumerate(cid[i]):
            cc = int(cc)
            n = off[cc + 1] - off[cc]
            if n == 0:
                continue
            if cc % world == rank:
                codes = c["codes"][off[cc]:off[cc + 1]]
                s = np.zeros(n, F)
                for m in range(M):
                    s = (s + tab[m, codes[:, m]]).astype(F)
                term1 = F(cd[i, p] - c["centroid_norms"][cc])
                dist = ((term1 + c["norm_table"][c["norm_codes"][off[cc]:off[cc + 1]]]).astype(F) - F(2) * s).astype(F)
                j = int(np.argmin(dist))  # first minimum = first scanned
                if best_d is None or dist[j] < best_d:
                    best_d, best_v, best_l = dist[j], ncode + j, int(c["ids"][off[cc] + j])
            ncode += n
            if ncode >= max_codes:
                break
        if best_d is None:
            keys[i] = pack_keys(np.array([np.finfo(F).max], F), np.array([0], np.uint32))[0]
        else:
            keys[i] = pack_keys(np.array([best_d], F), np.array([best_v], np.uint32))[0]
            labels[i] = best_l
    return keys, labels


def _worker(rank, world, port, ties, out_dir, shards=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import importlib
    import synth
    ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # replica groups x list shards (bench.py --list-shards): rank r is shard r % S of group r // S, every group
    # serves its own queries and its collectives stay inside the group
    S = shards or world
    group, gidx, job_rank = None, rank // S, rank
    if S != world:
        for gi in range(world // S):
            pg = dist.new_group(ranks=list(range(gi * S, (gi + 1) * S)))
            if gi == gidx:
                group = pg
    rank, world = rank % S, S
    c = synth.make_corpus(seed=51, nc=96, d=64, M=8, n_base=6000, nq=24, efConstruction=60)
    if gidx:
        c["queries"] = np.ascontiguousarray(c["queries"][::-1] + np.float32(gidx))
    if ties:  # every code identical: all distances inside a list tie, across lists they differ by term1
        c["codes"] = np.zeros_like(c["codes"])
        c["norm_codes"] = np.zeros_like(c["norm_codes"])
    nprobe, max_codes, ef = 12, 700, 32
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)

    # the coarse stage is split over the ranks by query and all-gathered, as on the GPUs
    lo, hi, per = D.query_slice(len(cid), rank, world)
    cid_pad = torch.zeros((per * world, nprobe), dtype=torch.int32)
    cd_pad = torch.zeros((per * world, nprobe), dtype=torch.float32)
    cid_pad[rank * per:rank * per + (hi - lo)] = torch.from_numpy(cid[lo:hi].astype(np.int32))
    cd_pad[rank * per:rank * per + (hi - lo)] = torch.from_numpy(cd[lo:hi])
    dist.all_gather_into_tensor(cid_pad, cid_pad[rank * per:(rank + 1) * per].clone(), group=group)
    dist.all_gather_into_tensor(cd_pad, cd_pad[rank * per:(rank + 1) * per].clone(), group=group)
    g_cid = cid_pad.numpy()[:len(cid)].astype(np.uint32)
    g_cd = cd_pad.numpy()[:len(cid)]
    assert np.array_equal(g_cid, cid) and np.array_equal(g_cd, cd)

    keys, labels = _shard_best(c, c["queries"], g_cid, g_cd, max_codes, rank, world, D.pack_keys)
    tk = torch.from_numpy(keys.copy())
    dist.all_reduce(tk, op=dist.ReduceOp.MIN, group=group)
    merged = tk.numpy()
    mine = merged == keys
    tl = torch.from_numpy(np.where(mine, labels, -1))
    dist.all_reduce(tl, op=dist.ReduceOp.MAX, group=group)
    dd, vv = D.unpack_keys(merged)
    ok = np.array_equal(tl.numpy(), ref_l[:, 0]) and np.array_equal(dd.view(np.uint32), ref_d[:, 0].view(np.uint32))
    open(os.path.join(out_dir, "rank%d.%s" % (job_rank, "ok" if ok else "fail")), "w").write(
        "%s\n%s\n" % (tl.numpy().tolist(), ref_l[:, 0].tolist()))
    dist.destroy_process_group()


def _shard_stream(c, q, cid, cd, max_codes, rank, owner, pack_keys):
    """Every code of the lists `rank` owns, in scan order, as unsigned (orderable distance << 32 | scan position) keys
    per query -- the trivially complete candidate stream -- plus a position -> label map."""
    from oracle import orc
    F = np.float32
    off = c["offsets"].astype(np.int64)
    M = c["code_size"]
    streams, labels = [], []
    for i in range(len(q)):
        tab = orc.inner_prod_table(q[i], c["pq_centroids"], M)
        ncode, ks, lab = 0, [], {}
        for p, cc in enumerate(cid[i]):
            cc = int(cc)
            n = off[cc + 1] - off[cc]
            if n == 0:
                continue
            if owner[cc] == rank:
                codes = c["codes"][off[cc]:off[cc + 1]]
                s = np.zeros(n, F)
                for m in range(M):
                    s = (s + tab[m, codes[:, m]]).astype(F)
                term1 = F(cd[i, p] - c["centroid_norms"][cc])
                dist = ((term1 + c["norm_table"][c["norm_codes"][off[cc]:off[cc + 1]]]).astype(F) - F(2) * s).astype(F)
                vpos = (ncode + np.arange(n)).astype(np.uint32)
                ks.append(pack_keys(dist, vpos).view(np.uint64) ^ np.uint64(0x8000000000000000))
                for j in range(n):
                    lab[int(vpos[j])] = int(c["ids"][off[cc] + j])
            ncode += n
            if ncode >= max_codes:
                break
        streams.append(np.concatenate(ks) if ks else np.zeros(0, np.uint64))
        labels.append(lab)
    return streams, labels


def _worker_topk(rank, world, port, out_dir):
    """k = 10 over gloo ranks with a spatial owner table: all-gather of the shards' k best keys -> k-way merge
    (ascending); all-gather of the candidate streams -> merged in scan order -> faiss heap replay (heap-array order,
    IndexIVF_HNSW.cpp:285-288).  Both against the unsharded oracle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import importlib
    import synth
    from oracle import orc
    ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = synth.make_corpus(seed=52, nc=96, d=64, M=8, n_base=6000, nq=20, efConstruction=60)
    nprobe, max_codes, ef, k = 12, 700, 32, 10
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=k)
    nq = len(ref_l)
    owner = D.partition_lists(c["centroids"], np.diff(c["offsets"].astype(np.int64)), world, "spatial")
    assert set(owner.tolist()) == set(range(world))
    streams, labels = _shard_stream(c, c["queries"], cid, cd, max_codes, rank, owner, D.pack_keys)
    SIGN = np.uint64(0x8000000000000000)
    init = np.uint64((0x7f7fffff | 0x80000000) << 32)
    # local top-k, ascending, as signed keys (what search_dev leaves in out_keys)
    loc = np.full((nq, k), init, np.uint64)
    for i, st in enumerate(streams):
        srt = np.sort(st)[:k]
        loc[i, :len(srt)] = srt
    tk = torch.from_numpy((loc ^ SIGN).view(np.int64).copy())
    merged = D.merge_topk_keys(D._all_gather_stack(tk, None), k).numpy().view(np.uint64) ^ SIGN
    # streams, padded to the longest of any rank
    lens = torch.tensor([len(st) for st in streams], dtype=torch.int32)
    lmax = lens.max().to(torch.int64).view(1)
    dist.all_reduce(lmax, op=dist.ReduceOp.MAX)
    L = max(1, int(lmax.item()))
    pad = np.zeros((nq, L), np.uint64)
    for i, st in enumerate(streams):
        pad[i, :len(st)] = st
    ms, total = D.merge_streams(D._all_gather_stack(torch.from_numpy(pad.view(np.int64).copy()), None),
                                D._all_gather_stack(lens, None), 1 << 20)
    ms = ms.numpy().view(np.uint64)
    all_labels = [None] * world
    dist.all_gather_object(all_labels, labels)
    ok = True
    for i in range(nq):
        lab = {}
        for r in range(world):
            lab.update(all_labels[r][i])
        # ascending merge == the reference's set
        got = sorted(lab[int(v & np.uint64(0xffffffff))] for v in merged[i] if v < init)
        ok &= got == sorted(int(x) for x in ref_l[i] if x >= 0)
        # heap replay over the merged stream == the reference's heap array
        hv = np.empty(k, np.float32)
        hl = np.empty(k, np.int64)
        orc.lib().orc_maxheap_heapify(k, orc._p(hv), orc._p(hl))
        seq = ms[i, :int(total[i])]
        assert (np.diff((seq & np.uint64(0xffffffff)).astype(np.int64)) > 0).all()   # global scan order
        dd, vv = D.unpack_keys((seq ^ SIGN).view(np.int64))
        for dj, vj in zip(dd, vv):
            if dj < hv[0]:
                orc.lib().orc_maxheap_pop(k, orc._p(hv), orc._p(hl))
                orc.lib().orc_maxheap_push(k, orc._p(hv), orc._p(hl), float(dj), lab[int(vj)])
        ok &= np.array_equal(hl, ref_l[i]) and np.array_equal(hv.view(np.uint32), ref_d[i].view(np.uint32))
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "fail")), "w").write("%s\n" % ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_topk_merge_equals_unsharded_oracle(tmp_path, world):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000) + world * 11
    mp.spawn(_worker_topk, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("rank%d.ok" % r)), "rank %d: merged top-k differs from the oracle" % r


@pytest.mark.parametrize("world,ties", [(2, False), (2, True), (3, False)])
def test_shard_merge_equals_unsharded_oracle(tmp_path, world, ties):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + world * 7 + (1 if ties else 0)
    mp.spawn(_worker, args=(world, port, ties, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("rank%d.ok" % r)), open(tmp_path / ("rank%d.fail" % r)).read()


def test_replica_groups_of_list_shards(tmp_path):
    """4 ranks as 2 replica groups x 2 list shards (bench.py --list-shards 2): each group merges its own batch
    inside its own process group and gets the unsharded oracle's result for ITS queries."""
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(4, port, False, str(tmp_path), 2), nprocs=4, join=True)
    for r in range(4):
        assert os.path.exists(os.path.join(str(tmp_path), "rank%d.ok" % r)), os.listdir(str(tmp_path))


def _worker_class(rank, world, port, out_dir, tiny_cap):
    """The REAL ShardedSearcher (ivf-hnsw_amd/distributed.py) over gloo ranks, each driving a CPU stand-in shard
    (tests/fake_shard.py): k = 1 (MIN / MAX all-reduce), k = 10 ascending and in faiss heap-array order, spatial owner
    table, against the unsharded oracle.  tiny_cap: one rank's candidate streams overflow a forced capacity -- every
    rank must raise (none may be left waiting in a collective)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import importlib
    import synth
    from fake_shard import FakeShard
    ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    c = synth.make_corpus(seed=53, nc=96, d=64, M=8, n_base=6000, nq=21, efConstruction=60)
    nprobe, max_codes, ef = 12, 700, 32
    ox = synth.oracle_index(c)
    sizes = np.diff(c["offsets"].astype(np.int64))
    owner = D.partition_lists(c["centroids"], sizes, world, "spatial")
    nq = len(c["queries"])
    cpu = torch.device("cpu")
    d_q = torch.from_numpy(c["queries"].copy())
    ok, note = True, ""
    if tiny_cap:
        # shard 0 owns everything but one short list: only ITS streams pass the (forced, common) capacity of 64
        owner = np.zeros(c["nc"], np.uint32)
        small = np.where((sizes > 0) & (sizes <= 64))[0]
        owner[int(small[0])] = 1
        sh = FakeShard(c, ox, rank, owner, D.pack_keys, D.unpack_keys)
        s = D.ShardedSearcher(sh, rank, world, nq, nprobe, cpu, k=10, stream_cap=64)
        dd = torch.empty((nq, 10), dtype=torch.float32)
        ll = torch.empty((nq, 10), dtype=torch.int64)
        try:
            s.step(d_q, dd, ll, max_codes, ef, heap_order=True)
            ok, note = False, "no overflow raised"
        except RuntimeError as e:
            ok, note = "exceeded" in str(e), str(e)
        dist.barrier()   # reached by every rank only if none hangs in the step's collectives
    else:
        for k, heap in ((1, False), (10, False), (10, True)):
            ox.set_params(nprobe, max_codes, ef)
            ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=k)
            sh = FakeShard(c, ox, rank, owner, D.pack_keys, D.unpack_keys)
            s = D.ShardedSearcher(sh, rank, world, nq, nprobe, cpu, k=k)
            dd = torch.empty((nq, k), dtype=torch.float32)
            ll = torch.empty((nq, k), dtype=torch.int64)
            s.step(d_q, dd, ll, max_codes, ef, heap_order=heap)
            lab, dis = ll.numpy(), dd.numpy()
            if k == 1 or heap:
                good = np.array_equal(lab, ref_l) and np.array_equal(dis.view(np.uint32), ref_d.view(np.uint32))
            else:
                good = np.array_equal(np.sort(lab, 1), np.sort(ref_l, 1)) and bool((np.diff(dis, axis=1) >= 0).all())
            ok &= bool(good)
            note += "k=%d heap=%s %s; " % (k, heap, good)
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "fail")), "w").write(note + "\n")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_searcher_class_over_gloo_ranks(tmp_path, world):
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000) + world * 13
    mp.spawn(_worker_class, args=(world, port, str(tmp_path), False), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("rank%d.ok" % r)), open(tmp_path / ("rank%d.fail" % r)).read()


def test_stream_overflow_on_one_shard_raises_on_every_rank(tmp_path):
    """a rank-local overflow test before the MAX all-reduce left the other ranks in the all-gathers."""
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_worker_class, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    for r in range(2):
        assert os.path.exists(tmp_path / ("rank%d.ok" % r)), open(tmp_path / ("rank%d.fail" % r)).read()


def test_key_packing_orders_like_distance_then_position():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import importlib
    ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")
    rng = np.random.default_rng(0)
    d = rng.normal(0, 1e4, 2000).astype(np.float32)
    d[:50] = d[50:100]                      # exact ties
    d[100] = 0.0
    d[101] = -0.0
    v = rng.permutation(2000).astype(np.uint32)
    k = D.pack_keys(d, v)
    order = np.argsort(k, kind="stable")
    want = np.lexsort((v, d + np.float32(0)))
    assert np.array_equal(order, want)
    dd, vv = D.unpack_keys(k)
    assert np.array_equal(vv, v) and np.array_equal(dd, d + np.float32(0))
    # FLT_MAX / "nothing found" sorts after every finite distance
    init = D.pack_keys(np.array([np.finfo(np.float32).max], np.float32), np.array([0], np.uint32))[0]
    assert (k < init).all()
