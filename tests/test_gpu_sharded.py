"""GPU parity of the sharded path on ONE GPU: the index is split list-wise into `world` shard
handles, each scans only its lists against the shared global plan, the packed keys are MIN-merged (here with
torch.minimum, on the 8-GPU node with an RCCL all-reduce: ivf-hnsw_amd/distributed.py) and each shard resolves the
labels it owns.  Must equal the unsharded device result and the oracle, bit for bit, for IVFADC and Grouping."""
import numpy as np
import pytest

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu


def _owner_table(pkg, c, world, partition):
    """list -> rank: the C ABI's default (c % world) or the balanced spatial partition of distributed.py."""
    import importlib
    if partition == "mod":
        return None
    D = importlib.import_module("ivfhnsw_amd.distributed")
    return D.partition_lists(c["centroids"], np.diff(c["offsets"].astype(np.int64)), world, "spatial")


def _shard_arrays(c, rank, world, owner=None):
    off = c["offsets"].astype(np.int64)
    owned = [cc for cc in range(len(off) - 1) if (cc % world if owner is None else owner[cc]) == rank]
    sel = np.concatenate([np.arange(off[cc], off[cc + 1]) for cc in owned]) if owned else np.zeros(0, np.int64)
    return c["ids"][sel], c["codes"][sel], c["norm_codes"][sel]


@pytest.mark.parametrize("world,partition", [(2, "mod"), (3, "spatial"), (8, "spatial"), (8, "mod")])
@pytest.mark.parametrize("grouping", [False, True])
def test_sharded_equals_unsharded(gpu, pkg, world, partition, grouping):
    import torch
    if grouping:
        c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    else:
        c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    nprobe, max_codes, ef = 16, 2500, 40
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
    ref_d, ref_l, cid, cd, st = ox.search_batch(c["queries"], k=1)
    nq = len(ref_l)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(c["queries"]).to(dev)
    d_cid = torch.from_numpy(cid.astype(np.int32)).to(dev)
    d_cd = torch.from_numpy(cd).to(dev)
    gr = c["graph"]
    shards, keys, total_codes = [], [], 0
    owner = _owner_table(pkg, c, world, partition)
    for r in range(world):
        g = gpu()
        ids, codes, ncodes = _shard_arrays(c, r, world, owner)
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], ids, codes, ncodes, c["centroid_norms"], c["pq_centroids"],
                     c["norm_table"], shard_rank=r, shard_world=world, list_owner=owner)
        if grouping:
            g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                              c["inter_centroid_dists"])
            g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
        dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        g.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=d_cid, d_coarse_dists=d_cd,
                     do_pruning=grouping, d_out_keys=kk)
        g.sync()
        total_codes += g.last_scan_counts()[0]
        shards.append((g, dd, ll))
        keys.append(kk)
    assert total_codes == st.ncode  # the shards partition the reference's scanned codes exactly
    merged = keys[0].clone()
    for kk in keys[1:]:
        merged = torch.minimum(merged, kk)
    label = torch.full((nq, 1), -1, dtype=torch.int64, device=dev)
    for g, dd, ll in shards:
        g.resolve_keys_dev(nq, 1, merged, dd, ll)
        g.sync()
        label = torch.maximum(label, ll)
        dist = dd
    assert np.array_equal(label.cpu().numpy(), ref_l)
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.parametrize("world", [2, 5])
@pytest.mark.parametrize("grouping", [False, True])
def test_sharded_topk_merge(gpu, pkg, world, grouping):
    """k = 10 across shards: the k smallest of the all-gathered shard keys (ascending), and the reference's
    heap ARRAY (IndexIVF_HNSW.cpp:285-288) from the shards' candidate streams merged in scan order and replayed --
    the same functions ShardedSearcher.step runs behind its collectives, here on shard handles of one GPU."""
    import importlib
    import torch
    D = importlib.import_module("ivfhnsw_amd.distributed")
    if grouping:
        c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    else:
        c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    nprobe, max_codes, ef, k = 16, 2500, 40, 10
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=k)     # faiss heap-array order
    nq = len(ref_l)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(c["queries"]).to(dev)
    d_cid = torch.from_numpy(cid.astype(np.int32)).to(dev)
    d_cd = torch.from_numpy(cd).to(dev)
    owner = _owner_table(pkg, c, world, "spatial")
    gr = c["graph"]
    shards, keys, streams, lens = [], [], [], []
    for r in range(world):
        g = gpu()
        ids, codes, ncodes = _shard_arrays(c, r, world, owner)
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], ids, codes, ncodes, c["centroid_norms"], c["pq_centroids"],
                     c["norm_table"], shard_rank=r, shard_world=world, list_owner=owner)
        if grouping:
            g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                              c["inter_centroid_dists"])
            g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
        dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, k), dtype=torch.int64, device=dev)
        kk = torch.empty((nq, k), dtype=torch.int64, device=dev)
        g.search_dev(nq, k, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=d_cid, d_coarse_dists=d_cd,
                     do_pruning=grouping, d_out_keys=kk, heap_order=True)
        ln = torch.empty((nq,), dtype=torch.int32, device=dev)
        cap = g.last_stream_dev(nq, d_len=ln)
        g.sync()
        assert int(ln.max().item()) <= cap
        shards.append((g, dd, ll))
        keys.append(kk)
        lens.append(ln)
    L = max(1, max(int(ln.max().item()) for ln in lens))
    for (g, _, _), ln in zip(shards, lens):
        st = torch.empty((nq, L), dtype=torch.int64, device=dev)
        g.last_stream_dev(nq, L, d_keys=st)
        g.sync()
        streams.append(st)

    def resolve(merged):
        label = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
        for g, dd, ll in shards:
            g.resolve_keys_dev(nq, k, merged, dd, ll)
            g.sync()
            label = torch.maximum(label, ll)
        return dd.cpu().numpy(), label.cpu().numpy()

    # ascending: the reference's result set
    asc_d, asc_l = resolve(D.merge_topk_keys(torch.stack(keys), k))
    assert (np.diff(asc_d, axis=1) >= 0).all()
    assert np.array_equal(np.sort(asc_l, axis=1), np.sort(ref_l, axis=1))
    # heap-array order: element for element what faiss's heap leaves
    merged, total = D.merge_streams(torch.stack(streams), torch.stack(lens), cap)
    hk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    shards[0][0].replay_stream_dev(nq, k, merged, total, merged.shape[1], hk)
    shards[0][0].sync()
    heap_d, heap_l = resolve(hk)
    assert np.array_equal(heap_l, ref_l)
    assert np.array_equal(heap_d.view(np.uint32), ref_d.view(np.uint32))


def test_sharded_device_generated_corpus(gpu):
    """ivfhnsw_gpu_upload_ivf_synthetic with shard_world > 1: every shard generates exactly its slice of the
    unsharded stream, so the merged result equals the unsharded device result and the oracle."""
    import torch
    t = synth.make_throughput_tables(seed=5, nc=512, d=128, M=16, n_total=200000)
    ids, codes, norm_codes = synth.synthetic_codes(99, t["offsets"], 16)
    gr = synth.orc.Hnsw.build(t["centroids"], M=16, efConstruction=60)
    t.update(ids=ids, codes=codes, norm_codes=norm_codes, centroid_norms=gr.centroid_norms(), graph=gr)
    ox = synth.oracle_index(t)
    ox.set_params(16, 5000, 40)
    rng = np.random.default_rng(3)
    q = (t["centroids"][rng.choice(512, 64)] + rng.normal(0, 10, size=(64, 128))).astype(np.float32)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(q, k=1)
    dev = torch.device("cuda", 0)
    d_q, d_cid, d_cd = (torch.from_numpy(a).to(dev) for a in (q, cid.astype(np.int32), cd))
    world, nq = 3, 64
    merged, shards = None, []
    for r in range(world):
        g = gpu()
        g.upload_ivf_synthetic(128, 16, t["offsets"], t["centroid_norms"], t["pq_centroids"], t["norm_table"], seed=99,
                               shard_rank=r, shard_world=world)
        dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        g.search_dev(nq, 1, d_q, dd, ll, 16, 5000, d_coarse_ids=d_cid, d_coarse_dists=d_cd, d_out_keys=kk)
        g.sync()
        merged = kk if merged is None else torch.minimum(merged, kk)
        shards.append((g, dd, ll))
    label = torch.full((nq, 1), -1, dtype=torch.int64, device=dev)
    for g, dd, ll in shards:
        g.resolve_keys_dev(nq, 1, merged, dd, ll)
        g.sync()
        label = torch.maximum(label, ll)
    assert np.array_equal(label.cpu().numpy(), ref_l)
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.parametrize("nsubc", [0, 8])
def test_sharded_searcher_step_with_opq(gpu, pkg, nsubc):
    """ShardedSearcher.step on one rank (no collectives) with an OPQ index: the walk must run on the rotated slice
    (ivfhnsw_gpu_rotate_dev), the tables on the rotated batch -- labels and distances of the oracle."""
    import importlib
    import torch
    pkg_dist = importlib.import_module("ivfhnsw_amd.distributed")
    c = corpus(seed=43, nc=256, d=96, M=8, n_base=20000, nq=80, nsubc=nsubc, opq=True)
    nprobe, max_codes, ef = 16, 2000, 48
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=bool(nsubc))
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    if nsubc:
        g.upload_grouping(nsubc, c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    dev = torch.device("cuda", 0)
    nq = len(ref_l)
    d_q = torch.from_numpy(c["queries"]).to(dev)
    dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    s = pkg_dist.ShardedSearcher(g, 0, 1, nq, nprobe, dev)
    s.step(d_q, dd, ll, max_codes, ef, do_pruning=bool(nsubc))
    g.sync()
    assert np.array_equal(ll.cpu().numpy(), ref_l)
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.parametrize("world,d,M", [(8, 128, 16), (9, 96, 16), (8, 128, 8), (10, 96, 8)])
def test_sharded_pipelined_scan(gpu, pkg, world, d, M):
    """Batches of >= 1024 queries on one of >= 8 list shards take scan_pipe_kernel (kernels_scan3.hip: table + scan software-pipelined
    over queries, no table in HBM).  Plans longer than its first pass (2048 codes), queries with nothing on a shard,
    and every code-book shape it is built for; merged keys must give the oracle's labels and distance bits."""
    import torch
    c = corpus(seed=300 + world, nc=128, d=d, M=M, n_base=40000, nq=1300)
    nprobe, max_codes, ef = 64, 30000, 80  # ~20 k codes per query: some shards hold more than one pass of them
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, cid, cd, st = ox.search_batch(c["queries"], k=1)
    nq = len(ref_l)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(c["queries"]).to(dev)
    d_cid = torch.from_numpy(cid.astype(np.int32)).to(dev)
    d_cd = torch.from_numpy(cd).to(dev)
    # an owner table that leaves rank 0 few lists: many queries find nothing there (empty plans in the pipeline)
    owner = (np.arange(128) % world).astype(np.int32)
    owner[owner == 0] = 1
    owner[:3] = 0
    shards, merged, total_codes = [], None, 0
    for r in range(world):
        g = gpu()
        ids, codes, ncodes = _shard_arrays(c, r, world, owner)
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], ids, codes, ncodes, c["centroid_norms"], c["pq_centroids"],
                     c["norm_table"], shard_rank=r, shard_world=world, list_owner=owner)
        dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        g.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=d_cid, d_coarse_dists=d_cd, d_out_keys=kk)
        g.sync()
        assert g.last_scan_kernel() == "scan_pipe_kernel"
        total_codes += g.last_scan_counts()[0]
        merged = kk if merged is None else torch.minimum(merged, kk)
        shards.append((g, dd, ll))
    assert total_codes == st.ncode
    label = torch.full((nq, 1), -1, dtype=torch.int64, device=dev)
    for g, dd, ll in shards:
        g.resolve_keys_dev(nq, 1, merged, dd, ll)
        g.sync()
        label = torch.maximum(label, ll)
    assert np.array_equal(label.cpu().numpy(), ref_l)
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.parametrize("grouping", [False, True])
def test_search_sharded_entry_point(gpu, pkg, grouping):
    """ivfhnsw_gpu_search_sharded: the whole shard step for one process holding all shard handles (what the bundled classes
    call with IVFHNSW_SHARDS=N).  Three shards on the one GPU share a device, so the keys are merged on the host here; on a
    node with a device per shard the same call merges them with RCCL all-reduces (ncclMin / ncclMax)."""
    if grouping:
        c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    else:
        c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    nprobe, max_codes, ef, world = 16, 2500, 40, 3
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
    owner = _owner_table(pkg, c, world, "spatial")
    gr = c["graph"]
    shards = []
    for r in range(world):
        g = gpu()
        ids, codes, ncodes = _shard_arrays(c, r, world, owner)
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], ids, codes, ncodes, c["centroid_norms"], c["pq_centroids"],
                     c["norm_table"], shard_rank=r, shard_world=world, list_owner=owner)
        if grouping:
            g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
            g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
        shards.append(g)
    for k in (1, 7):
        ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=k)
        dist, lab = pkg.search_sharded(shards, c["queries"], k, nprobe, max_codes, cid, cd, do_pruning=grouping)
        if k == 1:
            assert np.array_equal(lab, ref_l) and np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
        else:   # the reference's result set, ascending
            assert np.array_equal(np.sort(lab, 1), np.sort(ref_l, 1)) and (np.diff(dist, axis=1) >= 0).all()


def test_search_sharded_makes_the_rccl_calls(tmp_path):
    """A single shard with IVFHNSW_SHARDS_RCCL=1: ncclCommInitAll over one device and the two all-reduces (int64 MIN of the
    keys, MAX of the labels) are EXECUTED -- dtype, reduction, in-place buffers, group calls -- with results checked; what one
    GPU cannot show is the exchange between devices."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import __graft_entry__ as ge
import synth
pkg = ge.load_pkg()
c = synth.make_corpus(seed=13, nc=128, d=96, M=8, n_base=9000, nq=70, efConstruction=60)
ox = synth.oracle_index(c)
ox.set_params(8, 1500, 32)
ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)
g = pkg.GpuIndex(0)
g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
             c["pq_centroids"], c["norm_table"])
dist, lab = pkg.search_sharded([g], c["queries"], 1, 8, 1500, cid, cd)
assert np.array_equal(lab, ref_l) and np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
print("RCCL_SHARDED_OK")
""" % (root, root)
    env = dict(os.environ, IVFHNSW_SHARDS_RCCL="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_SHARDED_OK" in r.stdout, r.stderr[-3000:] + r.stdout[-500:]
    # the communicator really came from RCCL (the library logs nothing by default: ask it)
    env["NCCL_DEBUG"] = "VERSION"
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL" in (r.stdout + r.stderr).upper()
