"""The RCCL calls of the multi-GPU step, executed: a single-rank `nccl` (= RCCL) process group accepts every collective
ShardedSearcher.step issues -- all-gather of the coarse stage, int64 MIN all-reduce of the packed keys, int64 MAX of the
labels, all-gather of the k > 1 keys and of the candidate streams, and the two-part step on a view and a side stream -- so a one-GPU box can run the very code path the
8-GPU node takes (dtypes, shapes, in-place semantics), with results checked against the oracle.  What it cannot show is
the exchange between ranks: that is tests/test_distributed_cpu.py (gloo, world 2 / 3) and tools/rehearse_ranks.sh."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
import synth
pkg = ge.load_pkg()
D = importlib.import_module("ivfhnsw_amd.distributed")
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29631"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for nsubc in (0, 8):
    c = synth.make_corpus(seed=44, nc=256, d=128, M=16, n_base=20000, nq=96, nsubc=nsubc, opq=bool(nsubc))
    nprobe, max_codes, ef = 16, 2000, 48
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=bool(nsubc))
    g = pkg.GpuIndex(0)
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"], shard_rank=0, shard_world=1,
                 list_owner=np.zeros(c["nc"], np.uint32))
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    if nsubc:
        g.upload_grouping(nsubc, c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    nq = len(c["queries"])
    d_q = torch.from_numpy(c["queries"]).to(dev)
    for k, heap in ((1, False), (10, False), (10, True)):
        ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=k)
        dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, k), dtype=torch.int64, device=dev)
        s = D.ShardedSearcher(g, 0, 1, nq, nprobe, dev, k=k, force_collectives=True)
        s.step(d_q, dd, ll, max_codes, ef, do_pruning=bool(nsubc), heap_order=heap)
        torch.cuda.synchronize()
        lab, dis = ll.cpu().numpy(), dd.cpu().numpy()
        if k == 1 or heap:
            assert np.array_equal(lab, ref_l), (nsubc, k, heap)
            assert np.array_equal(dis.view(np.uint32), ref_d.view(np.uint32)), (nsubc, k, heap)
        else:
            assert np.array_equal(np.sort(lab, 1), np.sort(ref_l, 1)) and (np.diff(dis, axis=1) >= 0).all()
    # the two-part step (the default of a rank whose slice is >= 8192 queries): part 2 on a view and a side stream, the
    # all-gathers of both parts, ONE MIN and ONE MAX all-reduce -- the same calls on RCCL
    nqb = 4096
    qb = np.ascontiguousarray(np.tile(c["queries"], (nqb // nq + 1, 1))[:nqb] + np.float32(0.5) * (np.arange(nqb, dtype=np.float32)[:, None] %% 5))
    rdb, rlb, _, _, stb = ox.search_batch(qb, k=1)
    d_qb = torch.from_numpy(qb).to(dev)
    ddb = torch.empty((nqb, 1), dtype=torch.float32, device=dev)
    llb = torch.empty((nqb, 1), dtype=torch.int64, device=dev)
    s2 = D.ShardedSearcher(g, 0, 1, nqb, nprobe, dev, k=1, force_collectives=True, split_min=4096)
    assert s2.parts is not None and [p["n"] for p in s2.parts] == [2048, 2048]
    for _ in range(2):
        s2.step(d_qb, ddb, llb, max_codes, ef, do_pruning=bool(nsubc))
    torch.cuda.synchronize()
    assert np.array_equal(llb.cpu().numpy(), rlb) and np.array_equal(ddb.cpu().numpy().view(np.uint32), rdb.view(np.uint32)), nsubc
    assert s2.last_scan_counts()[0] == stb.ncode
    s2.close()
    g.close()
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


def test_every_collective_of_the_sharded_step_runs_on_rccl():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stderr[-3000:] + r.stdout[-500:]
