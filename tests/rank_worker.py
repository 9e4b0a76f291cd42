"""One rank of tests/test_gpu_ranks.py: a child process that joins a gloo group of `world` ranks sharing GPU 0 (RCCL refuses
two ranks on one device; on the 8-GPU node the backend is nccl and every rank has its own GPU) and runs the REAL
ivf-hnsw_amd/distributed.py::ShardedSearcher.step on its shard of the index, checked against the unsharded oracle:

  IVFADC              k = 1, k = 10 ascending, k = 10 in faiss heap-array order, spatial owner table (partition_lists)
  Grouping + OPQ      the same three, pruning on, owner table c % world
usage: python rank_worker.py <rank> <world> <port> <out_dir>
"""
import datetime
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shard_arrays(c, rank, owner):
    import numpy as np
    off = c["offsets"].astype(np.int64)
    owned = [cc for cc in range(len(off) - 1) if owner[cc] == rank]
    sel = np.concatenate([np.arange(off[cc], off[cc + 1]) for cc in owned]) if owned else np.zeros(0, np.int64)
    return c["ids"][sel], c["codes"][sel], c["norm_codes"][sel]


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import synth

    pkg = ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    notes, ok = [], True
    for nsubc in (0, 8):
        grouping = nsubc > 0
        c = synth.make_corpus(seed=61 + nsubc, nc=256, d=128, M=16, n_base=24000, nq=1100, nsubc=nsubc, opq=grouping)
        nprobe, max_codes, ef = 16, 2000, 48
        ox = synth.oracle_index(c)
        sizes = np.diff(c["offsets"].astype(np.int64))
        if grouping:
            owner = D.partition_lists(c["centroids"], sizes, world, "mod")
        else:
            owner = D.partition_lists(c["centroids"], sizes, world, "spatial")
        assert set(owner.tolist()) == set(range(world))
        ids, codes, ncodes = shard_arrays(c, rank, owner)
        g = pkg.GpuIndex(0)
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], ids, codes, ncodes, c["centroid_norms"], c["pq_centroids"],
                     c["norm_table"], opq_A=c["opq_A"], shard_rank=rank, shard_world=world, list_owner=owner)
        gr = c["graph"]
        g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
        if grouping:
            g.upload_grouping(nsubc, c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
        nq = len(c["queries"])
        d_q = torch.from_numpy(c["queries"]).to(dev)
        for k, heap in ((1, False), (10, False), (10, True)):
            ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
            ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=k, nthreads=4)
            dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
            ll = torch.empty((nq, k), dtype=torch.int64, device=dev)
            s = D.ShardedSearcher(g, rank, world, nq, nprobe, dev, k=k)   # binds the handle to torch's stream
            for _ in range(2):   # twice: the second step reuses every buffer
                s.step(d_q, dd, ll, max_codes, ef, do_pruning=grouping, heap_order=heap)
            torch.cuda.synchronize()
            lab, dis = ll.cpu().numpy(), dd.cpu().numpy()
            if k == 1 or heap:
                good = np.array_equal(lab, ref_l) and np.array_equal(dis.view(np.uint32), ref_d.view(np.uint32))
            else:
                good = np.array_equal(np.sort(lab, 1), np.sort(ref_l, 1)) and bool((np.diff(dis, axis=1) >= 0).all())
            ok &= bool(good)
            notes.append("nsubc=%d k=%d heap=%s: %s" % (nsubc, k, heap, good))
        # the k = 1 step as two overlapping parts (the handle + a view on a side stream): a batch whose slices reach split_min
        nqb = 8192
        qb = np.ascontiguousarray(np.tile(c["queries"], (nqb // nq + 1, 1))[:nqb] + np.float32(0.25) * (np.arange(nqb, dtype=np.float32)[:, None] % 7))
        ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
        rdb, rlb, _, _, stb = ox.search_batch(qb, k=1, nthreads=4)
        d_qb = torch.from_numpy(qb).to(dev)
        ddb = torch.empty((nqb, 1), dtype=torch.float32, device=dev)
        llb = torch.empty((nqb, 1), dtype=torch.int64, device=dev)
        s2 = D.ShardedSearcher(g, rank, world, nqb, nprobe, dev, k=1, split_min=4096)
        assert s2.parts is not None and [p["n"] for p in s2.parts] == [2048, 2048]
        for _ in range(2):
            s2.step(d_qb, ddb, llb, max_codes, ef, do_pruning=grouping)
        torch.cuda.synchronize()
        good = np.array_equal(llb.cpu().numpy(), rlb) and np.array_equal(ddb.cpu().numpy().view(np.uint32), rdb.view(np.uint32))
        tot = torch.tensor([float(s2.last_scan_counts()[0])], dtype=torch.float64)
        dist.all_reduce(tot)
        good = good and int(tot.item()) == stb.ncode    # the shards' two parts partition the scanned codes exactly
        ok &= bool(good)
        notes.append("nsubc=%d two-part step: %s" % (nsubc, good))
        s2.close()
        # a step on another torch stream than the bound one must be refused, not silently mis-ordered
        other = torch.cuda.Stream(device=dev)
        try:
            with torch.cuda.stream(other):
                s.step(d_q, dd, ll, max_codes, ef, do_pruning=grouping, heap_order=True)
            ok = False
            notes.append("step on a foreign stream was accepted")
        except RuntimeError:
            pass
        g.close()
    dist.barrier()
    open(os.path.join(out_dir, "rank%d.%s" % (rank, "ok" if ok else "fail")), "w").write("\n".join(notes) + "\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
