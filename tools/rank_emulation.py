#!/usr/bin/env python3
"""What ONE rank of an N-GPU step computes, run alone on one GPU (no collectives): rank r's shard of the workload's
corpus under the owner table bench.py --gpus N would use; the walk for its slice of the batch, then plan + tables +
scan of ITS lists for all queries, then the label resolution.  Prints per-stage times -- the compute part of
`bench.py --gpus N` per rank -- and, from the plan recomputed on the host, the codes every rank would score per step
(max / mean: the load balance of the owner table).
--world is the number of list shards of ONE replica group (bench.py --list-shards; 8 GPUs default to 2 groups x 4).
usage: python tools/rank_emulation.py [--world 8] [--rank 0] [--scaling weak|strong] [--partition spatial|mod] [--workload W]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--partition", choices=("spatial", "mod"), default="mod")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--split", type=int, default=0, help="permille of the rank's slice in the first of two parts on two "
                    "streams (0 = one part): what a two-part sharded step would overlap, collectives aside")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import bench
    import synth
    pkg = ge.load_pkg()
    D = importlib.import_module("ivfhnsw_amd.distributed")
    dev = torch.device("cuda", 0)
    W, r = args.world, args.rank
    name = args.workload or bench.DEFAULT_WORKLOAD
    C = bench.Corpus(pkg, synth, name, 1234, dev, 0, rank=r, world=W, partition=args.partition, pkg_dist=D)
    g, nprobe, max_codes, ef, grouping = C.g, C.nprobe, C.max_codes, C.ef, C.grouping
    nq = bench.STRONG_BATCH if args.scaling == "strong" else C.nq * W
    lo, hi, per = D.query_slice(nq, r, W)
    queries = C.queries(nq, 1235)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_q = torch.from_numpy(queries).to(dev)
    xr = torch.empty_like(d_q)
    g.rotate_dev(nq, d_q, xr)
    cid = torch.empty((nq, nprobe), dtype=torch.int32, device=dev)
    cd = torch.empty((nq, nprobe), dtype=torch.float32, device=dev)
    g.coarse_dev(nq, xr, nprobe, ef, cid, cd)          # what the all-gather would deliver
    dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    own_c = torch.empty((per, nprobe), dtype=torch.int32, device=dev)
    own_d = torch.empty((per, nprobe), dtype=torch.float32, device=dev)

    def step():
        g.rotate_dev(hi - lo, d_q[lo:hi], xr[lo:hi])
        g.coarse_dev(hi - lo, xr[lo:hi], nprobe, ef, own_c, own_d)
        g.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=cid, d_coarse_dists=cd, d_out_keys=kk,
                     do_pruning=grouping)
        g.resolve_keys_dev(nq, 1, kk, dd, ll)

    if args.split:
        # two-part form: the own slice as n1 | n2 queries; part p's scan covers the W * n_p queries every rank walked in its
        # part p (here: the first W * n1 and the last W * n2 of the batch), part 2 on a view and a second stream
        n2 = max(2048 // 1, ((hi - lo) * (1000 - args.split) // 1000 + 1024) // 2048 * 2048)
        n1 = (hi - lo) - n2
        v = g.view()
        st2 = torch.cuda.Stream(device=dev)
        v.set_stream(st2.cuda_stream)
        s0 = torch.cuda.current_stream()
        dd2, ll2, kk2 = torch.empty_like(dd), torch.empty_like(ll), torch.empty_like(kk)
        own_c2, own_d2 = torch.empty_like(own_c), torch.empty_like(own_d)
        a1, a2 = W * n1, W * n2

        def step():  # noqa: F811
            ev = torch.cuda.Event()
            ev.record(s0)
            st2.wait_event(ev)
            g.rotate_dev(n1, d_q[lo:lo + n1], xr[lo:lo + n1])
            g.coarse_dev(n1, xr[lo:lo + n1], nprobe, ef, own_c, own_d)
            g.search_dev(a1, 1, d_q[:a1], dd[:a1], ll[:a1], nprobe, max_codes, d_coarse_ids=cid[:a1], d_coarse_dists=cd[:a1],
                         d_out_keys=kk[:a1], do_pruning=grouping)
            with torch.cuda.stream(st2):
                v.rotate_dev(n2, d_q[lo + n1:hi], xr[lo + n1:hi])
                v.coarse_dev(n2, xr[lo + n1:hi], nprobe, ef, own_c2, own_d2)
                v.search_dev(a2, 1, d_q[a1:], dd2[:a2], ll2[:a2], nprobe, max_codes, d_coarse_ids=cid[a1:],
                             d_coarse_dists=cd[a1:], d_out_keys=kk2[:a2], do_pruning=grouping)
                v.resolve_keys_dev(a2, 1, kk2[:a2], dd2[:a2], ll2[:a2])
                e2 = torch.cuda.Event()
                e2.record(st2)
            g.resolve_keys_dev(a1, 1, kk[:a1], dd[:a1], ll[:a1])
            s0.wait_event(e2)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g.set_profiling(True)
    g.reset_stage_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / args.steps
    st = {a: round(b[0] / args.steps, 3) for a, b in g.stage_ms().items()}
    here = g.last_scan_counts()[0]
    if args.split:
        print("world %d rank %d two-part step (%d | %d of the own slice): %.3f ms per step of %d queries" % (W, r, n1, n2, t * 1e3, nq))
        return
    # every rank's scored codes from the plan rule on the host (IVFADC rule; Grouping prunes inside lists, same shares)
    bal = ""
    if not grouping:
        sizes = np.diff(C.tb["offsets"].astype(np.int64))
        ids = cid.cpu().numpy().astype(np.int64)
        sz = sizes[ids]
        before = np.cumsum(sz, axis=1) - sz
        take = (sz > 0) & ((before < max_codes) | (before == 0))
        owner = C.owner if C.owner is not None else np.zeros(C.nc, np.uint32)
        per_rank = np.bincount(owner[ids[take]], weights=sz[take], minlength=W)
        ranks_per_q = np.array([len(set(owner[ids[i][take[i]]].tolist())) for i in range(0, nq, max(1, nq // 2000))])
        bal = ("; codes per rank and step: max %.3g / mean %.3g = %.3f; ranks a query's scored lists touch: %.2f of %d"
               % (per_rank.max(), per_rank.mean(), per_rank.max() / per_rank.mean(), ranks_per_q.mean(), W))
        assert int(per_rank[r]) == here, (per_rank[r], here)
    print("world %d rank %d (%s, %s partition, %s): this rank's compute per step %.3f ms for %d queries -> %.2f M queries/s "
          "aggregate if collectives were free; stages %s; scan kernel %s; codes scored here per step %d%s"
          % (W, r, args.scaling, args.partition, name, t * 1e3, nq, nq / t / 1e6, st, g.last_scan_kernel(), here, bal),
          flush=True)


if __name__ == "__main__":
    main()
