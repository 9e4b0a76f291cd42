#!/usr/bin/env python3
"""What ONE rank of an N-GPU weak-scaling step computes, run alone on one GPU (no collectives): shard 0 of N of an
N x 100M corpus under N x 2^17 centroids; the walk for its 10 k queries, then tables + plan + scan of ITS lists for
all N x 10 k queries.  Prints per-stage times, i.e. the compute part of bench.py --gpus N per rank.
usage: python tools/rank_emulation.py [--world 8]"""
import argparse
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
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_pkg()
    dev = torch.device("cuda", 0)
    W = args.world
    n_total, nc, d, M, nprobe, max_codes, ef, nq1 = 100_000_000 * W, (1 << 17) * W, 128, 16, 32, 10000, 80, 10000
    nq = nq1 * W
    tb = synth.make_throughput_tables(1234, nc, d, M, n_total)
    rng = np.random.default_rng(1235)
    queries = (tb["centroids"][rng.choice(nc, nq)] + rng.normal(0, 12.0, size=(nq, d))).astype(np.float32)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32, device=dev)
    cn = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
    g = pkg.GpuIndex(0)
    g.upload_ivf_synthetic(d, M, tb["offsets"], cn, tb["pq_centroids"], tb["norm_table"], 1241, shard_rank=0, shard_world=W)
    g.upload_quantizer(counts, links, tb["centroids"], 0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_q = torch.from_numpy(queries).to(dev)
    cid = torch.empty((nq, nprobe), dtype=torch.int32, device=dev)
    cd = torch.empty((nq, nprobe), dtype=torch.float32, device=dev)
    g.coarse_dev(nq, d_q, nprobe, ef, cid, cd)          # what the all-gather would deliver
    dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    own_c = torch.empty((nq1, nprobe), dtype=torch.int32, device=dev)
    own_d = torch.empty((nq1, nprobe), dtype=torch.float32, device=dev)

    def step():
        g.coarse_dev(nq1, d_q[:nq1], nprobe, ef, own_c, own_d)
        g.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=cid, d_coarse_dists=cd, d_out_keys=kk)
        g.resolve_keys_dev(nq, 1, kk, dd, ll)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g.set_profiling(True)
    g.reset_stage_ms()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    st = {a: round(b[0] / 5, 3) for a, b in g.stage_ms().items()}
    print("world %d: one rank's compute per step %.3f ms -> %.2f M queries/s aggregate if collectives were free "
          "(N=1 measures 1.53 ms for 10 k); stages %s; codes scored here per step %d"
          % (W, t * 1e3, nq / t / 1e6, st, g.last_scan_counts()[0]), flush=True)


if __name__ == "__main__":
    main()
