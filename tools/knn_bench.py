"""Times ivfhnsw_gpu_knn_dev (kernels_knn.hip) at the shapes the bench uses it for: the 993 127-node neighbour table
behind the coarse graph (k = 16), Grouping's table (k = 64), and ground truth for 10 k queries against 10 M rows.
usage: python tools/knn_bench.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import __graft_entry__ as ge

pkg = ge.load_pkg()
dev = torch.device("cuda", 0)
g = pkg.GpuIndex(0)
g.set_stream(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 993127


def run(nq, nx, d, k, same):
    x = torch.randn((nx, d), device=dev) * 35 + 30
    q = x if same else torch.randn((nq, d), device=dev) * 35 + 30
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    g.knn_dev(nq, nx, d, q, x, k, ids, dist, exclude_self=same)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.knn_dev(nq, nx, d, q, x, k, ids, dist, exclude_self=same)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 2.0 * nq * nx * d
    print("knn nq=%d nx=%d d=%d k=%d: %.3f s, %.1f TFLOP/s (f32 MFMA)" % (nq, nx, d, k, dt, flop / dt / 1e12), flush=True)


run(n, n, 128, 16, True)
run(n, n, 96, 16, True)
run(n, n, 128, 64, True)
run(10000, 10_000_000, 128, 10, False)
