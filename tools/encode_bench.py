#!/usr/bin/env python3
"""Throughput of the construction side (ivfhnsw_gpu_encode = IndexIVF_HNSW::add_batch up to the append loop) on one
GPU: one reference-sized batch of base vectors (tests/test_ivfhnsw_sift1b.cpp:100-160 adds 10^6 per batch and
assigns with efSearch = 220, :108) against the 2^17-centroid graph of the bench corpus.  Not part of bench.py's
contract; prints one line per configuration.  Host pointers in and out, so PCIe is included.

usage: python tools/encode_bench.py [--n 1000000] [--nc 131072] [--ef 220]
"""
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
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nc", type=int, default=1 << 17)
    ap.add_argument("--ef", type=int, default=220)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--M", type=int, default=16)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_pkg()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    tb = synth.make_throughput_tables(1234, args.nc, args.d, args.M, 1000 * args.nc)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32, device=dev)
    x = (tb["centroids"][rng.choice(args.nc, args.n)] + rng.normal(0, 15.0, size=(args.n, args.d))).astype(np.float32)
    g = pkg.GpuIndex(0)
    g.upload_quantizer(counts, links, tb["centroids"], 0)
    for opq in (False, True):
        A = synth.random_rotation(rng, args.d) if opq else None
        g.upload_codebooks(args.d, args.M, tb["pq_centroids"], tb["norm_table"], A)
        g.encode(x[:10000], efSearch=args.ef)  # warm-up: workspace, code caches
        g.set_profiling(True)
        g.reset_stage_ms()
        t0 = time.perf_counter()
        idx, codes, ncodes = g.encode(x, efSearch=args.ef)
        t_assign = time.perf_counter() - t0
        walk_ms = g.stage_ms()["coarse"][0]
        g.set_profiling(False)
        t0 = time.perf_counter()
        idx2, codes2, ncodes2 = g.encode(x, precomputed_idx=idx)
        t_pre = time.perf_counter() - t0
        assert np.array_equal(codes, codes2) and np.array_equal(ncodes, ncodes2)
        print("opq=%d  n=%d  nc=%d: assign(ef %d)+encode %.1f ms (walk kernels %.1f ms) = %.2f M vectors/s;  "
              "encode with given assignments %.1f ms = %.2f M vectors/s  [host pointers: %.0f MB in, %.0f MB out]"
              % (opq, args.n, args.nc, args.ef, t_assign * 1e3, walk_ms, args.n / t_assign / 1e6, t_pre * 1e3,
                 args.n / t_pre / 1e6, x.nbytes / 1e6, (codes.nbytes + ncodes.nbytes + idx.nbytes) / 1e6), flush=True)


def groups():
    """IndexIVF_HNSW_Grouping::add_group for every centroid in one call: 2^20 points in 4096 groups, nsubc 64."""
    import torch
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_pkg()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    nc, d, M, n, nsubc = 4096, 128, 16, 1 << 20, 64
    tb = synth.make_throughput_tables(1234, nc, d, M, 1000 * nc)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32, device=dev)
    key = np.sort(rng.integers(0, nc, size=n)).astype(np.uint32)
    x = (tb["centroids"][key] + rng.normal(0, 15.0, size=(n, d))).astype(np.float32)
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(key, minlength=nc))
    g = pkg.GpuIndex(0)
    g.upload_quantizer(counts, links, tb["centroids"], 0)
    g.upload_codebooks(d, M, tb["pq_centroids"], tb["norm_table"])
    g.encode_groups(nsubc, np.arange(64, dtype=np.uint32), offsets[:65], x[:int(offsets[64])], 220)  # warm-up
    t0 = time.perf_counter()
    g.encode_groups(nsubc, np.arange(nc, dtype=np.uint32), offsets, x, 220)
    t = time.perf_counter() - t0
    print("add_group: %d points in %d groups, nsubc %d: %.1f ms = %.2f M vectors/s" % (n, nc, nsubc, t * 1e3, n / t / 1e6),
          flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "groups":
        groups()
        sys.exit(0)
    main()
