#!/usr/bin/env python3
"""k > 1 throughput of the search path on the bench corpus (configs[1] shape): both result orders
(heap_order 0 = ascending, 1 = the array faiss's max-heap leaves).  Host-pointer API, so PCIe is included.
usage: python tools/topk_bench.py [--ks 1,10,100]"""
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
    ap.add_argument("--ks", default="1,10,100")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_pkg()
    dev = torch.device("cuda", 0)
    n_total, nc, d, M, nprobe, max_codes, ef, nq = 100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000
    tb = synth.make_throughput_tables(1234, nc, d, M, n_total)
    rng = np.random.default_rng(1235)
    queries = (tb["centroids"][rng.choice(nc, nq)] + rng.normal(0, 12.0, size=(nq, d))).astype(np.float32)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32, device=dev)
    cn = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
    g = pkg.GpuIndex(0)
    g.upload_ivf_synthetic(d, M, tb["offsets"], cn, tb["pq_centroids"], tb["norm_table"], 1241)
    g.upload_quantizer(counts, links, tb["centroids"], 0)
    for k in [int(x) for x in args.ks.split(",")]:
        for heap in (False, True):
            if k == 1 and heap:
                continue
            g.search(queries, k, nprobe, max_codes, efSearch=ef, heap_order=heap)
            g.set_profiling(True)
            g.reset_stage_ms()
            t0 = time.perf_counter()
            for _ in range(3):
                g.search(queries, k, nprobe, max_codes, efSearch=ef, heap_order=heap)
            t = (time.perf_counter() - t0) / 3
            st = {a: round(b[0] / 3, 3) for a, b in g.stage_ms().items()}
            g.set_profiling(False)
            print("k=%d heap_order=%d: %.2f ms per 10k queries = %.2f M q/s  stages %s" % (k, heap, t * 1e3, nq / t / 1e6, st),
                  flush=True)


if __name__ == "__main__":
    main()
