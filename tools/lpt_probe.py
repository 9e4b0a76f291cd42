#!/usr/bin/env python3
"""Does the ORDER of a batch matter to the walk?  The walk's wavefronts pull queries from a counter, so the launch ends
with the tail of the last ones started; longest-first would shorten it.  Times ivfhnsw_gpu_coarse_dev on the bench corpus
for the batch as given, sorted by distance to the enter point (a proxy of the walk's length) both ways, and sorted by the
TRUE per-query length (the walk's distance to the nearest centroid as a stand-in is not available: an oracle-free upper
bound comes from timing sub-batches).  usage: python tools/lpt_probe.py [workload]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import __graft_entry__ as ge
    import bench
    import synth
    pkg = ge.load_pkg()
    dev = torch.device("cuda", 0)
    name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
    C = bench.Corpus(pkg, synth, name, 1234, dev, 0)
    g, nprobe, ef = C.g, C.nprobe, C.ef
    nq = C.nq
    q = C.queries(nq, 1235)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    cid = torch.empty((nq, nprobe), dtype=torch.int32, device=dev)
    cd = torch.empty((nq, nprobe), dtype=torch.float32, device=dev)

    def timed(qq, reps=30):
        d_q = torch.from_numpy(np.ascontiguousarray(qq)).to(dev)
        for _ in range(3):
            g.coarse_dev(nq, d_q, nprobe, ef, cid, cd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.coarse_dev(nq, d_q, nprobe, ef, cid, cd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    d_enter = ((q - C.vectors[0][None, :]) ** 2).sum(1)
    g.coarse_dev(nq, torch.from_numpy(q).to(dev), nprobe, ef, cid, cd)
    torch.cuda.synchronize()
    d_first = cd.cpu().numpy()[:, 0]
    d_last = cd.cpu().numpy()[:, nprobe - 1]
    rng = np.random.default_rng(0)
    rows = [("as given", np.arange(nq)), ("random permutation", rng.permutation(nq)),
            ("far from the enter point first", np.argsort(-d_enter)), ("near the enter point first", np.argsort(d_enter)),
            ("large nearest-centroid distance first", np.argsort(-d_first)), ("small ... first", np.argsort(d_first)),
            ("large nprobe-th distance first", np.argsort(-d_last)), ("as given (again)", np.arange(nq))]
    for label, order in rows:
        print("%-42s %.4f ms" % (label, timed(q[order])), flush=True)


if __name__ == "__main__":
    main()
