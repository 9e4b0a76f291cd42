#!/usr/bin/env python3
"""One batch as TWO uneven parts on two streams (the handle and a view of it): the walk of part 1 fills the chip for whole
rounds of its 4096 resident wavefronts, part 2's walk moves into the slots part 1's last round frees, and part 1's scan
(LDS-bound) runs beside part 2's walk (HBM-bound) instead of after a half-empty tail.  Prints ms per 10 k-query batch
for the whole batch in one call and for several splits, results compared.
usage: python tools/split_probe.py [workload]"""
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
    g, nprobe, ef, mc, grouping = C.g, C.nprobe, C.ef, C.max_codes, C.grouping
    nq = C.nq
    d_q = torch.from_numpy(C.queries(nq, 1235)).to(dev)
    s0 = torch.cuda.current_stream()
    g.set_stream(s0.cuda_stream)
    ctx = [(g, s0)]
    for _ in range(3):
        v = g.view()
        st = torch.cuda.Stream(device=dev)
        v.set_stream(st.cuda_stream)
        ctx.append((v, st))
    ref_d = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    ref_l = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    dd = torch.empty_like(ref_d)
    ll = torch.empty_like(ref_l)

    def whole():
        g.search_dev(nq, 1, d_q, ref_d, ref_l, nprobe, mc, efSearch=ef, do_pruning=grouping)

    def split(parts):
        bounds = np.r_[0, np.cumsum(parts)].tolist()
        assert bounds[-1] == nq

        def f():
            ev = torch.cuda.Event()
            ev.record(s0)
            for i in range(len(parts)):
                h, st = ctx[i]
                a, b = bounds[i], bounds[i + 1]
                if i:
                    st.wait_event(ev)
                h.search_dev(b - a, 1, d_q[a:b], dd[a:b], ll[a:b], nprobe, mc, efSearch=ef, do_pruning=grouping)
            for i in range(1, len(parts)):
                e2 = torch.cuda.Event()
                e2.record(ctx[i][1])
                s0.wait_event(e2)
        return f

    def timed(f, reps=40):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    print("whole batch, one call            %.4f ms" % timed(whole), flush=True)
    for parts in ((7800, 2200), (7000, 3000), (6000, 2500, 1500)):
        t = timed(split(parts))
        ok = bool(torch.equal(ll, ref_l)) and bool(torch.equal(dd.view(torch.int32), ref_d.view(torch.int32)))
        print("split %-28s  %.4f ms   results equal: %s" % (" | ".join(map(str, parts)), t, ok), flush=True)
    # what the overlap costs the kernels of part 1 (stage events of the main handle, HIP events on its stream)
    for label, f, n1 in (("whole", whole, nq), ("split 7800 | 2200", split((7800, 2200)), 7800)):
        g.set_profiling(True)
        g.reset_stage_ms()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        st = {a: round(b[0] / max(1, b[1]), 4) for a, b in g.stage_ms().items()}
        nc = g.last_scan_counts()[0]
        g.set_profiling(False)
        print("%-18s main handle's stages (ms per launch) %s; codes %d -> scan %.1f GB/s" %
              (label, st, nc, 17 * nc / (st["scan"] * 1e-3) / 1e9), flush=True)
    print("whole batch, one call (again)    %.4f ms" % timed(whole), flush=True)


if __name__ == "__main__":
    main()
