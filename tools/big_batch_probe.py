#!/usr/bin/env python3
"""A LARGE batch in one ivfhnsw_gpu_search_dev call against the same batch cut into 10 k-query chunks that alternate
between the handle and a view (two chunks in flight): how much of the two-batches-in-flight rate a single call of many
queries leaves on the table.  usage: python tools/big_batch_probe.py [nq=80000] [chunk=10000]"""
import os
import sys
import time

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
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    C = bench.Corpus(pkg, synth, bench.DEFAULT_WORKLOAD, 1234, dev, 0)
    C.upload(0, 1)
    g, nprobe, ef, mc = C.g, C.nprobe, C.ef, C.max_codes
    d_q = torch.from_numpy(C.queries(nq, 1235)).to(dev)
    s0 = torch.cuda.current_stream()
    g.set_stream(s0.cuda_stream)
    v = g.view()
    s1 = torch.cuda.Stream(device=dev)
    v.set_stream(s1.cuda_stream)
    ref_d = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    ref_l = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    dd, ll = torch.empty_like(ref_d), torch.empty_like(ref_l)

    def whole():
        g.search_dev(nq, 1, d_q, ref_d, ref_l, nprobe, mc, efSearch=ef)

    def chunks(split):
        def f():
            g.set_batch_split(split)
            ev = torch.cuda.Event()
            ev.record(s0)
            s1.wait_event(ev)
            for i, a in enumerate(range(0, nq, chunk)):
                b = min(nq, a + chunk)
                h = g if i % 2 == 0 else v
                h.search_dev(b - a, 1, d_q[a:b], dd[a:b], ll[a:b], nprobe, mc, efSearch=ef)
            e2 = torch.cuda.Event()
            e2.record(s1)
            s0.wait_event(e2)
            g.set_batch_split(1000)
        return f

    def timed(f, reps=10):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t = timed(whole)
    print("one call of %d queries (default split): %.3f ms = %.2f M queries/s, parts %s" % (nq, t * 1e3, nq / t / 1e6, g.last_batch_parts()), flush=True)
    for split in (0, 1000):
        t = timed(chunks(split))
        ok = bool(torch.equal(ll, ref_l)) and bool(torch.equal(dd.view(torch.int32), ref_d.view(torch.int32)))
        print("chunks of %d alternating handle / view, each chunk %s: %.3f ms = %.2f M queries/s, equal %s"
              % (chunk, "in one part" if split == 0 else "split by estimate", t * 1e3, nq / t / 1e6, ok), flush=True)


if __name__ == "__main__":
    main()
