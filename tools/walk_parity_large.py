#!/usr/bin/env python3
"""Parity of the device walk on a graph of the reference's size (993 127 nodes: 10-bit visited tags, late entry into
the visited set, neighbour rows beyond the Infinity Cache) against the CPU oracle, on the bench's synthetic centroids.
usage: python tools/walk_parity_large.py [--nc 993127] [--nq 4000]"""
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
    ap.add_argument("--nc", type=int, default=993127)
    ap.add_argument("--nq", type=int, default=4000)
    ap.add_argument("--ef", type=int, default=80)
    ap.add_argument("--k", type=int, default=32)
    a = ap.parse_args()
    import __graft_entry__ as ge
    import synth
    from oracle import orc
    pkg = ge.load_pkg()
    rng = np.random.default_rng(77)
    cents = synth.sift_like(rng, a.nc, 128)
    counts, links = synth.knn_graph_torch(cents, 16, 32)
    q = (cents[rng.choice(a.nc, a.nq)] + rng.normal(0, 12.0, size=(a.nq, 128))).astype(np.float32)
    g = pkg.GpuIndex(0)
    g.upload_quantizer(counts, links, cents, 0)
    t0 = time.time()
    ids, dist = g.coarse(q, a.k, a.ef)
    t_gpu = time.time() - t0
    graph = orc.Hnsw.from_arrays(counts, links, cents, 16, 0)
    t0 = time.time()
    bad = 0
    for i in range(a.nq):
        rid, rd = graph.search_knn(q[i], a.ef, a.k)
        n = len(rid)
        if not (np.array_equal(ids[i, :n], rid) and np.array_equal(dist[i, :n].view(np.uint32), rd.view(np.uint32))):
            bad += 1
    print("walk parity at %d nodes: %d / %d queries identical (ids and distance bits, k %d, ef %d); device %.2fs, "
          "oracle %.2fs" % (a.nc, a.nq - bad, a.nq, a.k, a.ef, t_gpu, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
