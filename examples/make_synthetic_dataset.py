#!/usr/bin/env python3
"""Writes a small synthetic data set in exactly the files the reference's drivers read (examples/run_*.sh of the
reference): centroids.fvecs, HNSW info/edges, pq / norm_pq / opq, the .index, queries.fvecs, groundtruth.ivecs.
No real SIFT/DEEP data is needed (none is available offline).  Index construction is done here in numpy + the CPU
checker's reference-identical HNSW construction, because construction is outside the MI355X search path.

usage: make_synthetic_dataset.py OUT_DIR [--grouping] [--opq] [--nc 1024] [--nb 100000] [--nq 1000] [--d 96]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hostio  # noqa: E402
import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--grouping", action="store_true")
ap.add_argument("--opq", action="store_true")
ap.add_argument("--nc", type=int, default=1024)
ap.add_argument("--nb", type=int, default=100000)
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--d", type=int, default=96)
ap.add_argument("--code_size", type=int, default=16)
ap.add_argument("--nsubc", type=int, default=16)
a = ap.parse_args()

os.makedirs(a.out, exist_ok=True)
c = synth.make_corpus(seed=2024, nc=a.nc, d=a.d, M=a.code_size, n_base=a.nb, nq=a.nq, efConstruction=200, opq=a.opq,
                      nsubc=a.nsubc if a.grouping else 0)
paths = hostio.dump_corpus(c, a.out)
gt = np.empty((a.nq, 1), np.int32)
for i in range(0, a.nq, 256):  # exact nearest neighbour of every query in the base set
    q = c["queries"][i:i + 256]
    gt[i:i + 256, 0] = ((q ** 2).sum(1)[:, None] - 2 * q @ c["base"].T + (c["base"] ** 2).sum(1)[None, :]).argmin(1)
hostio.write_xvecs(os.path.join(a.out, "groundtruth.ivecs"), gt)
open(os.path.join(a.out, "precomputed_idxs.ivecs"), "wb").close()
print("wrote", a.out, {k: os.path.basename(v) for k, v in paths.items()})
