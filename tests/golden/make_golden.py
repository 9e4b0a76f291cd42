#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: outputs of the CPU oracle on seeded synthetic corpora.

These are NOT reference outputs (the reference cannot be built or run here and ships no fixtures: parity
unpinned).  They pin the oracle itself across rounds -- a change in the oracle's arithmetic or
in the corpus generator shows up as a diff -- and give the GPU tests a second, stored expectation.

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import synth  # noqa: E402

CASES = {
    "ivf_pq16": (dict(seed=101, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80), 8, 1500, 32, False),
    "ivf_pq8_opq": (dict(seed=102, nc=64, d=64, M=8, n_base=4000, nq=32, efConstruction=60, opq=True), 8, 10 ** 9, 32, False),
    "grouping_pruned": (dict(seed=103, nc=128, d=128, M=16, n_base=8000, nq=32, nsubc=8, efConstruction=80), 8, 600, 40, True),
    "grouping_full": (dict(seed=103, nc=128, d=128, M=16, n_base=8000, nq=32, nsubc=8, efConstruction=80), 8, 600, 40, False),
}


def run_case(kw, nprobe, max_codes, ef, pruning):
    c = synth.make_corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=pruning)
    d1, l1, cid, cd, st = ox.search_batch(c["queries"], k=1)
    d5, l5, _, _, _ = ox.search_batch(c["queries"], k=5)
    # a digest of the inputs, so that a generator change is told apart from an oracle change
    digest = np.array([int(c["codes"].astype(np.uint64).sum()), int(c["ids"].astype(np.uint64).sum()),
                       int(c["graph"].links.astype(np.uint64).sum()), int(c["graph"].counts.astype(np.uint64).sum())],
                      np.uint64)
    return dict(dist1=d1, lab1=l1, coarse_ids=cid, coarse_dists=cd, dist5=d5, lab5=l5,
                counts=np.array([st.ncode, st.nseg, st.dist_evals], np.uint64), input_digest=digest), c


ENCODE_CASES = {
    # name: make_encode_case arguments (construction side, IndexIVF_HNSW.cpp:75-121)
    "encode_pq16_opq": dict(seed=111, nc=300, d=128, M=16, opq=True, n=700),
    "encode_pq8_d96": dict(seed=112, nc=300, d=96, M=8, opq=False, n=700, kind="deep"),
}


def run_encode_case(kw):
    s = synth.make_encode_case(**kw)
    idx, codes, ncodes, norms = s["ox"].add_batch_encode(s["x"])
    digest = np.array([int(s["x"].view(np.uint32).astype(np.uint64).sum()), int(s["graph"].links.astype(np.uint64).sum()),
                       int(s["cb"].view(np.uint32).astype(np.uint64).sum())], np.uint64)
    return dict(idx=idx, codes=codes, norm_codes=ncodes, norms=norms, input_digest=digest), s


if __name__ == "__main__":
    for name, kw in ENCODE_CASES.items():
        out, _ = run_encode_case(kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["codes"][:2], out["norm_codes"][:8])
    for name, (kw, nprobe, max_codes, ef, pruning) in CASES.items():
        out, _ = run_case(kw, nprobe, max_codes, ef, pruning)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["counts"], out["lab1"][:4, 0])
