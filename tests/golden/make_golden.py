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


GROUP_CASES = {
    # Grouping construction (IndexIVF_HNSW_Grouping.cpp:43-125): make_encode_case arguments + nsubc
    "add_group_nsubc8_opq": (dict(seed=121, nc=120, d=64, M=8, opq=True, n=1500, hnsw_M=16), 8),
}


def group_inputs(kw, nsubc):
    """Points of the case grouped by a seeded random centroid each (every fifth centroid stays empty)."""
    s = synth.make_encode_case(**kw)
    s["ox"].set_params(1, 0, 40)
    rng = np.random.default_rng(kw["seed"] + 1)
    nc, n = kw["nc"], kw["n"]
    live = np.array([c for c in range(nc) if c % 5], np.uint32)
    key = np.sort(live[rng.integers(0, len(live), size=n)])
    x = (s["cents"][key] + rng.normal(0, 9.0, size=(n, kw["d"]))).astype(np.float32)
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(key, minlength=nc))
    return s, x, offsets


def run_group_case(kw, nsubc):
    s, x, offsets = group_inputs(kw, nsubc)
    nc = kw["nc"]
    nn = np.zeros((nc, nsubc), np.uint32)
    alphas = np.zeros(nc, np.float32)
    sub = np.zeros(len(x), np.uint32)
    codes = np.zeros((len(x), kw["M"]), np.uint8)
    ncodes = np.zeros(len(x), np.uint8)
    for c in range(nc):
        a, b = int(offsets[c]), int(offsets[c + 1])
        nn[c], al, sub[a:b], codes[a:b], ncodes[a:b] = s["ox"].add_group_encode(nsubc, c, x[a:b])
        if b > a:
            alphas[c] = al
    digest = np.array([int(x.view(np.uint32).astype(np.uint64).sum()), int(s["graph"].links.astype(np.uint64).sum())],
                      np.uint64)
    return dict(nn=nn, alphas=alphas, sub=sub, codes=codes, norm_codes=ncodes, input_digest=digest), (s, x, offsets)


if __name__ == "__main__":
    for name, (kw, nsubc) in GROUP_CASES.items():
        out, _ = run_group_case(kw, nsubc)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["alphas"][:6], out["sub"][:10])
    for name, kw in ENCODE_CASES.items():
        out, _ = run_encode_case(kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["codes"][:2], out["norm_codes"][:8])
    for name, (kw, nprobe, max_codes, ef, pruning) in CASES.items():
        out, _ = run_case(kw, nprobe, max_codes, ef, pruning)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["counts"], out["lab1"][:4, 0])
