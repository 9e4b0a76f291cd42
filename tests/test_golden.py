"""Stored expectations (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

Not reference outputs -- the reference ships no fixtures and cannot be run here (parity unpinned).  They pin the
oracle and the corpus generator across rounds (CPU test) and give the device path a second expectation (GPU test).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

NAMES = sorted(make_golden.CASES)


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    out, _ = make_golden.run_case(*make_golden.CASES[name])
    g = _load(name)
    assert np.array_equal(out["input_digest"], g["input_digest"]), "the corpus generator changed"
    for key in ("lab1", "lab5", "coarse_ids", "counts"):
        assert np.array_equal(out[key], g[key]), key
    for key in ("dist1", "dist5", "coarse_dists"):
        assert np.array_equal(out[key].view(np.uint32), g[key].view(np.uint32)), key


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_reproduces_golden(gpu, name):
    kw, nprobe, max_codes, ef, pruning = make_golden.CASES[name]
    import synth
    c = synth.make_corpus(**kw)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    if c["nsubc"]:
        g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                          c["inter_centroid_dists"])
    gold = _load(name)
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    assert np.array_equal(lab, gold["lab1"])
    assert np.array_equal(dist.view(np.uint32), gold["dist1"].view(np.uint32))
    assert g.last_scan_counts() == (int(gold["counts"][0]), int(gold["counts"][1]))
    # the coarse stage alone (rotated queries when OPQ is on)
    xq = c["queries"]
    if c["opq_A"] is not None:
        from oracle import orc
        xq = np.stack([orc.opq_apply(c["opq_A"], x) for x in xq])
    ids, cd = g.coarse(xq, nprobe, ef)
    assert np.array_equal(ids, gold["coarse_ids"])
    assert np.array_equal(cd.view(np.uint32), gold["coarse_dists"].view(np.uint32))
    dist5, lab5 = g.search(c["queries"], 5, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    for i in range(len(lab5)):
        o1, o2 = np.lexsort((lab5[i], dist5[i])), np.lexsort((gold["lab5"][i], gold["dist5"][i]))
        assert np.array_equal(lab5[i][o1], gold["lab5"][i][o2])
        assert np.array_equal(dist5[i][o1].view(np.uint32), gold["dist5"][i][o2].view(np.uint32))


# ---------------------------------------------------------------------------------------------- construction side
ENC_NAMES = sorted(make_golden.ENCODE_CASES)


@pytest.mark.parametrize("name", ENC_NAMES)
def test_oracle_reproduces_encode_golden(name):
    out, _ = make_golden.run_encode_case(make_golden.ENCODE_CASES[name])
    g = _load(name)
    assert np.array_equal(out["input_digest"], g["input_digest"]), "the case generator changed"
    for key in ("idx", "codes", "norm_codes"):
        assert np.array_equal(out[key], g[key]), key
    assert np.array_equal(out["norms"].view(np.uint32), g["norms"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENC_NAMES)
def test_device_reproduces_encode_golden(gpu, name):
    import synth
    s = synth.make_encode_case(**make_golden.ENCODE_CASES[name])
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(s["d"], s["M"], s["cb"], s["nt"], s["A"])
    gold = _load(name)
    idx, codes, ncodes = g.encode(s["x"], efSearch=40)
    assert np.array_equal(idx, gold["idx"])
    assert np.array_equal(codes, gold["codes"])
    assert np.array_equal(ncodes, gold["norm_codes"])


GROUP_NAMES = sorted(make_golden.GROUP_CASES)


@pytest.mark.parametrize("name", GROUP_NAMES)
def test_oracle_reproduces_add_group_golden(name):
    out, _ = make_golden.run_group_case(*make_golden.GROUP_CASES[name])
    g = _load(name)
    assert np.array_equal(out["input_digest"], g["input_digest"]), "the case generator changed"
    for key in ("nn", "sub", "codes", "norm_codes"):
        assert np.array_equal(out[key], g[key]), key
    assert np.array_equal(out["alphas"].view(np.uint32), g["alphas"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", GROUP_NAMES)
def test_device_reproduces_add_group_golden(gpu, name):
    kw, nsubc = make_golden.GROUP_CASES[name]
    s, x, offsets = make_golden.group_inputs(kw, nsubc)
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(s["d"], s["M"], s["cb"], s["nt"], s["A"])
    gold = _load(name)
    nn, alphas, sub, codes, ncodes = g.encode_groups(nsubc, np.arange(kw["nc"], dtype=np.uint32), offsets, x, 40)
    assert np.array_equal(nn, gold["nn"])
    assert np.array_equal(alphas.view(np.uint32), gold["alphas"].view(np.uint32))
    assert np.array_equal(sub, gold["sub"]) and np.array_equal(codes, gold["codes"])
    assert np.array_equal(ncodes, gold["norm_codes"])
