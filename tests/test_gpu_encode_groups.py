"""GPU parity of Grouping construction: ivfhnsw_gpu_encode_groups -- IndexIVF_HNSW_Grouping::
add_group up to its distribution loops (IndexIVF_HNSW_Grouping.cpp:43-125: neighbour centroids, alpha, sub-centroid
of every point, codes against the sub-centroids) -- against the oracle's restatement, group by group: same
neighbours, bit-identical alpha, same sub-centroid, code bytes and norm byte for every point.  And the class path:
the Grouping .index written after one add_group per centroid equals the oracle-assembled file byte for byte.
"""
import os
import subprocess

import numpy as np
import pytest

import hostio
import synth
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "cpp", "hostlib_tool.bin")


def _grouped(s, nc, seed, empty_frac=0.15):
    """Points of s["x"] grouped by a random centroid each; a share of the centroids gets no points."""
    rng = np.random.default_rng(seed)
    n = len(s["x"])
    live = rng.permutation(nc)[:max(1, int(nc * (1 - empty_frac)))]
    pre = live[rng.integers(0, len(live), size=n)].astype(np.uint32)
    # realistic groups: move every point next to its centroid
    x = (s["cents"][pre] + (s["x"] - s["cents"][rng.integers(0, nc, size=n)]) * 0.3).astype(np.float32)
    order = np.argsort(pre, kind="stable")
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(pre, minlength=nc))
    return pre, x, order, offsets


def _oracle_groups(ox, nsubc, nc, x_sorted, offsets):
    out = []
    for c in range(nc):
        out.append(ox.add_group_encode(nsubc, c, x_sorted[int(offsets[c]):int(offsets[c + 1])]))
    return out


@pytest.mark.parametrize("d,M,nsubc,opq,kind", [
    (128, 16, 64, False, "sift"),   # the reference's preset: nsubc 64 (run_sift1b_grouping_OPQ.sh)
    (128, 16, 8, True, "sift"),
    (96, 8, 16, True, "deep"),
    (32, 4, 5, False, "sift"),      # nsubc not a multiple of 4: uneven shares per thread
])
def test_encode_groups_equals_oracle(gpu, d, M, nsubc, opq, kind):
    nc = 150
    s = synth.make_encode_case(500 + d + nsubc, nc, d, M, opq, n=3000, hnsw_M=16, kind=kind)
    s["ox"].set_params(1, 0, 80)
    pre, x, order, offsets = _grouped(s, nc, 7)
    xs = np.ascontiguousarray(x[order])
    ref = _oracle_groups(s["ox"], nsubc, nc, xs, offsets)
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(d, M, s["cb"], s["nt"], s["A"])
    marker = np.full(nc, 123.0, np.float32)
    nn, alphas, sub, codes, ncodes = g.encode_groups(nsubc, np.arange(nc, dtype=np.uint32), offsets, xs, 80,
                                                     alphas_in=marker)
    for c in range(nc):
        a, b = int(offsets[c]), int(offsets[c + 1])
        rnn, ralpha, rsub, rcodes, rnc = ref[c]
        assert np.array_equal(nn[c], rnn), c
        if a == b:
            assert alphas[c] == 123.0          # an empty group keeps the caller's value (Grouping.cpp:63-64)
            continue
        assert np.float32(alphas[c]).view(np.uint32) == np.float32(ralpha).view(np.uint32), (c, alphas[c], ralpha)
        assert np.array_equal(sub[a:b], rsub), c
        assert np.array_equal(codes[a:b], rcodes), c
        assert np.array_equal(ncodes[a:b], rnc), c
    # one group per call (what add_group does) gives the same bytes as the batched call
    for c in (0, 17, 99):
        a, b = int(offsets[c]), int(offsets[c + 1])
        one = g.encode_groups(nsubc, np.array([c], np.uint32), np.array([0, b - a], np.uint64), xs[a:b], 80)
        assert np.array_equal(one[0][0], nn[c]) and np.array_equal(one[2], sub[a:b])
        assert np.array_equal(one[3], codes[a:b]) and np.array_equal(one[4], ncodes[a:b])


def test_a_group_larger_than_one_tile_and_chunk_boundaries(gpu):
    """One group of 5000 points (many 64-point tiles) between small ones, d 16 so that the oracle stays quick."""
    nc, d, M, nsubc = 64, 16, 4, 6
    s = synth.make_encode_case(601, nc, d, M, False, n=6000, hnsw_M=8)
    s["ox"].set_params(1, 0, 40)
    rng = np.random.default_rng(3)
    pre = np.concatenate([np.full(5000, 10), rng.integers(0, nc, size=1000)]).astype(np.uint32)
    x = (s["cents"][pre] + rng.normal(0, 9.0, size=(6000, d))).astype(np.float32)
    order = np.argsort(pre, kind="stable")
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(pre, minlength=nc))
    xs = np.ascontiguousarray(x[order])
    ref = _oracle_groups(s["ox"], nsubc, nc, xs, offsets)
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(d, M, s["cb"], s["nt"])
    nn, alphas, sub, codes, ncodes = g.encode_groups(nsubc, np.arange(nc, dtype=np.uint32), offsets, xs, 40)
    for c in range(nc):
        a, b = int(offsets[c]), int(offsets[c + 1])
        assert np.array_equal(nn[c], ref[c][0])
        if b > a:
            assert np.float32(alphas[c]).view(np.uint32) == np.float32(ref[c][1]).view(np.uint32)
            assert np.array_equal(sub[a:b], ref[c][2]) and np.array_equal(codes[a:b], ref[c][3])
            assert np.array_equal(ncodes[a:b], ref[c][4])


def test_encode_groups_argument_errors(gpu):
    s = synth.make_encode_case(602, 40, 32, 4, False, n=10, hnsw_M=6)
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(32, 4, s["cb"], s["nt"])
    off = np.array([0, 10], np.uint64)
    with pytest.raises(RuntimeError, match="efSearch"):
        g.encode_groups(8, np.array([3], np.uint32), off, s["x"], 8)          # needs efSearch >= nsubc + 1
    with pytest.raises(RuntimeError, match="out of range"):
        g.encode_groups(4, np.array([40], np.uint32), off, s["x"], 16)
    with pytest.raises(RuntimeError, match="centroids"):
        g.encode_groups(45, np.array([3], np.uint32), off, s["x"], 64)        # more sub-centroids than centroids


@pytest.mark.parametrize("opq", [False, True])
def test_class_add_group_writes_the_oracles_grouping_index(tmp_path, opq):
    d, M, nc, nsubc, n = 64, 8, 120, 8, 2400
    s = synth.make_encode_case(603, nc, d, M, opq, n=n, hnsw_M=16)
    graph = orc.Hnsw.build(s["cents"], M=16, efConstruction=500)   # what the tool's build_quantizer constructs
    ox = orc.Index(d, M, graph, s["cb"], s["nt"], np.zeros(nc + 1, np.uint64), np.zeros(0, np.uint32),
                   np.zeros((0, M), np.uint8), np.zeros(0, np.uint8), np.zeros(nc, np.float32), opq_A=s["A"])
    ox.set_params(1, 0, 40)
    pre, x, order, offsets = _grouped(s, nc, 11)
    xs = np.ascontiguousarray(x[order])
    ids_sorted = (1000 + np.arange(n, dtype=np.uint32))[order]
    nn = np.zeros((nc, nsubc), np.uint32)
    alphas = np.zeros(nc, np.float32)
    sizes = np.zeros((nc, nsubc), np.uint32)
    all_ids, all_codes, all_nc = [], [], []
    for c in range(nc):
        a, b = int(offsets[c]), int(offsets[c + 1])
        rnn, ralpha, rsub, rcodes, rnc = ox.add_group_encode(nsubc, c, xs[a:b])
        nn[c] = rnn
        if b == a:
            continue
        alphas[c] = ralpha
        o = np.argsort(rsub, kind="stable")            # sub-group by sub-group, arrival order inside (:127-155)
        sizes[c] = np.bincount(rsub, minlength=nsubc)
        all_ids.append(ids_sorted[a:b][o])
        all_codes.append(rcodes[o])
        all_nc.append(rnc[o])
    want = orc.Index(d, M, graph, s["cb"], s["nt"], offsets, np.concatenate(all_ids), np.concatenate(all_codes),
                     np.concatenate(all_nc), graph.centroid_norms(), opq_A=s["A"], nsubc=nsubc, alphas=alphas,
                     nn_centroid_idxs=nn, subgroup_sizes=sizes, inter_centroid_dists=graph.inter_centroid_dists(nn))
    p = {k: str(tmp_path / v) for k, v in dict(cent="c.fvecs", info="i", edges="e", pq="pq", npq="npq", opq="opq",
                                               base="b.fvecs", pre="pre.u32", out="out.index", want="want.index").items()}
    want.write(p["want"])
    hostio.write_xvecs(p["cent"], s["cents"])
    hostio.write_pq(p["pq"], d, M, s["cb"])
    hostio.write_pq(p["npq"], 1, 1, s["nt"])
    if opq:
        hostio.write_opq(p["opq"], s["A"])
    hostio.write_xvecs(p["base"], x)
    pre.tofile(p["pre"])
    r = subprocess.run([TOOL, "add_group", str(d), str(nc), str(M), str(nsubc), p["cent"], p["info"], p["edges"], p["pq"],
                        p["npq"], p["opq"] if opq else "-", p["base"], p["pre"], str(n), p["out"]],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(p["out"], "rb").read() == open(p["want"], "rb").read()
