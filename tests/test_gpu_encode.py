"""GPU parity of the construction side: ivfhnsw_gpu_encode -- the chain
IndexIVF_HNSW::add_batch runs before it appends (IndexIVF_HNSW.cpp:75-121: assign, residual, [OPQ], PQ codes,
decode, [OPQ back], reconstruct, squared norm, norm code) -- against the oracle's restatement: the same
centroid, the same code bytes and the same norm byte for every vector.  Also the class path
(IndexIVF_HNSW::add_batch + write through libivfhnsw.so): the .index file equals the one assembled from the
oracle's output byte for byte.  The faiss leafs inside (compute_codes' distance order) are faiss's published
SSE behaviour, restated on both sides -- parity unpinned against faiss itself.
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


_setup = synth.make_encode_case


def _gpu_encoder(gpu, s):
    g = gpu()
    gr = s["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(s["d"], s["M"], s["cb"], s["nt"], s["A"])
    return g


@pytest.mark.parametrize("d,M,opq,kind", [
    (128, 16, False, "sift"),   # dsub 8: the PQ16 presets
    (128, 16, True, "sift"),
    (128, 8, False, "sift"),    # dsub 16: PQ8 (configs[0])
    (96, 16, True, "deep"),     # dsub 6: DEEP1B + OPQ
    (128, 32, False, "sift"),   # dsub 4
    (80, 8, True, "sift"),      # dsub 10: the generic path
    (48, 4, False, "deep"),     # dsub 12
])
def test_encode_equals_oracle(gpu, d, M, opq, kind):
    s = _setup(200 + d + M, 700, d, M, opq, n=1337, kind=kind)
    g = _gpu_encoder(gpu, s)
    ref_idx, ref_codes, ref_nc, _ = s["ox"].add_batch_encode(s["x"])
    idx, codes, ncodes = g.encode(s["x"], efSearch=40)
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(codes, ref_codes)
    assert np.array_equal(ncodes, ref_nc)
    # the reference's other entry: assignments supplied by the caller (precomputed_idx, :79-80)
    pre = np.random.default_rng(1).integers(0, 700, size=len(s["x"])).astype(np.uint32)
    ref_idx, ref_codes, ref_nc, _ = s["ox"].add_batch_encode(s["x"], pre)
    idx, codes, ncodes = g.encode(s["x"], precomputed_idx=pre)
    assert np.array_equal(idx, pre) and np.array_equal(ref_idx, pre)
    assert np.array_equal(codes, ref_codes)
    assert np.array_equal(ncodes, ref_nc)


def test_equal_code_words_first_one_wins(gpu):
    """compute_code keeps the FIRST minimum (strict '<'): duplicate code words and duplicate norm words."""
    s = _setup(301, 300, 64, 8, False, n=600)
    s["cb"][:, 200] = s["cb"][:, 17]
    s["cb"][:, 5] = s["cb"][:, 90]
    s["nt"][100:110] = s["nt"][100]
    s["ox"] = orc.Index(64, 8, s["graph"], s["cb"], s["nt"], np.zeros(301, np.uint64), np.zeros(0, np.uint32),
                        np.zeros((0, 8), np.uint8), np.zeros(0, np.uint8), np.zeros(300, np.float32))
    s["ox"].set_params(1, 0, 40)
    g = _gpu_encoder(gpu, s)
    # make some vectors land exactly on duplicated words: residual == code word
    x = s["x"].copy()
    pre = np.arange(len(x), dtype=np.uint32) % 300
    for i in range(0, 100):
        x[i] = s["cents"][pre[i]] + np.concatenate([s["cb"][m, 200 if i % 2 else 90] for m in range(8)])
    ref = s["ox"].add_batch_encode(x, pre)
    idx, codes, ncodes = g.encode(x, precomputed_idx=pre)
    assert np.array_equal(codes, ref[1]) and np.array_equal(ncodes, ref[2])
    assert not (codes == 200).any() and not (codes == 90).any()   # 17 and 5 come first
    assert not ((ncodes > 100) & (ncodes < 110)).any()


def test_more_vectors_than_one_internal_chunk(gpu):
    s = _setup(302, 256, 16, 4, True, n=300_000, hnsw_M=6)
    g = _gpu_encoder(gpu, s)
    ref_idx, ref_codes, ref_nc, _ = s["ox"].add_batch_encode(s["x"])
    idx, codes, ncodes = g.encode(s["x"], efSearch=40)
    assert np.array_equal(idx, ref_idx) and np.array_equal(codes, ref_codes) and np.array_equal(ncodes, ref_nc)


def test_encode_state_and_argument_errors(gpu):
    s = _setup(303, 100, 32, 4, False, n=10)
    g = gpu()
    gr = s["graph"]
    g._enc_M = 4
    with pytest.raises(RuntimeError, match="upload_codebooks"):
        g.encode(s["x"], efSearch=10)
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    g.upload_codebooks(32, 4, s["cb"], s["nt"])
    with pytest.raises(RuntimeError, match="efSearch"):
        g.encode(s["x"])                                   # no assignments and no efSearch
    with pytest.raises(RuntimeError, match="out of range"):
        g.encode(s["x"], precomputed_idx=np.full(10, 100, np.uint32))
    idx, codes, ncodes = g.encode(np.zeros((0, 32), np.float32), efSearch=10)
    assert len(idx) == 0 and codes.shape == (0, 4)


@pytest.mark.parametrize("opq,precomputed", [(False, False), (True, False), (False, True)])
def test_class_add_batch_writes_the_oracles_index(tmp_path, opq, precomputed):
    """IndexIVF_HNSW::add_batch in three batches through libivfhnsw.so, then write(): the file equals the index
    assembled from the oracle's codes (lists in order of arrival, ids as given, centroid norms recomputed)."""
    d, M, nc, n = 64, 8, 200, 2500
    s = _setup(304, nc, d, M, opq, n=n, hnsw_M=16)
    # the tool rebuilds the graph from the centroid file with M = 16, efConstruction = 500: build the same here
    graph = orc.Hnsw.build(s["cents"], M=16, efConstruction=500)
    ox = orc.Index(d, M, graph, s["cb"], s["nt"], np.zeros(nc + 1, np.uint64), np.zeros(0, np.uint32),
                   np.zeros((0, M), np.uint8), np.zeros(0, np.uint8), np.zeros(nc, np.float32), opq_A=s["A"])
    ox.set_params(1, 0, 40)
    pre = np.random.default_rng(9).integers(0, nc, size=n).astype(np.uint32) if precomputed else None
    idx, codes, ncodes, _ = ox.add_batch_encode(s["x"], pre)
    order = np.argsort(idx, kind="stable")          # list by list, arrival order inside a list
    offsets = np.zeros(nc + 1, np.uint64)
    offsets[1:] = np.cumsum(np.bincount(idx, minlength=nc))
    ids = (1000 + np.arange(n, dtype=np.uint32))[order]
    want = orc.Index(d, M, graph, s["cb"], s["nt"], offsets, ids, codes[order], ncodes[order], graph.centroid_norms(),
                     opq_A=s["A"])
    p = {k: str(tmp_path / v) for k, v in dict(cent="c.fvecs", info="i", edges="e", pq="pq", npq="npq", opq="opq",
                                               base="b.fvecs", pre="pre.u32", out="out.index", want="want.index").items()}
    want.write(p["want"])
    hostio.write_xvecs(p["cent"], s["cents"])
    hostio.write_pq(p["pq"], d, M, s["cb"])
    hostio.write_pq(p["npq"], 1, 1, s["nt"])
    if opq:
        hostio.write_opq(p["opq"], s["A"])
    hostio.write_xvecs(p["base"], s["x"])
    if precomputed:
        pre.tofile(p["pre"])
    r = subprocess.run([TOOL, "add_batch", str(d), str(nc), str(M), p["cent"], p["info"], p["edges"], p["pq"], p["npq"],
                        p["opq"] if opq else "-", p["base"], p["pre"] if precomputed else "-", str(n), "1000", p["out"]],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(p["out"], "rb").read() == open(p["want"], "rb").read()
