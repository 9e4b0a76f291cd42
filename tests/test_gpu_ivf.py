"""GPU parity: IndexIVF_HNSW::search (IndexIVF_HNSW.cpp:234-296) through the C ABI vs the CPU oracle.

Bar: labels identical and distances bit-identical (the kernels evaluate every float in the oracle's
order).  The coarse stage is supplied by the oracle here (the reference's search2 split,
IndexIVF_HNSW.cpp:453-492); tests/test_gpu_hnsw.py covers the on-device walk.
"""
import numpy as np
import pytest

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu


def _upload(g, c):
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])


def _same(dist, lab, ref_d, ref_l):
    assert np.array_equal(lab, ref_l), "labels differ at queries %s" % np.nonzero((lab != ref_l).any(1))[0][:10]
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32)), "distances are not bit-identical"


@pytest.mark.parametrize("kw,nprobe,max_codes,ef", [
    (dict(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128), 16, 3000, 40),     # PQ16, max_codes bites
    (dict(seed=12, nc=256, d=128, M=8, n_base=20000, nq=64), 8, 10 ** 9, 40),      # PQ8 (config 1 shape)
    (dict(seed=13, nc=128, d=96, M=16, n_base=10000, nq=64), 32, 5000, 64),        # DEEP shape: dsub = 6
    (dict(seed=14, nc=128, d=128, M=32, n_base=8000, nq=32), 4, 1000, 20),         # PQ32
    (dict(seed=15, nc=64, d=64, M=4, n_base=4000, nq=32, efConstruction=60), 64, 10 ** 9, 64),  # every list probed
    # the limit is checked AFTER a list is scored (IndexIVF_HNSW.cpp:290-292): max_codes 0 and 1 still score
    # the first non-empty list, empty lists in front of it are passed over
    (dict(seed=16, nc=128, d=64, M=8, n_base=1500, nq=48, efConstruction=60, empty_frac=0.5), 16, 0, 32),
    (dict(seed=16, nc=128, d=64, M=8, n_base=1500, nq=48, efConstruction=60, empty_frac=0.5), 16, 1, 32),
    (dict(seed=17, nc=512, d=64, M=8, n_base=20000, nq=48, efConstruction=60, empty_frac=0.3), 200, 2500, 210),  # > 64 probes: chunks
])
def test_ivf_top1_matches_oracle(gpu, kw, nprobe, max_codes, ef):
    c = corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, cid, cd, st = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, coarse_ids=cid, coarse_dists=cd)
    _same(dist, lab, ref_d, ref_l)
    ncodes, nseg = g.last_scan_counts()
    assert (ncodes, nseg) == (st.ncode, st.nseg)  # the reference's `ncode` (IndexIVF_HNSW.cpp:290)


def test_single_query_calls_match_batch(gpu):
    """The reference API is one query per call (tests/test_ivfhnsw_sift1b.cpp:193-208): nq = 1 takes the
    split-scan path (several workgroups per query + atomicMin) and must agree with the batch."""
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    ox = synth.oracle_index(c)
    ox.set_params(16, 3000, 40)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    for i in range(0, 24):
        dist, lab = g.search(c["queries"][i], 1, 16, 3000, coarse_ids=cid[i], coarse_dists=cd[i])
        _same(dist, lab, ref_d[i:i + 1], ref_l[i:i + 1])


def test_opq_rotation(gpu):
    c = corpus(seed=21, nc=128, d=128, M=16, n_base=10000, nq=64, opq=True)
    ox = synth.oracle_index(c)
    ox.set_params(8, 4000, 40)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    dist, lab = g.search(c["queries"], 1, 8, 4000, coarse_ids=cid, coarse_dists=cd)
    _same(dist, lab, ref_d, ref_l)


def test_empty_and_padded_probes(gpu):
    """Empty lists are skipped (IndexIVF_HNSW.cpp:271); a query whose probes are all empty returns the
    heapify state FLT_MAX / -1 (:265); 0xffffffff coarse slots are ignored."""
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    sizes = np.diff(c["offsets"].astype(np.int64))
    empty = np.nonzero(sizes == 0)[0]
    assert empty.size >= 2
    g = gpu()
    _upload(g, c)
    nprobe = 4
    cid = np.full((3, nprobe), 0xffffffff, np.uint32)
    cd = np.zeros((3, nprobe), np.float32)
    cid[0, :2] = empty[:2]                      # only empty lists
    full = np.nonzero(sizes > 0)[0]
    cid[1] = [empty[0], full[0], empty[1], full[1]]
    cd[1] = [1.0, 2.0, 3.0, 4.0]
    # query 2: nothing but padding
    q = c["queries"][:3]
    dist, lab = g.search(q, 1, nprobe, 10 ** 9, coarse_ids=cid, coarse_dists=cd)
    assert lab[0, 0] == -1 and dist[0, 0] == np.finfo(np.float32).max
    assert lab[2, 0] == -1 and dist[2, 0] == np.finfo(np.float32).max
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, 10 ** 9, 40)
    rd, rl, _ = ox.search_coarse(q[1], cid[1], cd[1])
    assert lab[1, 0] == rl[0] and dist[1, 0] == rd[0]


def test_ties_first_scanned_wins(gpu):
    """Strict '<' (IndexIVF_HNSW.cpp:285): among equal distances the earliest scanned code is kept.
    Every code of every list is made identical, so all distances within a list tie."""
    c = dict(corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128))
    c["codes"] = np.zeros_like(c["codes"])
    c["norm_codes"] = np.zeros_like(c["norm_codes"])
    ox = synth.oracle_index(c)
    ox.set_params(8, 10 ** 9, 40)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"][:32], k=1)
    g = gpu()
    _upload(g, c)
    dist, lab = g.search(c["queries"][:32], 1, 8, 10 ** 9, coarse_ids=cid, coarse_dists=cd)
    _same(dist, lab, ref_d, ref_l)
    # and it is the first id of the first probed non-empty list among the minimal ones
    off = c["offsets"].astype(np.int64)
    for i in range(32):
        heads = [c["ids"][off[l]] for l in cid[i] if off[l + 1] > off[l]]
        assert lab[i, 0] in heads  # the first element of one of the probed lists


def test_synthetic_device_corpus_matches_host_stream(gpu):
    """ivfhnsw_gpu_upload_ivf_synthetic generates codes on the device; tests/synth.py reproduces the
    same byte stream, so the oracle can check searches over it."""
    t = synth.make_throughput_tables(seed=5, nc=512, d=128, M=16, n_total=200000)
    ids, codes, norm_codes = synth.synthetic_codes(99, t["offsets"], 16)
    gr = synth.orc.Hnsw.build(t["centroids"], M=16, efConstruction=60)
    t.update(ids=ids, codes=codes, norm_codes=norm_codes, centroid_norms=gr.centroid_norms(), graph=gr)
    ox = synth.oracle_index(t)
    ox.set_params(16, 5000, 40)
    rng = np.random.default_rng(3)
    q = (t["centroids"][rng.choice(512, 64)] + rng.normal(0, 10, size=(64, 128))).astype(np.float32)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(q, k=1)
    g = gpu()
    g.upload_ivf_synthetic(128, 16, t["offsets"], t["centroid_norms"], t["pq_centroids"], t["norm_table"], seed=99)
    dist, lab = g.search(q, 1, 16, 5000, coarse_ids=cid, coarse_dists=cd)
    _same(dist, lab, ref_d, ref_l)


@pytest.mark.parametrize("d,M,nq", [(128, 16, 77), (96, 8, 33), (64, 8, 64), (48, 4, 19)])
def test_opq_rotation_mfma_and_scalar_forms(gpu, d, M, nq):
    """OPQ rotation (IndexIVF_HNSW.cpp:240): d % 32 == 0 takes the MFMA kernel (v_mfma_f32_32x32x2_f32 = k-ordered
    fmaf chain), other d the scalar kernel; both must reproduce the oracle's fmaf chain bit for bit, including
    batches that are not a multiple of the 32-query tile.  Checked through the whole search (a wrong rotation
    changes the table and the distances)."""
    c = corpus(seed=90 + d, nc=64, d=d, M=M, n_base=3000, nq=nq, opq=True, efConstruction=50)
    ox = synth.oracle_index(c)
    ox.set_params(8, 10 ** 9, 32)
    ref_d, ref_l, cid, cd, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    dist, lab = g.search(c["queries"], 1, 8, 10 ** 9, coarse_ids=cid, coarse_dists=cd)
    _same(dist, lab, ref_d, ref_l)
