"""The host-side class surface (include/ivf-hnsw/*.h + libivfhnsw.so): file formats against the oracle's
independent reader/writer (CPU), host graph construction/walk against the oracle (CPU), search()/search_batch()
through the classes against the oracle (GPU), and -- when the reference tree is present at build time -- the
reference's OWN driver binary, built unchanged against these headers, run end to end (GPU)."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import hostio
import synth
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "cpp", "hostlib_tool.bin")
REF_DRV = os.path.join(ROOT, "oracle", "_ref")


def tool(*args, extra_env=None):
    assert os.path.exists(TOOL), "run __graft_entry__.build()"
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("OMP_NUM_THREADS", str(min(8, len(os.sched_getaffinity(0)))))  # a CPU quota makes wide teams crawl
    env.setdefault("OMP_WAIT_POLICY", "passive")
    r = subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def small(**kw):
    base = dict(seed=61, nc=64, d=32, M=4, n_base=2500, nq=16, efConstruction=40, empty_frac=0.2)
    base.update(kw)
    return synth.make_corpus(**base)


# ---------------------------------------------------------------------------------------------- formats (CPU)
@pytest.mark.parametrize("nsubc", [0, 4])
def test_index_file_written_by_oracle_roundtrips_through_the_class(tmp_path, nsubc):
    c = small(nsubc=nsubc)
    src, dst = str(tmp_path / "a.index"), str(tmp_path / "b.index")
    synth.oracle_index(c).write(src)
    tool("index_roundtrip", "grouping" if nsubc else "ivf", c["d"], c["nc"], c["code_size"], nsubc, src, dst)
    assert open(src, "rb").read() == open(dst, "rb").read()


def test_hnsw_files_roundtrip_and_host_construction_equals_oracle(tmp_path):
    rng = np.random.default_rng(4)
    cents = synth.sift_like(rng, 300, 32)
    g = orc.Hnsw.build(cents, M=6, efConstruction=50)
    pi, pe, pd = (str(tmp_path / n) for n in ("i", "e", "c.fvecs"))
    g.save(pi, pe)
    hostio.write_xvecs(pd, cents)
    tool("hnsw_roundtrip", pi, pd, pe, pi + "2", pe + "2")
    assert open(pi, "rb").read() == open(pi + "2", "rb").read()
    assert open(pe, "rb").read() == open(pe + "2", "rb").read()
    # serial construction on the host reproduces the oracle's graph link for link
    tool("hnsw_build", pd, 300, 32, 6, 50, pi + "3", pe + "3")
    assert open(pi, "rb").read() == open(pi + "3", "rb").read()
    assert open(pe, "rb").read() == open(pe + "3", "rb").read()


@pytest.mark.gpu
def test_build_quantizer_on_the_device_equals_the_serial_loop_over_exact_candidates(tmp_path):
    """IVFHNSW_BUILD=device: IndexIVF_HNSW::build_quantizer through ivfhnsw_gpu_build_graph -- the files it saves are the
    ones the oracle's serial insertion loop over exact candidates leaves (M 16, maxM 32, 64 candidates); without the
    variable it stays the reference-identical serial construction."""
    rng = np.random.default_rng(14)
    cents = synth.clustered_centroids(rng, 3000, 128)
    pd = str(tmp_path / "c.fvecs")
    hostio.write_xvecs(pd, cents)
    env = dict(os.environ, IVFHNSW_BUILD="device")
    subprocess.run([TOOL, "build_quantizer", pd, "3000", "128", "16", "100", str(tmp_path / "i_dev"), str(tmp_path / "e_dev")],
                   check=True, env=env, capture_output=True)
    ref = orc.Hnsw.build_exact(cents, 16, 32, 64)
    ref.save(str(tmp_path / "i_ref"), str(tmp_path / "e_ref"))
    assert open(tmp_path / "e_dev", "rb").read() == open(tmp_path / "e_ref", "rb").read()
    tool("build_quantizer", pd, 3000, 128, 16, 100, str(tmp_path / "i_ser"), str(tmp_path / "e_ser"))
    ser = orc.Hnsw.build(cents, M=16, efConstruction=100)
    ser.save(str(tmp_path / "i_ser2"), str(tmp_path / "e_ser2"))
    assert open(tmp_path / "e_ser", "rb").read() == open(tmp_path / "e_ser2", "rb").read()


def test_host_walk_equals_oracle(tmp_path):
    rng = np.random.default_rng(5)
    base = synth.sift_like(rng, 150, 32)
    cents = np.concatenate([base, base[:40]])  # duplicates: exact ties
    g = orc.Hnsw.build(cents, M=6, efConstruction=50)
    pi, pe, pd, pq = (str(tmp_path / n) for n in ("i", "e", "c.fvecs", "q.fvecs"))
    g.save(pi, pe)
    hostio.write_xvecs(pd, cents)
    q = (base[rng.choice(150, 20)] + rng.normal(0, 4, (20, 32))).astype(np.float32)
    q[:4] = base[:4]
    hostio.write_xvecs(pq, q)
    for ef, k in [(8, 8), (30, 10), (64, 64)]:
        out = str(tmp_path / "o.bin")
        tool("hnsw_search", pi, pd, pe, pq, 20, ef, k, out)
        raw = np.fromfile(out, np.uint8)
        ids = raw[:20 * k * 4].view(np.uint32).reshape(20, k)
        dist = raw[20 * k * 4:].view(np.float32).reshape(20, k)
        for i in range(20):
            rid, rd = g.search_knn(q[i], ef, k)
            assert np.array_equal(ids[i, :len(rid)], rid)
            assert np.array_equal(dist[i, :len(rid)].view(np.uint32), rd.view(np.uint32))


def test_pq_and_opq_files_roundtrip(tmp_path):
    rng = np.random.default_rng(6)
    a, b = str(tmp_path / "pq"), str(tmp_path / "pq2")
    hostio.write_pq(a, 32, 4, rng.normal(size=(4, 256, 8)).astype(np.float32))
    tool("pq_roundtrip", a, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    a, b = str(tmp_path / "opq"), str(tmp_path / "opq2")
    hostio.write_opq(a, synth.random_rotation(rng, 32))
    tool("vt_roundtrip", a, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # a truncated / foreign file is refused loudly
    open(a, "wb").write(b"nope")
    r = subprocess.run([TOOL, "vt_roundtrip", a, b], capture_output=True, text=True)
    assert r.returncode == 1 and "LTra" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="reference tree not present (GPU box)")
def test_reference_drivers_compile_unchanged_against_these_headers():
    """Source compatibility of the drop-in surface: every driver of the reference parses AND links."""
    import glob
    drivers = sorted(glob.glob("/root/reference/tests/*.cpp"))
    assert len(drivers) == 12
    for f in drivers:
        r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-fopenmp", "-I" + os.path.join(ROOT, "include"), f],
                           capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr[-1500:])


# ---------------------------------------------------------------------------------------------- search (GPU)
def _class_search(tmp_path, c, nprobe, max_codes, ef, pruning, k=1, extra_env=None):
    p = hostio.dump_corpus(c, str(tmp_path))
    out = str(tmp_path / "res.bin")
    nq = len(c["queries"])
    tool("search", "grouping" if c["nsubc"] else "ivf", c["d"], c["nc"], c["code_size"], c["nsubc"], p["centroids"],
         p["info"], p["edges"], p["pq"], p["norm_pq"], p["opq"], p["index"], p["queries"], nq, k, nprobe, max_codes, ef,
         int(pruning), out, extra_env=extra_env)
    raw = np.fromfile(out, np.uint8)
    lab = raw[:2 * nq * k * 8].view(np.int64).reshape(2, nq, k)
    dist = raw[2 * nq * k * 8:].view(np.float32).reshape(2, nq, k)
    return lab, dist


@pytest.mark.gpu
@pytest.mark.parametrize("kw,nprobe,max_codes,ef,pruning", [
    (dict(seed=71, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80), 8, 1500, 32, False),
    (dict(seed=72, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80, opq=True), 8, 1500, 32, False),
    (dict(seed=73, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80, nsubc=8), 8, 700, 40, True),
    (dict(seed=74, nc=128, d=96, M=8, n_base=6000, nq=32, efConstruction=80, nsubc=8, opq=True), 8, 900, 40, False),
])
def test_class_search_equals_oracle(tmp_path, kw, nprobe, max_codes, ef, pruning):
    """The drivers' load sequence (build_quantizer from files, read_ProductQuantizer, read, rotate_quantizer) then
    search() one query per call and search_batch(): labels and distances identical to the oracle."""
    # the corpus generator rotates the graph in place for OPQ; the centroid FILE must hold unrotated centroids
    c = synth.make_corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=pruning)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    lab, dist = _class_search(tmp_path, c, nprobe, max_codes, ef, pruning)
    for mode in (0, 1):  # per-query calls, batch
        assert np.array_equal(lab[mode], ref_l)
        assert np.array_equal(dist[mode].view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.gpu
def test_class_search_k10_returns_the_reference_heap_array(tmp_path):
    c = synth.make_corpus(seed=71, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80)
    ox = synth.oracle_index(c)
    ox.set_params(8, 1500, 32)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=10)
    lab, dist = _class_search(tmp_path, c, 8, 1500, 32, False, k=10)
    for mode in (0, 1):  # search() per query, search_batch(): both element for element what faiss's heap leaves
        assert np.array_equal(lab[mode], ref_l)
        assert np.array_equal(dist[mode].view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 5])
@pytest.mark.parametrize("kw,nprobe,max_codes,ef,pruning,k", [
    (dict(seed=72, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80, opq=True), 8, 1500, 32, False, 1),
    (dict(seed=73, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80, nsubc=8), 8, 700, 40, True, 1),
    (dict(seed=71, nc=128, d=128, M=16, n_base=8000, nq=32, efConstruction=80), 8, 1500, 32, False, 10),
    (dict(seed=74, nc=128, d=96, M=8, n_base=6000, nq=32, efConstruction=80, nsubc=8, opq=True), 8, 900, 40, False, 7),
])
def test_class_search_on_list_shards_equals_oracle(tmp_path, shards, kw, nprobe, max_codes, ef, pruning, k):
    """IVFHNSW_SHARDS=N: the classes split the lists list-wise over N device handles (one per GPU of a node; here all
    on GPU 0) and every search() / search_batch() is the shard step of SURVEY 8e merged on the host -- k = 1 by the
    smallest key, k > 1 by replaying faiss's heap over the shards' candidate streams merged in scan order.  The
    reference's drivers reach the multi-GPU split through this without a line changed."""
    c = synth.make_corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=pruning)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=k)
    lab, dist = _class_search(tmp_path, c, nprobe, max_codes, ef, pruning, k=k, extra_env={"IVFHNSW_SHARDS": str(shards)})
    for mode in (0, 1):  # search() per query, search_batch()
        assert np.array_equal(lab[mode], ref_l)
        assert np.array_equal(dist[mode].view(np.uint32), ref_d.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("opq,k,shards", [(False, 1, 1), (True, 1, 1), (False, 5, 1), (True, 5, 3)])
def test_class_sibling_entry_points_equal_oracle(tmp_path, opq, k, shards):
    """search_debug, search_enn, search2 and search2m (IndexIVF_HNSW.cpp:328-534) through the class surface.
    search_debug == search; search_enn == the search with nprobe 1 and k 1; search2 == the search on the caller's
    coarse stage (here the HOST walk of the class's quantizer, which must equal the oracle's walk); search2m == one
    heap per probe, each list alone (the reference compares against heap 0 from racing OpenMP threads,
    IndexIVF_HNSW.cpp:523 -- undefined; the per-probe result is the defined part)."""
    c = synth.make_corpus(seed=75, nc=128, d=128, M=16, n_base=8000, nq=24, efConstruction=80, opq=opq)
    nprobe, max_codes, ef = 6, 1200, 32
    nq = len(c["queries"])
    p = hostio.dump_corpus(c, str(tmp_path))
    out = str(tmp_path / "sib.bin")
    tool("siblings", "ivf", c["d"], c["nc"], c["code_size"], 0, p["centroids"], p["info"], p["edges"], p["pq"],
         p["norm_pq"], p["opq"], p["index"], p["queries"], nq, k, nprobe, max_codes, ef, 0, out,
         extra_env={"IVFHNSW_SHARDS": str(shards)})
    raw = np.fromfile(out, np.uint8)
    n_lab = nq * k + nq + nq * k + nq * nprobe * k
    lab = raw[:n_lab * 8].view(np.int64)
    dist = raw[n_lab * 8:n_lab * 12].view(np.float32)
    rest = raw[n_lab * 12:]
    enn_c = rest[:nq * 4].view(np.uint32)
    cids = rest[nq * 4:nq * 4 + nq * nprobe * 4].view(np.uint32).reshape(nq, nprobe)
    cds = rest[nq * 4 + nq * nprobe * 4:].view(np.float32).reshape(nq, nprobe)

    def cut(a):
        o = [0, nq * k, nq * k + nq, 2 * nq * k + nq, n_lab]
        return (a[o[0]:o[1]].reshape(nq, k), a[o[1]:o[2]].reshape(nq, 1), a[o[2]:o[3]].reshape(nq, k),
                a[o[3]:o[4]].reshape(nq, nprobe, k))

    l_dbg, l_enn, l_s2, l_s2m = cut(lab)
    d_dbg, d_enn, d_s2, d_s2m = cut(dist)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, ref_cid, ref_cd, _ = ox.search_batch(c["queries"], k=k)
    same = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
    assert np.array_equal(l_dbg, ref_l) and same(d_dbg, ref_d)
    assert np.array_equal(cids, ref_cid) and same(cds, ref_cd)      # the host walk of the class == the oracle's
    assert np.array_equal(l_s2, ref_l) and same(d_s2, ref_d)
    ox.set_params(1, max_codes, ef)
    e_d, e_l, e_cid, _, _ = ox.search_batch(c["queries"], k=1)
    assert np.array_equal(l_enn, e_l) and same(d_enn, e_d) and np.array_equal(enn_c, e_cid[:, 0])
    ox.set_params(1, 2 ** 62, ef)
    for i in range(nq):
        for j in range(nprobe):
            pd, pl, _ = ox.search_coarse(c["queries"][i], ref_cid[i, j:j + 1], ref_cd[i, j:j + 1], k=k)
            assert np.array_equal(l_s2m[i, j], pl) and same(d_s2m[i, j], pd)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DRV, "test_ivfhnsw_deep1b")),
                    reason="reference drivers not built (make -C oracle ref_drivers needs /root/reference)")
@pytest.mark.parametrize("driver,opq", [("test_ivfhnsw_deep1b", False), ("test_ivfhnsw_deep1b", True),
                                        ("test_ivfhnsw_grouping_deep1b", False)])
def test_reference_driver_binary_runs_on_this_library(tmp_path, driver, opq):
    """The reference's own driver (built unchanged, oracle/Makefile ref_drivers) loads synthetic files in the
    reference's formats and searches through this repo's device path: the Recall@1 it prints must be the recall of
    the oracle's labels."""
    grouping = "grouping" in driver
    c = synth.make_corpus(seed=81, nc=128, d=96, M=8, n_base=8000, nq=64, efConstruction=80, opq=opq,
                          nsubc=8 if grouping else 0)
    nprobe, max_codes, ef = 8, 1200, 32
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
    _, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    gt = ((c["queries"][:, None, :] - c["base"][None, :, :]) ** 2).sum(2).argsort(1)[:, :5].astype(np.int32)
    want = float((ref_l[:, 0] == gt[:, 0]).mean())
    p = hostio.dump_corpus(c, str(tmp_path))
    hostio.write_xvecs(str(tmp_path / "gt.ivecs"), gt)
    open(tmp_path / "precomputed_idxs.ivecs", "wb").close()
    args = ["-M", 16, "-efConstruction", 80, "-nb", 8000, "-nt", 1000, "-nsubt", 1000, "-nc", c["nc"], "-nq", 64,
            "-ngt", 5, "-d", 96, "-code_size", 8, "-opq", "on" if opq else "off", "-k", 1, "-nprobe", nprobe,
            "-max_codes", max_codes, "-efSearch", ef, "-path_base", "unused", "-path_learn", "unused",
            "-path_q", p["queries"], "-path_gt", str(tmp_path / "gt.ivecs"), "-path_centroids", p["centroids"],
            "-path_precomputed_idx", str(tmp_path / "precomputed_idxs.ivecs"), "-path_info", p["info"],
            "-path_edges", p["edges"], "-path_pq", p["pq"], "-path_opq_matrix", p["opq"], "-path_norm_pq", p["norm_pq"],
            "-path_index", p["index"]]
    if grouping:
        args += ["-nsubc", 8, "-pruning", "on"]
    r = subprocess.run([os.path.join(REF_DRV, driver)] + [str(a) for a in args], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    m = re.search(r"Recall@1: ([0-9.eE+-]+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert abs(float(m.group(1)) - want) < 1e-6, (m.group(1), want)


# ---------------------------------------------------------------------------------------------- training
@pytest.mark.gpu
def test_opq_training_gives_a_rotation_that_helps(tmp_path):
    """OPQMatrix::train (Lloyd iterations and X^T Y on the device, the d x d SVD on the host): on points whose variance is mixed across sub-spaces the learnt matrix must be
    orthonormal and the product quantizer trained behind it must beat the one trained on the raw points."""
    rng = np.random.default_rng(31)
    d, M, n = 32, 4, 4000
    z = rng.normal(size=(n, d)) * np.linspace(6.0, 0.2, d)        # strongly unequal variances ...
    mix = synth.random_rotation(rng, d)
    x = (z @ mix.T).astype(np.float32)                            # ... smeared over all coordinates
    px, pv = str(tmp_path / "x.fvecs"), str(tmp_path / "opq.vt")
    hostio.write_xvecs(px, x)
    r = tool("opq_train", d, M, n, px, 12, pv)
    err = dict(l.split() for l in r.stdout.splitlines() if l.startswith("err_"))
    assert float(err["err_opq"]) < 0.9 * float(err["err_plain"]), r.stdout
    raw = open(pv, "rb").read()
    assert raw[:4] == b"LTra"
    cnt = struct.unpack("<Q", raw[5:13])[0]
    assert cnt == d * d
    A = np.frombuffer(raw[13:13 + 4 * d * d], np.float32).reshape(d, d).astype(np.float64)
    assert np.abs(A @ A.T - np.eye(d)).max() < 1e-4


@pytest.mark.gpu
def test_grouping_train_pq_writes_usable_code_books(tmp_path):
    """IndexIVF_HNSW_Grouping::train_pq (assignment and the Lloyd iterations on the device): the residual code book
    must quantise the training residuals far better than the points themselves, the norm code book must be 256
    finite values around the squared norms."""
    rng = np.random.default_rng(32)
    d, M, nc, nsubc, n = 32, 4, 64, 4, 3000
    cents = synth.sift_like(rng, nc, d)
    x = (cents[rng.integers(0, nc, size=n)] + rng.normal(0, 6.0, size=(n, d))).astype(np.float32)
    p = {k: str(tmp_path / v) for k, v in dict(cent="c.fvecs", info="i", edges="e", x="x.fvecs", pq="pq", npq="npq").items()}
    hostio.write_xvecs(p["cent"], cents)
    hostio.write_xvecs(p["x"], x)
    tool("grouping_train", d, nc, M, nsubc, p["cent"], p["info"], p["edges"], p["x"], n, p["pq"], p["npq"])
    raw = open(p["pq"], "rb").read()
    hd = struct.unpack("<QQQQ", raw[:32])
    assert hd == (d, M, 8, d * 256)
    cb = np.frombuffer(raw[32:], np.float32).reshape(M, 256, d // M)
    assert np.isfinite(cb).all() and np.abs(cb).max() < 100        # residual scale, not data scale
    rawn = open(p["npq"], "rb").read()
    assert struct.unpack("<QQQQ", rawn[:32]) == (1, 1, 8, 256)
    nt = np.frombuffer(rawn[32:], np.float32)
    norms = (x.astype(np.float64) ** 2).sum(1)
    assert np.isfinite(nt).all() and nt.min() > 0.3 * norms.min() and nt.max() < 3 * norms.max()
