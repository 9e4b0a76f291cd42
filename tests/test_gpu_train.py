"""Code-book training on the device against the oracle's restatement: one and several Lloyd
iterations of ProductQuantizer::train (assignments and code words bit for bit), the encode that follows with the
trained code book (IndexIVF_HNSW.cpp:536-593 train_pq, then :75-121 add_batch), and OPQ's X^T Y on the matrix cores.
faiss's own clustering is absent from the reference tree: what is pinned here is this repo's contract (oracle/
ivfhnsw_oracle.c orc_pq_lloyd, orc_xty), parity with faiss itself is unpinned."""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("d,M,n,niter", [(64, 8, 20000, 1), (64, 8, 20000, 4), (96, 16, 9000, 2), (128, 8, 6000, 2),
                                          (1, 1, 5000, 3)])
def test_lloyd_iterations_match_oracle(gpu, d, M, n, niter):
    rng = np.random.default_rng(100 + d + niter)
    x = rng.normal(0, 9.0, size=(n, d)).astype(np.float32) if d > 1 else \
        rng.normal(2.6e5, 3e4, size=(n, 1)).astype(np.float32)          # d = 1: the norm code book
    dsub = d // M
    c0 = np.stack([x[rng.choice(n, 256, replace=False), m * dsub:(m + 1) * dsub] for m in range(M)])
    ref_c, ref_a = orc.pq_lloyd(x, M, c0, niter)
    g = gpu()
    dev_c, dev_a = g.pq_train(x, M, c0, niter)
    assert np.array_equal(dev_a, ref_a)
    assert _same(dev_c.reshape(-1), ref_c.reshape(-1))
    assert not _same(dev_c.reshape(-1), np.ascontiguousarray(c0, np.float32).reshape(-1))


def test_empty_clusters_keep_their_code_word(gpu):
    """A code word nothing is assigned to stays as it is (the host loop's `if (cnt[c])`): seed half of the code book
    far away from every point."""
    rng = np.random.default_rng(7)
    n, d, M = 4000, 32, 4
    x = rng.normal(0, 1.0, size=(n, d)).astype(np.float32)
    c0 = np.stack([x[rng.choice(n, 256, replace=False), m * 8:(m + 1) * 8] for m in range(M)])
    c0[:, 128:] += 1000.0
    ref_c, ref_a = orc.pq_lloyd(x, M, c0, 2)
    dev_c, dev_a = gpu().pq_train(x, M, c0, 2)
    assert np.array_equal(dev_a, ref_a) and _same(dev_c.reshape(-1), ref_c.reshape(-1))
    assert _same(dev_c[:, 128:].reshape(-1), c0[:, 128:].astype(np.float32).reshape(-1))
    assert (dev_a < 128).all()


def test_encode_with_the_trained_code_book_matches_oracle(gpu):
    """train_pq then add_batch: residuals -> Lloyd on the device -> the encode chain with that code book."""
    e = synth.make_encode_case(seed=77, nc=96, d=64, M=8, opq=False, n=6000)
    g = gpu()
    gr = e["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    idx, _ = g.coarse(e["x"], 1, 40)
    res = (e["x"] - e["cents"][idx[:, 0]]).astype(np.float32)
    c0 = np.stack([res[:256, m * 8:(m + 1) * 8] for m in range(8)])
    cb, _ = g.pq_train(res, 8, c0, 3)
    ref_cb, _ = orc.pq_lloyd(res, 8, c0, 3)
    assert _same(cb.reshape(-1), ref_cb.reshape(-1))
    g.upload_codebooks(64, 8, cb, e["nt"])
    d_idx, d_codes, d_nc = g.encode(e["x"], efSearch=40)
    ox = orc.Index(64, 8, gr, ref_cb, e["nt"], np.zeros(97, np.uint64), np.zeros(0, np.uint32),
                   np.zeros((0, 8), np.uint8), np.zeros(0, np.uint8), np.zeros(96, np.float32))
    ox.set_params(1, 0, 40)
    o_idx, o_codes, o_nc, _ = ox.add_batch_encode(e["x"])
    assert np.array_equal(d_idx, o_idx) and np.array_equal(d_codes, o_codes) and np.array_equal(d_nc, o_nc)


@pytest.mark.parametrize("n,d", [(5000, 128), (2048, 96), (4100, 40), (300, 32)])
def test_xty_matches_the_mfma_order(gpu, pkg, n, d):
    """OPQ's X^T Y: fmaf chains over chunks of 2048 points on v_mfma_f32_32x32x2_f32 (d % 32 == 0) or the scalar form."""
    rng = np.random.default_rng(n + d)
    X = rng.normal(0, 5.0, size=(n, d)).astype(np.float32)
    Y = (X @ synth.random_rotation(rng, d) + rng.normal(0, 0.5, size=(n, d))).astype(np.float32)
    C = gpu().xty(X, Y)
    ref = orc.xty(X, Y, 2048)
    assert _same(C.reshape(-1), ref.reshape(-1))
    exact = X.astype(np.float64).T @ Y.astype(np.float64)
    assert np.abs(C - exact).max() <= 1e-5 * np.abs(exact).max()
