"""Error behaviour of the C ABI on a live device: wrong call order and bad arguments are refused with a status
code and a message (the reference itself exit()s or corrupts memory in these cases; SURVEY.md 8b "error
conventions"), and a refused call leaves the handle usable."""
import numpy as np
import pytest

from conftest import corpus

pytestmark = pytest.mark.gpu


def _upload(g, c):
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])


def test_call_order_and_argument_checks(gpu, pkg):
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    g = gpu()
    q = c["queries"][:4]
    cid = np.zeros((4, 8), np.uint32)
    cd = np.zeros((4, 8), np.float32)

    with pytest.raises(pkg.IvfHnswError) as e:            # nothing uploaded yet
        g.search(q, 1, 8, 1000, coarse_ids=cid, coarse_dists=cd)
    assert e.value.code == pkg.ERR_STATE

    _upload(g, c)
    with pytest.raises(pkg.IvfHnswError) as e:            # device walk requested without a graph
        g.search(q, 1, 8, 1000, efSearch=40)
    assert e.value.code == pkg.ERR_STATE and "upload_quantizer" in str(e.value)

    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    with pytest.raises(pkg.IvfHnswError) as e:            # the reference's precondition efSearch >= nprobe
        g.search(q, 1, 8, 1000, efSearch=4)
    assert e.value.code == pkg.ERR_INVALID and "efSearch" in str(e.value)

    with pytest.raises(pkg.IvfHnswError) as e:            # only one of the two coarse arrays
        g.search(q, 1, 8, 1000, coarse_ids=cid)
    assert e.value.code == pkg.ERR_INVALID

    with pytest.raises(pkg.IvfHnswError) as e:
        g.search(q, 2000, 8, 1000, efSearch=40)           # k beyond the supported 1024
    assert e.value.code == pkg.ERR_INVALID

    # the handle still works after every refusal
    dist, lab = g.search(q, 1, 8, 1000, efSearch=40)
    assert (lab >= 0).all()


def test_upload_validation(gpu, pkg):
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    g = gpu()
    bad = c["offsets"].copy()
    bad[5], bad[6] = bad[6], bad[5] - 1 if bad[5] else 0   # not monotone
    if not (np.diff(bad.astype(np.int64)) < 0).any():
        bad[7] = bad[6] - 1
    with pytest.raises(pkg.IvfHnswError) as e:
        g.upload_ivf(c["d"], c["code_size"], bad, c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                     c["pq_centroids"], c["norm_table"])
    assert e.value.code == pkg.ERR_INVALID and "monotone" in str(e.value)

    with pytest.raises(pkg.IvfHnswError) as e:            # code_size must be a multiple of 4 (IndexIVF_HNSW.cpp:805)
        g.upload_ivf(126, 6, c["offsets"], c["ids"], np.zeros((30000, 6), np.uint8), c["norm_codes"],
                     c["centroid_norms"], np.zeros(256 * 126, np.float32), c["norm_table"])
    assert e.value.code == pkg.ERR_INVALID

    _upload(g, c)
    links = c["graph"].links.copy()
    links[3, 0] = 10 ** 6                                  # link outside the graph
    with pytest.raises(pkg.IvfHnswError) as e:
        g.upload_quantizer(c["graph"].counts, links, c["graph"].vectors, 0)
    assert e.value.code == pkg.ERR_INVALID and "out of range" in str(e.value)

    # grouping tables must partition every list
    nsubc = 4
    sg = np.zeros((256, nsubc), np.uint32)
    with pytest.raises(pkg.IvfHnswError) as e:
        g.upload_grouping(nsubc, np.zeros(256, np.float32), np.zeros((256, nsubc), np.uint32), sg,
                          np.zeros((256, nsubc), np.float32))
    assert e.value.code == pkg.ERR_INVALID and "sum to" in str(e.value)


def test_large_batch_is_chunked(gpu):
    """More than 2^17 queries in one call are processed in slices with a bounded workspace."""
    c = corpus(seed=15, nc=64, d=64, M=4, n_base=4000, nq=32, efConstruction=60)
    g = gpu()
    _upload(g, c)
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    n = (1 << 17) + 777
    q = np.tile(c["queries"], (n // 32 + 1, 1))[:n]
    dist, lab = g.search(q, 1, 8, 500, efSearch=16)
    d0, l0 = g.search(c["queries"], 1, 8, 500, efSearch=16)
    assert np.array_equal(lab.reshape(-1)[:32 * (n // 32)].reshape(-1, 32), np.tile(l0.reshape(1, 32), (n // 32, 1)))
    assert np.array_equal(dist[-777 % 32 or None:][:0], dist[:0])  # shape sanity
    assert lab[-1, 0] == l0[(n - 1) % 32, 0] and dist[-1, 0] == d0[(n - 1) % 32, 0]


def test_options_and_neighbour_table_arguments_are_checked(gpu, pkg):
    g = gpu()
    g.set_option("scan_pipe", 0)
    g.set_option("scan_pipe", -1)
    with pytest.raises(pkg.IvfHnswError) as e:
        g.set_option("scan_pipe", 2)
    assert e.value.code == pkg.ERR_INVALID
    with pytest.raises(pkg.IvfHnswError) as e:
        g.set_option("no_such_option", 1)
    assert e.value.code == pkg.ERR_INVALID and "unknown key" in str(e.value)
    x = np.zeros((40, 128), np.float32)
    with pytest.raises(pkg.IvfHnswError):
        g.knn(x, 4, mode=3)
    with pytest.raises(pkg.IvfHnswError):
        g.build_graph(x, M=16, maxM=8)          # M > maxM
    with pytest.raises(pkg.IvfHnswError):
        g.build_graph(x, M=16, maxM=32, ncand=100)
    with pytest.raises(pkg.IvfHnswError) as e:
        g.build_graph(np.zeros((40, 24), np.float32), M=8, maxM=16, ncand=16)   # the reference's distance stops at d - d % 16
    assert e.value.code == pkg.ERR_INVALID and "multiple of 16" in str(e.value)
    # the handle is still usable
    ids, _ = g.knn(x + np.arange(40, dtype=np.float32)[:, None], 3)
    assert ids.shape == (40, 3)
