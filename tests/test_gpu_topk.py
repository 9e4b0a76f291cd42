"""GPU parity for k > 1 (IndexIVF_HNSW.cpp:265,285-288).  heap_order = 0: the same SET of (distance, label) pairs as
the reference's max-heap ends with, ascending.  heap_order = 1: the very ARRAY faiss's heap leaves, element for
element (device replay of pop/push over a superset of the admitted codes in scan order).
"""
import numpy as np
import pytest

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu


def _pairs_sorted(dist, lab):
    o = np.lexsort((lab, dist))
    return dist[o], lab[o]


@pytest.mark.parametrize("k", [2, 10, 100, 1000])
@pytest.mark.parametrize("grouping", [False, True])
def test_topk_set_matches_oracle(gpu, k, grouping):
    if grouping:
        c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    else:
        c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    nprobe, max_codes, ef = 16, 3000, 40
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    q = c["queries"][:48]
    ref_d, ref_l, cid, cd, _ = ox.search_batch(q, k=k)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    if grouping:
        g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                          c["inter_centroid_dists"])
        gr = c["graph"]
        g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    dist, lab = g.search(q, k, nprobe, max_codes, coarse_ids=cid, coarse_dists=cd)
    for i in range(len(q)):
        assert (np.diff(dist[i]) >= 0).all()  # ascending
        rd, rl = _pairs_sorted(ref_d[i], ref_l[i])
        gd, gl = _pairs_sorted(dist[i], lab[i])
        assert np.array_equal(gd.view(np.uint32), rd.view(np.uint32)), "query %d distances" % i
        assert np.array_equal(gl, rl), "query %d labels" % i


def test_topk_fewer_codes_than_k(gpu):
    """Unfilled slots keep the heapify state FLT_MAX / -1 (IndexIVF_HNSW.cpp:265)."""
    c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    sizes = np.diff(c["offsets"].astype(np.int64))
    small = int(np.argmin(np.where(sizes > 0, sizes, 10 ** 9)))
    n = int(sizes[small])
    k = n + 5
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    cid = np.array([[small]], np.uint32)
    cd = np.array([[123.0]], np.float32)
    dist, lab = g.search(c["queries"][:1], k, 1, 10 ** 9, coarse_ids=cid, coarse_dists=cd)
    assert (lab[0, n:] == -1).all() and (dist[0, n:] == np.finfo(np.float32).max).all()
    off = int(c["offsets"][small])
    assert sorted(lab[0, :n].tolist()) == sorted(c["ids"][off:off + n].tolist())


@pytest.mark.parametrize("k", [2, 3, 10, 100, 1000])
@pytest.mark.parametrize("grouping", [False, True])
def test_heap_order_is_the_reference_array(gpu, k, grouping):
    if grouping:
        c = corpus(seed=41, nc=256, d=128, M=16, n_base=30000, nq=96, nsubc=16)
    else:
        c = corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128)
    nprobe, max_codes, ef = 16, 3000, 40
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
    q = c["queries"][:64]
    ref_d, ref_l, cid, cd, _ = ox.search_batch(q, k=k)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    if grouping:
        g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"],
                          c["inter_centroid_dists"])
        gr = c["graph"]
        g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    dist, lab = g.search(q, k, nprobe, max_codes, coarse_ids=cid, coarse_dists=cd, do_pruning=grouping,
                         heap_order=True)
    assert np.array_equal(lab, ref_l)                                    # same slot, same label
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))


def test_heap_order_with_ties(gpu):
    """All codes equal inside a list: many exact distance ties; which of them the heap keeps, and where, depends on
    the pop/push sequence -- it must still be the oracle's array."""
    c = dict(corpus(seed=11, nc=256, d=128, M=16, n_base=30000, nq=128))
    c["codes"] = np.zeros_like(c["codes"])
    c["norm_codes"] = np.zeros_like(c["norm_codes"])
    ox = synth.oracle_index(c)
    ox.set_params(8, 10 ** 9, 40)
    q = c["queries"][:32]
    for k in (4, 50):
        ref_d, ref_l, cid, cd, _ = ox.search_batch(q, k=k)
        g = gpu()
        g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                     c["pq_centroids"], c["norm_table"])
        dist, lab = g.search(q, k, 8, 10 ** 9, coarse_ids=cid, coarse_dists=cd, heap_order=True)
        assert np.array_equal(lab, ref_l)
        assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
