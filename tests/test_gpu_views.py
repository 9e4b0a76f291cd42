"""Views (ivfhnsw_gpu_create_view): a second search context on the same device tables.

Bar: a view returns exactly what its parent returns (labels equal, distances bit-identical, both equal to the
oracle), also when parent and view have batches in flight at the same time on two streams; uploads on a view
are refused.
"""
import numpy as np
import pytest
import torch

from conftest import corpus
import synth

pytestmark = pytest.mark.gpu


def _upload(g, c):
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)


@pytest.mark.parametrize("kw,nprobe,max_codes,ef", [
    (dict(seed=31, nc=512, d=128, M=16, n_base=40000, nq=512), 16, 3000, 40),
    (dict(seed=32, nc=256, d=96, M=16, n_base=20000, nq=256, opq=True), 32, 5000, 64),
])
def test_view_matches_parent_and_oracle(gpu, pkg, kw, nprobe, max_codes, ef):
    c = corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    v = g.view()
    try:
        d0, l0 = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef)
        d1, l1 = v.search(c["queries"], 1, nprobe, max_codes, efSearch=ef)
        for dd, ll in ((d0, l0), (d1, l1)):
            assert np.array_equal(ll, ref_l)
            assert np.array_equal(dd.view(np.uint32), ref_d.view(np.uint32))
        with pytest.raises(pkg.IvfHnswError):
            _upload(v, c)
        with pytest.raises(pkg.IvfHnswError):
            v.view()
    finally:
        v.close()


def test_two_batches_in_flight(gpu):
    """Parent and view on two streams, different batches submitted back to back, several rounds."""
    c = corpus(seed=33, nc=512, d=128, M=16, n_base=40000, nq=1024)
    nprobe, max_codes, ef = 16, 3000, 40
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef)
    ref_d, ref_l, _, _, _ = ox.search_batch(c["queries"], k=1)
    g = gpu()
    _upload(g, c)
    v = g.view()
    try:
        dev = torch.device("cuda", 0)
        ctx = []
        half = c["queries"].shape[0] // 2
        for h, sl in ((g, slice(0, half)), (v, slice(half, 2 * half))):
            st = torch.cuda.Stream(device=dev)
            h.set_stream(st.cuda_stream)
            q = torch.from_numpy(np.ascontiguousarray(c["queries"][sl])).to(dev)
            ctx.append((h, st, sl, q, torch.empty((half, 1), dtype=torch.float32, device=dev),
                        torch.empty((half, 1), dtype=torch.int64, device=dev)))
        torch.cuda.synchronize()
        for _ in range(5):
            for h, st, sl, q, dd, ll in ctx:
                dd.fill_(0)
                ll.fill_(0)
            torch.cuda.synchronize()
            for h, st, sl, q, dd, ll in ctx:
                h.search_dev(half, 1, q, dd, ll, nprobe, max_codes, efSearch=ef)
            torch.cuda.synchronize()
            for h, st, sl, q, dd, ll in ctx:
                assert np.array_equal(ll.cpu().numpy(), ref_l[sl])
                assert np.array_equal(dd.cpu().numpy().view(np.uint32), ref_d[sl].view(np.uint32))
    finally:
        v.close()
