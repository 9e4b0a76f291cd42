"""Seeded differential sweep: random small configurations of the whole search path (dimension, code size, list
count, empty lists, OPQ, Grouping with and without pruning, nprobe, max_codes, efSearch, k, batch size) against
the oracle -- labels and distances bit for bit, the reference's heap-array order for k > 1.  Catches the corner
a hand-written case list does not think of; every failure prints its configuration.
"""
import numpy as np
import pytest

import synth
from oracle import orc  # noqa: F401  (the oracle library must be built)

pytestmark = pytest.mark.gpu


def _configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d = int(rng.choice([16, 32, 48, 64, 96, 128]))
        M = int(rng.choice([m for m in (4, 8, 16, 32) if d % m == 0 and d // m <= 16]))
        nc = int(rng.choice([24, 64, 200, 500]))
        grouping = bool(rng.random() < 0.4)
        nsubc = int(rng.choice([4, 8, 16])) if grouping else 0
        if grouping:
            nc = max(nc, 2 * nsubc + 8)
        nprobe = int(rng.integers(1, min(nc // 2, 40) + 1))
        ef = int(nprobe + rng.integers(0, 60))
        kw = dict(seed=int(1000 + i), nc=nc, d=d, M=M, n_base=int(rng.integers(nc, 40 * nc)), nq=int(rng.integers(1, 50)),
                  efConstruction=40, empty_frac=float(rng.choice([0.0, 0.1, 0.5])), opq=bool(rng.random() < 0.4),
                  nsubc=nsubc)
        max_codes = int(rng.choice([0, 1, 50, 500, 5000, 10 ** 9]))
        k = int(rng.choice([1, 1, 1, 2, 5, 17, 64]))
        pruning = grouping and bool(rng.random() < 0.6)
        out.append((kw, nprobe, max_codes, ef, k, pruning))
    return out


@pytest.mark.parametrize("case", _configs(100, 20261004), ids=lambda c: "d%d-M%d-nc%d-g%d-np%d-mc%d-k%d" % (
    c[0]["d"], c[0]["M"], c[0]["nc"], c[0]["nsubc"], c[1], min(c[2], 99999), c[4]))
def test_random_configuration_matches_oracle(gpu, case):
    kw, nprobe, max_codes, ef, k, pruning = case
    c = synth.make_corpus(**kw)
    ox = synth.oracle_index(c)
    ox.set_params(nprobe, max_codes, ef, do_pruning=pruning)
    ref_d, ref_l, _, _, st = ox.search_batch(c["queries"], k=k)
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    if c["nsubc"]:
        g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    dist, lab = g.search(c["queries"], k, nprobe, max_codes, efSearch=ef, do_pruning=pruning, heap_order=True)
    assert np.array_equal(lab, ref_l), (case, np.nonzero((lab != ref_l).any(1))[0][:5])
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32)), case
    assert g.last_scan_counts()[0] == st.ncode, case
