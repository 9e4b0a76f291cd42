"""Not a parity test: records what a drop-in caller sees when it keeps the reference's one-query-per-call loop
(tests/test_ivfhnsw_sift1b.cpp:193-208) instead of the batched entry point."""
import time

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def test_single_query_latency_is_reported(gpu, capsys):
    tb = synth.make_throughput_tables(7, 1 << 14, 128, 16, 10_000_000)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32)
    cn = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
    g = gpu()
    g.upload_ivf_synthetic(128, 16, tb["offsets"], cn, tb["pq_centroids"], tb["norm_table"], 11)
    g.upload_quantizer(counts, links, tb["centroids"], 0)
    rng = np.random.default_rng(1)
    q = (tb["centroids"][rng.choice(1 << 14, 300)] + rng.normal(0, 12, (300, 128))).astype(np.float32)
    for x in q[:20]:
        g.search(x, 1, 32, 10000, efSearch=80)
    t0 = time.perf_counter()
    for x in q:
        g.search(x, 1, 32, 10000, efSearch=80)
    per_call = (time.perf_counter() - t0) / len(q)
    t0 = time.perf_counter()
    g.search(q, 1, 32, 10000, efSearch=80)
    batch = time.perf_counter() - t0
    # the latency form of the walk (what the classes' search() prepares): same answers, fewer microseconds
    d_ref, l_ref = g.search(q, 1, 32, 10000, efSearch=80)
    g.prepare_latency()
    for x in q[:20]:
        g.search(x, 1, 32, 10000, efSearch=80)
    t0 = time.perf_counter()
    for i, x in enumerate(q):
        d1, l1 = g.search(x, 1, 32, 10000, efSearch=80)
        assert l1[0, 0] == l_ref[i, 0] and d1[0, 0] == d_ref[i, 0]
    per_call_lat = (time.perf_counter() - t0) / len(q)
    with capsys.disabled():
        print("\n[latency] one query per call: %.0f us/query (throughput walk), %.0f us/query (latency walk, results "
              "checked); one call for %d queries: %.0f us/query"
              % (per_call * 1e6, per_call_lat * 1e6, len(q), batch / len(q) * 1e6))
    assert per_call < 0.05 and per_call_lat < per_call
