#!/usr/bin/env python3
"""Where one query per call spends its time (the reference drivers' loop, tests/test_ivfhnsw_sift1b.cpp:193-208)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import synth
pkg = ge.load_pkg()
big = len(sys.argv) > 1 and sys.argv[1] == "1B"
nc = 993127 if big else 1 << 17
tb = synth.make_throughput_tables(7, nc, 128, 16, 1_000_000_000 if big else 100_000_000)
counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32)
cn = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
g = pkg.GpuIndex(0)
g.upload_ivf_synthetic(128, 16, tb["offsets"], cn, tb["pq_centroids"], tb["norm_table"], 11)
g.upload_quantizer(counts, links, tb["centroids"], 0)
rng = np.random.default_rng(1)
q = (tb["centroids"][rng.choice(nc, 1400)] + rng.normal(0, 12, (1400, 128))).astype(np.float32)
def probe(label):
  for nq in (1, 4, 16, 64, 256, 1024):
      for i in range(0, 40 * nq, nq):
          g.search(q[i % 300:i % 300 + nq], 1, 32, 10000, efSearch=80)
      g.set_profiling(True); g.reset_stage_ms()
      n = 100
      t0 = time.perf_counter()
      for i in range(n):
          g.search(q[(i * nq) % 100:(i * nq) % 100 + nq], 1, 32, 10000, efSearch=80)
      el = (time.perf_counter() - t0) / n
      st = g.stage_ms(); g.set_profiling(False)
      print(label + " nq %4d: %.0f us per call; stages (us per call): %s" % (
          nq, el * 1e6, {k: round(v[0] / max(1, v[1]) * 1e3, 1) for k, v in st.items()}))

# the serial CPU port on the same corpus and queries, one query per call as the reference's drivers loop
# (tests/test_ivfhnsw_sift1b.cpp:193-208); only the lists these queries probe are materialised on the host
from oracle import orc
graph = orc.Hnsw.from_arrays(counts, links, tb["centroids"], 16, 0)
arrays = synth.synthetic_codes_sparse(11, tb["offsets"], 16)
ox = orc.Index(128, 16, graph, tb["pq_centroids"], tb["norm_table"], tb["offsets"], arrays[0], arrays[1], arrays[2], cn)
ox.set_params(32, 10000, 80)
nqc = 400
_, _, cid, _, _ = ox.search_batch(q[:nqc], 1, 1)
synth.synthetic_codes_sparse(11, tb["offsets"], 16, cid.ravel()[cid.ravel() < nc], into=arrays)
ox.search_batch(q[:nqc], 1, 1)
t0 = time.perf_counter()
ref_d, ref_l, _, _, _ = ox.search_batch(q[:nqc], 1, 1)
cpu_us = (time.perf_counter() - t0) / nqc * 1e6
print("[cpu port, serial] %.0f us per query (%d queries, one thread)" % (cpu_us, nqc))

def check(label):
    ok = True
    t0 = time.perf_counter()
    for i in range(nqc):
        d1, l1 = g.search(q[i], 1, 32, 10000, efSearch=80)
        ok &= l1[0, 0] == ref_l[i, 0] and d1.view(np.uint32)[0, 0] == ref_d.view(np.uint32)[i, 0]
    us = (time.perf_counter() - t0) / nqc * 1e6
    print("%s one query per call, stage events off: %.0f us per call (cpu port %.0f); results equal to the port: %s"
          % (label, us, cpu_us, ok))

probe("[throughput walk]")
check("[throughput walk]")
g.prepare_latency()
probe("[latency walk]   ")
check("[latency walk]   ")
