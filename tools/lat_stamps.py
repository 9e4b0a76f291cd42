import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge, synth
pkg = ge.load_pkg()
nc = 1 << 17
tb = synth.make_throughput_tables(7, nc, 128, 16, 100_000_000)
counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32)
g = pkg.GpuIndex(0)
g.upload_quantizer(counts, links, tb["centroids"], 0)
g.prepare_latency()
rng = np.random.default_rng(1)
q = (tb["centroids"][rng.choice(nc, 8)] + rng.normal(0, 12, (8, 128))).astype(np.float32)
for i in range(6):
    g.coarse(q[i], 32, 80)
