"""Times ivfhnsw_gpu_build_graph at the reference's 993 127 centroids (iid and clustered tables) and reports what the walk
finds on the result.  usage: python tools/graph_bench.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import synth

pkg = ge.load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 993127
for kind in ("iid", "clustered"):
    rng = np.random.default_rng(1)
    if kind == "iid":
        x = synth.sift_like(rng, n, 128)
    else:
        x = synth.clustered_centroids(rng, n, 128)
    g = pkg.GpuIndex(0)
    t0 = time.time()
    counts, links = g.build_graph(x, 16, 32, 64)
    t1 = time.time()
    q = (x[rng.integers(0, n, 10000)] + rng.normal(0, 8.0, (10000, 128))).astype(np.float32)
    gt, _ = g.knn(x, 1, q)
    g.upload_quantizer(counts, links, x, 0)
    ids, _ = g.coarse(q, 1, 80)
    t2 = time.time()
    kc, kl = synth.knn_graph(x, 16, 32)
    t3 = time.time()
    g2 = pkg.GpuIndex(0)
    g2.upload_quantizer(kc, kl, x, 0)
    ids2, _ = g2.coarse(q, 1, 80)
    print("%s n=%d: build_graph %.1f s (mean degree %.1f, max %d); walk ef 80 finds the true nearest for %.4f; "
          "plain k-NN graph: %.1f s, degree %.1f, walk %.4f" % (kind, n, t1 - t0, counts.mean(), counts.max(),
          (ids[:, 0] == gt[:, 0]).mean(), t3 - t2, kc.mean(), (ids2[:, 0] == gt[:, 0]).mean()), flush=True)
    g.close(); g2.close()
