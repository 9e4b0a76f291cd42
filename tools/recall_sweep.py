"""Recall@1 of the device path on the clustered real-pipeline corpus as a function of the query noise (picks the noise the
recall-bearing bench entry uses).  usage: python tools/recall_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import synth

pkg = ge.load_pkg()
for qn in (8.0, 12.0, 16.0, 20.0):
    c = synth.make_recall_corpus(pkg, seed=2024, nc=4096, n_base=1_000_000, nq=4000, query_noise=qn)
    g = pkg.GpuIndex(0)
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    g.upload_quantizer(c["counts"], c["links"], c["centroids"], 0)
    out = []
    for nprobe, mc, ef in ((32, 10000, 80), (64, 30000, 100), (256, 10 ** 9, 300)):
        _, lab = g.search(c["queries"], 1, nprobe, mc, efSearch=ef)
        out.append("%.4f" % float((lab[:, 0] == c["gt"]).mean()))
    print("query noise %.0f: Recall@1 %s at (32,10000,80) / (64,30000,100) / (256,inf,300); gt is the source row for %.3f"
          % (qn, " / ".join(out), float((c["gt"] == c["query_src"]).mean())), flush=True)
    g.close()
