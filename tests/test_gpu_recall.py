"""A recall-bearing index from the library's own pipeline on clustered data: graph construction,
code-book training, assignment + encoding and the exact ground truth all run on the device (synth.make_recall_corpus); the
search path's labels must equal the CPU port's on it, so the two have the SAME Recall@1 -- the figure BASELINE.json's metric
is quoted at (the reference's drivers print it, tests/test_ivfhnsw_sift1b.cpp:173-215)."""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def test_recall_of_the_device_path_equals_the_cpu_ports(gpu, pkg):
    c = synth.make_recall_corpus(pkg, seed=2024, nc=4096, n_base=1_000_000, nq=4000, log=print)
    assert c["assign_agrees_with_generator"] > 0.5
    nprobe, max_codes, ef = 32, 10000, 80
    g = gpu()
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"])
    g.upload_quantizer(c["counts"], c["links"], c["centroids"], 0)
    dist, lab = g.search(c["queries"], 1, nprobe, max_codes, efSearch=ef)
    graph = orc.Hnsw.from_arrays(c["counts"], c["links"], c["centroids"], 16, 0)
    ox = orc.Index(c["d"], c["code_size"], graph, c["pq_centroids"], c["norm_table"], c["offsets"], c["ids"], c["codes"],
                   c["norm_codes"], c["centroid_norms"])
    ox.set_params(nprobe, max_codes, ef)
    rd, rl, _, _, st = ox.search_batch(c["queries"], 1, 16)
    assert np.array_equal(lab, rl) and np.array_equal(dist.view(np.uint32), rd.view(np.uint32))
    assert g.last_scan_counts()[0] == st.ncode
    r_gpu = float((lab[:, 0] == c["gt"]).mean())
    r_cpu = float((rl[:, 0] == c["gt"]).mean())
    print("\n[recall] 1M clustered vectors, 4096 centroids, PQ16, (32, 10000, 80): Recall@1 %.4f on the device = %.4f on the "
          "CPU port; the generating centroid is the assigned one for %.3f of the vectors" % (r_gpu, r_cpu, c["assign_agrees_with_generator"]))
    assert r_gpu == r_cpu
    assert 0.15 < r_gpu < 0.999       # a figure that can move: neither saturated nor noise
    # more probes and codes can only help a query whose true neighbour was out of reach
    d2, l2 = g.search(c["queries"], 1, 64, 10 ** 9, efSearch=120)
    assert float((l2[:, 0] == c["gt"]).mean()) >= r_gpu - 0.002
    graph.free()
