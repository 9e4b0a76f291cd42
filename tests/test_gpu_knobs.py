"""Every knob the device path still has (round 3 removed the measured-slower kernel forms and theirs) selects a different kernel form of the SAME computation.  The knobs are read
once per process, so each setting runs in its own child process on one seeded corpus; all of them must print the
digest of the default path (coarse ids and distances, final labels and distances, grouping search included).
"""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import __graft_entry__ as ge
import synth
pkg = ge.load_pkg()
h = hashlib.sha256()
for kw in (dict(seed=91, nc=512, d=128, M=16, n_base=20000, nq=200, efConstruction=60),
           dict(seed=92, nc=256, d=96, M=8, n_base=12000, nq=100, efConstruction=60, nsubc=8, opq=True)):
    c = synth.make_corpus(**kw)
    g = pkg.GpuIndex(0)
    g.upload_ivf(c["d"], c["code_size"], c["offsets"], c["ids"], c["codes"], c["norm_codes"], c["centroid_norms"],
                 c["pq_centroids"], c["norm_table"], opq_A=c["opq_A"])
    gr = c["graph"]
    g.upload_quantizer(gr.counts, gr.links, gr.vectors, gr.enterpoint)
    if c["nsubc"]:
        g.upload_grouping(c["nsubc"], c["alphas"], c["nn_centroid_idxs"], c["subgroup_sizes"], c["inter_centroid_dists"])
    for ef, k in ((16, 16), (80, 32), (140, 32)):
        ids, dist = g.coarse(c["queries"], k, ef)
        h.update(ids.tobytes()); h.update(dist.tobytes())
    dist, lab = g.search(c["queries"], 1, 16, 2000, efSearch=60, do_pruning=bool(c["nsubc"]))
    h.update(dist.tobytes()); h.update(lab.tobytes())
    dist, lab = g.search(c["queries"], 5, 16, 2000, efSearch=60, heap_order=True)
    h.update(dist.tobytes()); h.update(lab.tobytes())
print("DIGEST", h.hexdigest())
'''


def _run(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
    assert line, r.stdout[-500:]
    return line[0]


def test_every_knob_setting_reproduces_the_default_path():
    base = _run({})
    for env in ({"IVFHNSW_WALK_PREFILTER": "0"},     # no rejection filter
                {"IVFHNSW_WALK_PREFILTER": "1"},     # gather form of the filter
                {"IVFHNSW_WALK_MERGE": "0"},         # admissions one by one
                {"IVFHNSW_WALK_VIS": "bitmap"},      # global visited bitmaps
                {"IVFHNSW_WALK_TAGW": "10"}, {"IVFHNSW_WALK_TAGW": "12"}, {"IVFHNSW_WALK_TAGW": "16"},  # visited-set tag widths
                # survivors of the filter entered into the visited set late (default only beyond 257 k nodes)
                {"IVFHNSW_WALK_LATE_VISIT": "1"}, {"IVFHNSW_WALK_LATE_VISIT": "1", "IVFHNSW_WALK_TAGW": "10"},
                {"IVFHNSW_WALK_LATE_VISIT": "1", "IVFHNSW_WALK_TAGW": "16"}, {"IVFHNSW_WALK_LATE_VISIT": "0"},
                {"IVFHNSW_TAIL": "0"},               # small batches through the four separate launches
                {"IVFHNSW_PLAN_GROUP4": "0"},        # Grouping plan by one wavefront per query (default: four)
                {"IVFHNSW_PLAN_DEDUPE": "1"},        # ... every distinct neighbour centroid once, through the LDS hash set
                {"IVFHNSW_PLAN_DEDUPE": "0"},        # ... every (row, sub-group) pair's row gathered
                {"IVFHNSW_SPLIT": "0"},              # large batches in one part (default: two parts on two streams)
                {"IVFHNSW_PLAN_LUT": "0"},           # plan and tables as two launches (default: one, plan_lut_kernel)
                {"IVFHNSW_SCAN_PIPE": "1"}):         # table + scan pipelined over queries (kernels_scan3.hip), one shard too
        assert _run(env) == base, env
