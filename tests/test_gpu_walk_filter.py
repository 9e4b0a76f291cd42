"""GPU parity of the walk's exact rejection filter (kernels_hnsw.hip, DESIGN.md "coarse walk").

The filter drops a neighbour without reading its float row when a lower bound of its distance, computed from
a byte copy of the row, already exceeds the result set's current maximum.  It must never change a result, so
every case here is the plain oracle comparison (same ids, bit-identical distances) on inputs chosen to push
the bound: queries far outside the range the bytes cover (clamped components on either side), signed data,
data the bytes represent exactly (zero quantisation slack, exact ties at the boundary), rows shorter than the
128-byte record (zero padded), degrees beyond one 32-row batch, and dimensions that select the gather form
(d > 128) or no filter at all (d > 2048, constant tables).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def _walk_both(g, graph, queries, k, ef):
    g.upload_quantizer(graph.counts, graph.links, graph.vectors, graph.enterpoint)
    ids, dist = g.coarse(queries, k, ef)
    for i, q in enumerate(queries):
        rid, rd = graph.search_knn(q, ef, k)
        n = len(rid)
        assert np.array_equal(ids[i, :n], rid), "query %d: ids differ\n%s\n%s" % (i, ids[i], rid)
        assert np.array_equal(dist[i, :n].view(np.uint32), rd.view(np.uint32)), "query %d: distances differ" % i
        assert (ids[i, n:] == 0xffffffff).all()


def _stress_queries(rng, cents, n):
    """Queries of every kind the bound has a separate term for."""
    nc, d = cents.shape
    near = cents[rng.choice(nc, n)] + rng.normal(0, 10.0, size=(n, d))
    q = [near,
         near * 3.0,                                  # most components above the byte range
         near - 4.0 * np.abs(near).mean(),            # most components below it
         near + rng.normal(0, 200.0, size=(n, d)),    # both sides, large excess
         cents[rng.choice(nc, n)],                    # distance 0 to a node
         np.full((4, d), 1.0e4), np.full((4, d), -1.0e4), np.zeros((4, d))]
    spike = near.copy()
    spike[:, ::7] = 1.0e5                             # excess beyond what the XH/XL planes can hold (saturates)
    q.append(spike)
    return np.ascontiguousarray(np.concatenate(q), np.float32)


@pytest.mark.parametrize("ef,k", [(16, 16), (80, 32), (200, 64)])
def test_queries_outside_the_byte_range(gpu, ef, k):
    rng = np.random.default_rng(101)
    cents = synth.sift_like(rng, 4096, 128)
    graph = orc.Hnsw.build(cents, M=16, efConstruction=60)
    _walk_both(gpu(), graph, _stress_queries(rng, cents, 24), k, ef)


def test_signed_unit_vectors_d96(gpu):
    """DEEP1B-like rows: unit vectors with components of both signs, 96 of the record's 128 bytes used."""
    rng = np.random.default_rng(102)
    cents = rng.normal(0, 1.0, size=(3000, 96))
    cents = (cents / np.linalg.norm(cents, axis=1, keepdims=True)).astype(np.float32)
    graph = orc.Hnsw.build(cents, M=16, efConstruction=60)
    q = _stress_queries(rng, cents, 24)
    q[:24] = cents[rng.choice(3000, 24)] + rng.normal(0, 0.05, size=(24, 96)).astype(np.float32)
    _walk_both(gpu(), graph, q, 32, 64)


def test_rows_the_bytes_represent_exactly(gpu):
    """Integer rows spanning exactly 0..255: step 1, quantisation slack ~0, so the bound is as tight as it
    gets, and with every row present twice there are exact ties at the boundary of the result set."""
    rng = np.random.default_rng(103)
    base = synth.sift_like(rng, 1024, 64)
    base[0, 0], base[1, 0] = 0.0, 255.0
    cents = np.concatenate([base, base])[rng.permutation(2048)]
    graph = orc.Hnsw.build(cents, M=12, efConstruction=60)
    q = np.concatenate([base[rng.choice(1024, 48)] + rng.integers(-3, 4, size=(48, 64)), base[:16]]).astype(np.float32)
    for k, ef in [(8, 8), (16, 40), (64, 64)]:
        _walk_both(gpu(), graph, q, k, ef)


@pytest.mark.parametrize("d", [16, 32, 64, 112])
def test_short_rows_are_zero_padded(gpu, d):
    rng = np.random.default_rng(104 + d)
    cents = synth.sift_like(rng, 1500, d)
    graph = orc.Hnsw.build(cents, M=8, efConstruction=40)
    _walk_both(gpu(), graph, _stress_queries(rng, cents, 12), 16, 48)


def test_degree_beyond_one_row_batch(gpu):
    """maxM = 64: link lanes 32..63 take the second 32-row batch of the node's block."""
    rng = np.random.default_rng(105)
    cents = synth.sift_like(rng, 3000, 128)
    graph = orc.Hnsw.build(cents, M=32, efConstruction=80)
    assert graph.counts.max() > 32
    _walk_both(gpu(), graph, _stress_queries(rng, cents, 16), 32, 64)


def test_link_lists_with_repeated_ids(gpu):
    """The same neighbour twice in a link list (never in a graph the reference built, but an uploaded one may): the
    second occurrence counts as visited (hnswalg.cpp:80-82) -- the device's visited set assumes distinct ids per
    list, so the upload drops the repeats; results must be those of the oracle walking the lists as given."""
    rng = np.random.default_rng(115)
    cents = synth.sift_like(rng, 4096, 128)
    built = orc.Hnsw.build(cents, M=16, efConstruction=60)
    counts = built.counts.copy()
    links = built.links.copy().reshape(len(counts), -1)
    maxM = links.shape[1]
    for i in range(len(counts)):
        c = int(counts[i])
        extra = min(maxM - c, 1 + i % 3)
        if c and extra > 0:
            links[i, c:c + extra] = links[i, rng.integers(0, c, extra)]   # repeats of links it already has
            counts[i] = c + extra
    graph = orc.Hnsw.from_arrays(counts, links, built.vectors, 16, built.enterpoint)
    assert (counts > built.counts).mean() > 0.5
    _walk_both(gpu(), graph, _stress_queries(rng, cents, 16), 32, 80)


@pytest.mark.parametrize("d", [144, 256])
def test_gather_form_beyond_128_dims(gpu, d):
    rng = np.random.default_rng(106 + d)
    cents = synth.sift_like(rng, 1200, d)
    graph = orc.Hnsw.build(cents, M=8, efConstruction=40)
    _walk_both(gpu(), graph, _stress_queries(rng, cents, 12), 16, 48)


def test_tables_without_a_filter(gpu):
    """d > 2048 and a constant table (no byte scale exists) run the unfiltered walk."""
    rng = np.random.default_rng(107)
    cents = synth.sift_like(rng, 200, 2064)
    graph = orc.Hnsw.build(cents, M=6, efConstruction=20)
    _walk_both(gpu(), graph, synth.sift_like(rng, 8, 2064), 8, 24)
    const = np.full((64, 32), 7.0, np.float32)
    graph = orc.Hnsw.build(const, M=4, efConstruction=10)
    _walk_both(gpu(), graph, synth.sift_like(rng, 6, 32), 4, 8)


LATE_CHILD = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import __graft_entry__ as ge
import test_gpu_walk_filter as m
pkg = ge.load_pkg()
made = []
def gpu():
    made.append(pkg.GpuIndex(0))
    return made[-1]
for ef, k in ((16, 16), (80, 32), (200, 64)):
    m.test_queries_outside_the_byte_range(gpu, ef, k)
m.test_signed_unit_vectors_d96(gpu)
m.test_rows_the_bytes_represent_exactly(gpu)
m.test_degree_beyond_one_row_batch(gpu)
m.test_link_lists_with_repeated_ids(gpu)
print("LATE OK", len(made))
'''


def test_late_visit_form_in_a_child_process():
    """The form the walk takes on graphs beyond 257 k nodes (survivors of the filter entered into the visited set at
    the next expansion; IVFHNSW_WALK_LATE_VISIT=1 forces it, read once per process): the cases above again.  Link
    lists with repeated ids are uploaded without the repeats (the reference skips them as visited)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, IVFHNSW_WALK_LATE_VISIT="1")
    r = subprocess.run([sys.executable, "-c", LATE_CHILD % dict(root=root)], capture_output=True, text=True, env=e,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "LATE OK" in r.stdout
