"""The multi-rank step on the GPU with REAL ranks: child processes in a gloo group, all on GPU 0 (a one-GPU
box; RCCL refuses two ranks on one device -- its calls are covered at world 1 by test_gpu_rccl_single_rank.py), each
running ivf-hnsw_amd/distributed.py::ShardedSearcher.step on its own shard handle, against the unsharded oracle; and
`bench.py --gpus 2` started from a bare shell (no WORLD_SIZE): the parent launches the ranks itself and relays the line.
Three processes touch the GPU at most (this one and two ranks)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_sharded_searcher_two_ranks_on_the_gpu(tmp_path):
    world, port = 2, 41000 + os.getpid() % 2000
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), str(r), str(world), str(port),
                               str(tmp_path)], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d exited %d:\n%s" % (r, p.returncode, outs[r][-3000:])
        f = tmp_path / ("rank%d.ok" % r)
        assert f.exists(), "rank %d: %s" % (r, (tmp_path / ("rank%d.fail" % r)).read_text())
        assert f.read_text().count(": True") == 8


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell, gloo ranks on the one GPU: ONE JSON line with n_gpus 2, the
    single-copy layout (1 group x 2 list shards) as `value`, the 2 x 1 replica layout beside it, and rank 0's results
    equal to the one-rank run of the same batch."""
    w = "synthetic-10M-pq16-nc16384-nprobe32"
    common = ["--steps", "3", "--warmup", "1", "--sustain-s", "0", "--workload", w, "--no-cpu-baseline", "--no-secondary"]
    env = _env()
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "20000", "--in-flight", "1",
                          "--no-split", "--dump", str(tmp_path / "one.npz")] + common, env=env, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    env["IVFHNSW_BENCH_BACKEND"] = "gloo"
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump", str(tmp_path / "two.npz")]
                         + common, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["list_shards"] == 2 and out["config"]["replica_groups"] == 1
    assert out["config"]["batch"] == 20000 and out["value"] > 0
    assert out["replica_groups"]["groups"] == 2 and out["replica_groups"]["list_shards_per_group"] == 1
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32))
