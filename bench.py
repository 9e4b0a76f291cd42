#!/usr/bin/env python3
"""bench.py -- throughput of the IVFADC search hot path on MI355X.

One "step" = one pass of the whole hot path (rotate -> HNSW coarse walk -> PQ inner-product table ->
scan plan -> ADC list scan -> top-1 select) over one batch of queries that is already resident in HBM.

Default workload = the shape BASELINE.json's metric is quoted on ("SIFT1B PQ16 nprobe=32"): synthetic
1B x 128-d, the reference's 993 127 centroids, PQ16, at the paper operating point (nprobe, max_codes, efSearch) =
(32, 10000, 80) (reference examples/run_sift1b.sh:37-43), batch 10 k.  It fits ONE GPU (26 GB).  SIFT1B-shaped
synthetic data: see tests/synth.py.  The same JSON line carries `secondary` results for configs[1] (100M, 2^17
centroids) and configs[2] (the 1B corpus at (64, 30000, 100)).

N > 1: the SAME 1B corpus, its inverted lists sharded list-wise over the N ranks by an owner table
(--partition: c % N, or a load-balanced spatial partition of the centroids, ivf-hnsw_amd/distributed.py); every rank
holds the replicated
tables and graph.  The coarse walk is split over the ranks by query and all-gathered, every rank scans the lists
it owns for ALL queries, and the packed (distance, scan position) keys are MIN-all-reduced over RCCL, the owner's
labels MAX-all-reduced.
  --scaling weak   (default): N x 10 k queries per step -- per-GPU work fixed (10 k walks, 1/N of N x the scan).
  --scaling strong           : 80 k queries per step at every N -- total work fixed.
  --list-shards S  (default N): the N ranks as N / S replica groups of S list shards each; a group holds the whole
                     corpus, serves its own S x 10 k batch, and its collectives stay inside the group.  S = N -- ONE copy
                     of the 1B-vector code array sharded across the N GPUs, the layout BASELINE.json's north_star names --
                     is what `value` reports.  The same line carries `replica_groups`: the same job as 2 groups x N/2
                     shards (a 1B x PQ16 index is 26 GB of a GPU's 288, and list sharding replicates per-(query, shard)
                     work: DESIGN.md 7), measured in the same run on the same ranks.

`python bench.py --gpus N` from a bare shell (WORLD_SIZE unset, N > 1) starts the N ranks itself: the parent -- before it
imports torch, loads the library or touches the GPU in any way -- runs `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process, which
prints rank 0's JSON line on the inherited stdout, and exits with its code.  Under an external launcher (WORLD_SIZE set)
the process is a rank and runs as before.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak (spec)
HBM_ACHIEVABLE_GBPS = 6290.0   # measured float4 copy rate of the same guide (HBM section): the practical ceiling of a stream
METRIC = "queries/sec @ Recall@1, SIFT1B PQ16 nprobe=32; ADC scan HBM GB/s vs peak"

WORKLOADS = {
    # name: (n_total, nc, d, M, nprobe, max_codes, efSearch, nq)
    "synthetic-100M-pq16-nc131072-nprobe32": (100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000),
    "synthetic-10M-pq16-nc16384-nprobe32": (10_000_000, 1 << 14, 128, 16, 32, 10000, 80, 10000),
    "synthetic-1M-pq8-nc4096-nprobe8": (1_000_000, 4096, 128, 8, 8, 10 ** 9, 40, 2000),
    # Grouping + Pruning + OPQ at the reference's preset (examples/run_sift1b_grouping_OPQ.sh:7-53): nsubc 64
    "grouping-100M-pq16-nc131072-nsubc64-opq-pruning": (100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000),
    "grouping-10M-pq16-nc16384-nsubc64-opq-pruning": (10_000_000, 1 << 14, 128, 16, 32, 10000, 80, 10000),
    # ... and at the reference's full size (BASELINE.json configs[3]) on ONE GPU
    "grouping-1B-pq16-nc993127-nsubc64-opq-pruning": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    # The 1B shapes of the metric line, configs[2] and configs[4] on ONE GPU (21 GB of lists; the CPU leg builds a
    # host copy of them -- pass --no-cpu-baseline where 25 GB of host memory are not to spare).  Parameters:
    # examples/run_sift1b.sh:37-43 (the two paper points) and examples/run_deep1b_OPQ.sh.
    "synthetic-1B-pq16-nc993127-nprobe32": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    "synthetic-1B-pq16-nc993127-nprobe64": (1_000_000_000, 993127, 128, 16, 64, 30000, 100, 10000),
    "deep-1B-d96-opq-pq16-nc999973-nprobe128": (1_000_000_000, 999973, 96, 16, 128, 100000, 130, 10000),
    # the metric's shape on CLUSTERED centroids (tight clusters of ~64, what k-means centroids of real descriptors look
    # like: a query's nearest centroids are each other's neighbours), same lists, codes and operating point
    "clustered-1B-pq16-nc993127-nprobe32": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    "clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    "clustered-100M-pq16-nc131072-nprobe32": (100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000),
}
WORKLOAD_FLAGS = {"deep-1B-d96-opq-pq16-nc999973-nprobe128": {"kind": "deep", "opq": True},
                  "clustered-1B-pq16-nc993127-nprobe32": {"kind": "clustered"},
                  "clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning": {"kind": "clustered"},
                  "clustered-100M-pq16-nc131072-nprobe32": {"kind": "clustered"}}
DEFAULT_WORKLOAD = "synthetic-1B-pq16-nc993127-nprobe32"
STRONG_BATCH = 80000


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Corpus:
    """Synthetic tables + graph of one workload and its device index (this rank's shard)."""

    def __init__(self, pkg, synth, name, seed, dev, local_rank, rank=0, world=1, scale=1, partition="mod", pkg_dist=None):
        n_total, nc, d, M, self.nprobe, self.max_codes, self.ef, self.nq = WORKLOADS[name]
        n_total, nc = n_total * scale, nc * scale
        if n_total >= 2 ** 32:
            raise SystemExit("corpus of %d vectors does not fit uint32 ids" % n_total)
        self.name, self.n_total, self.nc, self.d, self.M, self.seed = name, n_total, nc, d, M, seed
        flags = WORKLOAD_FLAGS.get(name, {})
        self.kind = flags.get("kind", "sift")
        self.grouping = "grouping" in name
        t0 = time.time()
        self.tb = tb = synth.make_throughput_tables(seed, nc, d, M, n_total, kind=self.kind)
        # the coarse graph: hnswlib's insertion loop with exact candidates on the device (ivfhnsw_gpu_build_graph; M 16,
        # maxM 32 as IndexIVF_HNSW.cpp:50 builds it), or -- rounds 1-2 -- a plain 16-NN graph with reverse links
        self.graph_kind = os.environ.get("IVFHNSW_BENCH_GRAPH", "insert")
        if self.graph_kind == "knn":
            self.counts, self.links = synth.knn_graph(tb["centroids"], 16, 32, device=dev)
        else:
            gb = pkg.GpuIndex(local_rank)
            self.counts, self.links = gb.build_graph(tb["centroids"], 16, 32, 64)
            gb.close()
        self.centroid_norms = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
        self.opq_A, self.gt, self.vectors = None, None, tb["centroids"]
        if flags.get("opq") and not self.grouping:
            # OPQ: the graph holds rotated centroids at search time (rotate_quantizer, IndexIVF_HNSW.cpp:789-800);
            # for a throughput corpus the synthetic centroids simply ARE the rotated ones
            self.opq_A = synth.random_rotation(np.random.default_rng(seed + 4), d)
        if self.grouping:
            self.gt = synth.make_grouping_tables(seed + 3, tb, 64, device=dev)
            self.opq_A = synth.random_rotation(np.random.default_rng(seed + 4), d)
            self.vectors = synth.rotated_vectors(tb["centroids"], self.opq_A)
        t_tab = time.time() - t0
        self.code_seed = seed + 7
        self.pkg, self.pkg_dist, self.local_rank, self.partition, self.t_tab = pkg, pkg_dist, local_rank, partition, t_tab
        self.g, self.owner = None, None
        self.upload(rank, world, verbose=rank == 0)

    def upload(self, rank, world, verbose=False):
        """This rank's shard of the corpus as `rank` of `world` list shards on the device (a new handle; the previous
        one, if any, stays open and is the caller's to close: another layout of the same tables and graph)."""
        tb, d, M = self.tb, self.d, self.M
        self.owner = None
        if world > 1:
            sizes = np.diff(tb["offsets"].astype(np.int64))
            load = self.pkg_dist.expected_list_load(sizes, self.counts, self.links)
            self.owner = self.pkg_dist.partition_lists(tb["centroids"], sizes, world, self.partition, load=load)
        t0 = time.time()
        self.g = g = self.pkg.GpuIndex(self.local_rank)
        g.upload_ivf_synthetic(d, M, tb["offsets"], self.centroid_norms, tb["pq_centroids"], tb["norm_table"],
                               self.code_seed, opq_A=self.opq_A, shard_rank=rank, shard_world=world,
                               list_owner=self.owner)
        g.upload_quantizer(self.counts, self.links, self.vectors, 0)
        if self.grouping:
            g.upload_grouping(64, self.gt["alphas"], self.gt["nn_centroid_idxs"], self.gt["subgroup_sizes"],
                              self.gt["inter_centroid_dists"])
        if verbose:
            log("[bench] %s: tables + graph %.1fs (avg degree %.1f), shard %d/%d on device %.1fs, %.2f GB held"
                % (self.name, self.t_tab, self.counts.mean(), rank, world, time.time() - t0, g.memory_bytes() / 1e9))
        return g

    def queries(self, nq, seed):
        rng = np.random.default_rng(seed)
        # points near centroids, so that walks end in populated regions
        return (self.tb["centroids"][rng.choice(self.nc, nq)]
                + rng.normal(0, 0.03 if self.kind == "deep" else 12.0, size=(nq, self.d))).astype(np.float32)

    def oracle(self, synth, orc):
        """The CPU port over a full host copy of the device's byte stream (the cpu_baseline / parity leg only)."""
        if getattr(self, "_host", None) is None:   # one host copy, shared by the two oracle builds
            self._host = synth.synthetic_codes(self.code_seed, self.tb["offsets"], self.M)
        ids_h, codes_h, ncodes_h = self._host
        graph = orc.Hnsw.from_arrays(self.counts, self.links, self.vectors, 16, 0)
        kw = {}
        if self.grouping:
            kw = dict(nsubc=64, alphas=self.gt["alphas"], nn_centroid_idxs=self.gt["nn_centroid_idxs"],
                      subgroup_sizes=self.gt["subgroup_sizes"], inter_centroid_dists=self.gt["inter_centroid_dists"])
        return orc.Index(self.d, self.M, graph, self.tb["pq_centroids"], self.tb["norm_table"], self.tb["offsets"],
                         ids_h, codes_h, ncodes_h, self.centroid_norms, opq_A=self.opq_A, **kw)


def timed_steps(torch, step, barrier, n):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    barrier()
    return time.perf_counter() - t0


def pmc_lds(workload):
    """What actually bounds the ADC scan (profiles/, same PMC passes): the LDS array's busy share.  Every code costs
    code_size random 4-byte gathers from the query's table; a 32-lane group of ds_read_b32 on 32 banks is a
    balls-in-bins draw (~3 x the conflict-free rate), and the counters put the LDS array at >= 90 % busy."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json"))).get("workloads", {}).get(workload)
        if e and "lds_idx_active_cycles" in e:
            busy = e["lds_idx_active_cycles"] / (e["cus"] * e["launch_cycles"])
            return {"busy_frac": round(busy, 3),
                    "bank_conflict_share_of_lds_cycles": round(e["lds_bank_conflict_cycles"] / e["lds_idx_active_cycles"], 3),
                    "lds_cycles_per_lds_instruction": round(e["lds_idx_active_cycles"] / e["lds_instructions"], 2),
                    "effective_clock_ghz": round(e["launch_cycles"] / (e["launch_us"] * 1e3), 3),
                    "source": e.get("lds_source", "profiles/scan_traffic.json"),
                    "note": "SQ_LDS_IDX_ACTIVE / (CUs x GRBM_GUI_ACTIVE / 8): the scan is bound by LDS bank conflicts of its "
                            "table gathers (DESIGN.md 3.1), which is why `frac` against HBM stops near 0.6"}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    return None


def scan_roofline(g, M, stage, traffic_gb, steps):
    """`roofline` of the dominant kernel: algorithmic bytes (code_size + 1 per scored code) per launch
    over the kernel's average launch time (HIP events the library records around that launch on its stream).  A batch
    of >= 8192 queries runs as two uneven parts on two streams (capi.cpp search_dev_split): two scan launches per step,
    and the figures are per LAUNCH (what rocprofv3's average is, too): the step's codes and kernel time over both."""
    ncodes_step, _ = g.last_scan_counts()
    scan_ms, scan_n = stage["scan"]
    parts = max(1, int(round(scan_n / max(1, steps))))
    ncodes = ncodes_step / parts
    avg_ms = scan_ms / max(1, scan_n)
    bpc = M + 1
    achieved = bpc * ncodes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "launches_per_step": parts,
        "bound": "hbm", "kernel": g.last_scan_kernel(), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": None if traffic_gb is None else round(traffic_gb / parts, 4),
        "traffic_unit": "GB per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of the step's scan launches / launches, "
                        "profiles/scan_traffic.json)",
        "algorithmic_gb_per_launch": round(bpc * ncodes / 1e9, 4), "bytes_per_code": bpc, "codes_per_launch": int(ncodes),
        "avg_launch_ms": round(avg_ms, 4),
        # beside the contract's figure, never instead of it: what HBM3E really streams on this part (the guide's float4 copy,
        # MI355X_MICROARCH.md: 6.29 TB/s = 79 % of the 8 TB/s `peak`)
        "achievable_gbps": HBM_ACHIEVABLE_GBPS, "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBPS, 4),
    }


def pmc_traffic(workload):
    """HBM bytes per launch measured by the rocprofv3 PMC passes committed under profiles/ (bench.py cannot collect
    counters itself); only reported for the workload they were taken on."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
        e = tj.get("workloads", {}).get(workload)
        if e:
            return round(e["bytes_per_launch"] / 1e9, 4), round(e["walk_bytes_per_launch"] / 1e9, 4)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def launch_ranks(n):
    """`bench.py --gpus N` from a bare shell: start N fresh rank processes and relay rank 0's line and the exit code.
    Runs BEFORE this process has imported torch, loaded libivfhnsw_hip.so or made any GPU call -- a process that has
    initialised the GPU must never be replaced or forked into ranks -- and the ranks are children, not an exec."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL / dmabuf IPC on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without WORLD_SIZE: starting the ranks: %s" % (n, " ".join(cmd[1:9])))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~0.4 s (200 steps of ~1.85 ms); 20 steps are 37 ms, shorter than the clocks take to settle
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=os.environ.get("IVFHNSW_BENCH_WORKLOAD", DEFAULT_WORKLOAD))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--list-shards", type=int, default=0,
                    help="N > 1: list shards per replica group (a divisor of N; default N = one copy of the codes over the "
                         "N GPUs).  The N ranks form N / S groups; a group holds the whole corpus in S list shards and "
                         "serves its own batches, collectives stay inside the group (DESIGN.md 7)")
    ap.add_argument("--no-replica-layout", action="store_true",
                    help="N > 1: skip the extra `replica_groups` measurement (2 groups x N/2 shards on the same ranks)")
    ap.add_argument("--no-split", action="store_true", help="run every batch in ONE part (ivfhnsw_gpu_set_batch_split 0): "
                    "profiling runs that want one launch shape per kernel")
    ap.add_argument("--no-one-part", action="store_true", help="skip the extra one_part measurement (profiling runs of the "
                    "default path: every launch in the process then has the two-part shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[2] extra results")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("IVFHNSW_BENCH_SCALING", "weak"),
                    help="N > 1: weak = N x 10 k queries per step (per-GPU work fixed); strong = 80 k queries at every N")
    ap.add_argument("--batch", type=int, default=0, help="queries per step (overrides the workload's / the scaling mode's)")
    ap.add_argument("--partition", choices=("spatial", "mod"), default="mod",
                    help="owner table of the list shards: c %% N, or the load-balanced spatial bisection (DESIGN.md 7)")
    ap.add_argument("--scale", type=int, default=1,
                    help="legacy weak scaling: corpus and centroid multiplier (round 1 ran the 100M workload x N)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight for the extra 'pipelined' figure (1 GPU only; 1 = skip it)")
    ap.add_argument("--sustain-s", type=float, default=1.2, help="length of the extra sustained measurement, seconds")
    ap.add_argument("--dump", default=None, help="write rank 0's labels/distances of the last step to this .npz")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # oracle threads must not spin inside a CPU quota
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    import synth

    pkg = ge.load_pkg()
    import importlib
    pkg_dist = importlib.import_module("ivfhnsw_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # rehearsal on a single-GPU box: IVFHNSW_BENCH_BACKEND=gloo puts every rank on GPU 0 and runs the same
    # collectives over gloo (RCCL refuses two ranks on one device); the driver's real runs use nccl (= RCCL).
    backend = os.environ.get("IVFHNSW_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:  # counting devices does not initialise the GPU
        raise SystemExit("--gpus %d over RCCL needs %d GPUs, this box shows %d (one-GPU rehearsal: IVFHNSW_BENCH_BACKEND=gloo)"
                         % (world, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # replica groups x list shards: rank r is shard r % S of group r // S.  Default S = N: one copy of the code arrays
    # sharded over the N GPUs (north_star); the 2 x N/2 layout is measured beside it (`replica_groups`).
    S = args.list_shards if args.list_shards > 0 else world
    S = min(S, world)
    if world % S:
        raise SystemExit("--list-shards %d does not divide --gpus %d" % (S, world))
    R = world // S
    S2 = world // 2 if (world > 1 and world % 2 == 0 and S == world and not args.no_replica_layout) else 0

    def make_group(shards):
        """This rank's process group in the layout of `shards` list shards per replica group (None = the world)."""
        mine = None
        if world > 1 and 1 < shards < world:
            for gi in range(world // shards):  # every rank creates every group (torch.distributed's rule)
                pg = dist.new_group(ranks=list(range(gi * shards, (gi + 1) * shards)))
                if gi == rank // shards:
                    mine = pg
        return mine

    srank, gidx = rank % S, rank // S
    group = make_group(S)
    group2 = make_group(S2) if S2 else None
    C = Corpus(pkg, synth, args.workload, args.seed, dev, local_rank, srank, S, args.scale, args.partition, pkg_dist)
    g, d, M, nprobe, ef = C.g, C.d, C.M, C.nprobe, C.ef
    max_codes = C.max_codes
    if os.environ.get("IVFHNSW_BENCH_MAX_CODES"):   # experiment knob: longer / shorter scans per query
        max_codes = int(os.environ["IVFHNSW_BENCH_MAX_CODES"])
    grouping = C.grouping
    # nq = the batch of THIS rank's group (every group its own queries); the job's batch is R of them
    if args.batch > 0:
        nq = args.batch
    elif args.scaling == "strong":
        nq = STRONG_BATCH // R
    else:
        nq = C.nq * S * args.scale
    nq_job = nq * R
    queries = C.queries(nq, args.seed + 1 + 7919 * gidx)

    # everything the timed region touches lives in HBM already
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.no_split:
        g.set_batch_split(0)
    d_q = torch.from_numpy(queries).to(dev)
    d_dist = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    d_lab = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    sharded = pkg_dist.ShardedSearcher(g, srank, S, nq, nprobe, dev, group=group) if S > 1 else None

    def step():
        if S == 1:
            g.search_dev(nq, 1, d_q, d_dist, d_lab, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        else:
            # coarse walk for this rank's slice of the batch -> all-gather -> scan own shard for all queries ->
            # MIN over shards of the packed keys -> owner resolves labels -> MAX over shards
            sharded.step(d_q, d_dist, d_lab, max_codes, ef, do_pruning=grouping)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: the first steps size the per-batch workspace (hipMalloc) and bring the clocks up; they run
    # whatever --warmup says, so that a short run (--warmup 0) does not time allocations
    for _ in range(3):
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events over the timed region around the kernel the roofline is about (the scan): an event pair costs ~7 us of
    # stream time, six pairs per step were 2.3 % of the step.  The other stages' times (the walk's too) come from a few
    # extra steps with every stage bracketed, after the region.
    prof_handles = sharded.handles() if sharded is not None else [g]
    for h_ in prof_handles:
        h_.set_profiling(2)
        h_.reset_stage_ms()
    elapsed = timed_steps(torch, step, barrier, args.steps)
    if rank == 0:
        log("[bench] timed region: %d steps in %.3fs" % (args.steps, elapsed))
    def stage_sum():
        tot = {}
        for h_ in prof_handles:
            for k_, (ms_, n_) in h_.stage_ms().items():
                a_, b_ = tot.get(k_, (0.0, 0))
                tot[k_] = (a_ + ms_, b_ + n_)
        return tot

    stage = stage_sum()
    n_aux = 10
    for h_ in prof_handles:
        h_.set_profiling(1)
        h_.reset_stage_ms()
    timed_steps(torch, step, barrier, n_aux)
    stage_all = stage_sum()
    for h_ in prof_handles:
        h_.set_profiling(False)
    for name_, (ms_, n_) in stage_all.items():  # small stages: per step from the extra steps, scaled to the region
        if name_ != "scan":
            stage[name_] = (ms_ / n_aux * args.steps, int(round(n_ / n_aux * args.steps)))
    ncodes, nsegs = (sharded or g).last_scan_counts()  # per step, this shard (both parts of a two-part sharded step)
    lab_gpu = d_lab.cpu().numpy()[:, 0].copy()
    dist_gpu = d_dist.cpu().numpy()[:, 0].copy()

    # the same step sustained for >= 1 s (the K-step region above is tens of milliseconds: too short for the clocks
    # and the driver's SMI sampler to mean anything); stage events off, results must not change
    n_sus = max(args.steps, int(math.ceil(args.sustain_s / max(1e-6, elapsed / args.steps))))
    t_sus = timed_steps(torch, step, barrier, n_sus) if args.sustain_s > 0 else 0.0
    sus_same = bool((d_lab.cpu().numpy()[:, 0] == lab_gpu).all())

    # N > 1: the same job as 2 replica groups x N/2 list shards, on the same ranks in the same run (`replica_groups`,
    # reported beside `value`, never as it): another upload of the same tables, graph and byte stream
    replica = None
    if S2:
        s2rank, g2idx = rank % S2, rank // S2
        g2 = C.upload(s2rank, S2, verbose=rank == 0)
        g2.set_stream(torch.cuda.current_stream().cuda_stream)
        nq2 = (STRONG_BATCH // 2) if args.scaling == "strong" else C.nq * S2 * args.scale
        if args.batch > 0:
            nq2 = max(1, args.batch * S2 // S)
        q2 = torch.from_numpy(C.queries(nq2, args.seed + 1 + 7919 * g2idx)).to(dev)
        dd2 = torch.empty((nq2, 1), dtype=torch.float32, device=dev)
        ll2 = torch.empty((nq2, 1), dtype=torch.int64, device=dev)
        sh2 = pkg_dist.ShardedSearcher(g2, s2rank, S2, nq2, nprobe, dev, group=group2) if S2 > 1 else None

        def step2():
            if S2 == 1:
                g2.search_dev(nq2, 1, q2, dd2, ll2, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
            else:
                sh2.step(q2, dd2, ll2, max_codes, ef, do_pruning=grouping)

        for _ in range(3 + args.warmup):
            step2()
        barrier()
        el2 = timed_steps(torch, step2, barrier, args.steps)
        red2 = torch.tensor([el2], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(red2, op=dist.ReduceOp.MAX)
        el2 = float(red2[0].item())
        replica = {"groups": 2, "list_shards_per_group": S2, "batch_per_group": nq2, "batch": 2 * nq2,
                   "value": round(2 * nq2 * args.steps / el2, 1), "unit": "queries/s",
                   "ms_per_step": round(el2 / args.steps * 1e3, 4),
                   "note": "every group holds the whole corpus in %d list shard(s) and serves its own batch; collectives "
                           "stay inside the group" % S2}
        if sh2 is not None:
            sh2.close()
        g2.close()
        g.set_stream(torch.cuda.current_stream().cuda_stream)

    # the host-pointer entry point of the C ABI (queries in, distances and labels out over PCIe): SURVEY 8d's
    # end-to-end figure; never `value` (the contract wants inputs resident in HBM), reported beside it
    host_qps = None
    if world == 1:
        g.search(queries, 1, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        t_h = time.perf_counter()
        for _ in range(3):
            g.search(queries, 1, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        host_qps = 3 * nq / (time.perf_counter() - t_h)

    # `value` is what a plain search_dev call delivers: since round 3 a batch of >= 8192 queries runs as two uneven parts on
    # two streams inside the call (ivfhnsw_gpu_set_batch_split; the parts' sizes follow the call's parameters).  Reported beside
    # it, never as it: the same batch in ONE part (IVFHNSW_SPLIT=0) -- one launch per kernel, the shape rounds 1-2 measured
    # and the one the scan's rate reads best on.
    one_part = None
    parts_used = g.last_batch_parts() if world == 1 else (nq, 0)
    split_active = world == 1 and nq >= 8192 and os.environ.get("IVFHNSW_SPLIT", "") != "0" and not args.no_split
    if split_active and not args.no_one_part:
        g.set_batch_split(0)
        sp_d, sp_l = torch.empty_like(d_dist), torch.empty_like(d_lab)
        for _ in range(3):
            g.search_dev(nq, 1, d_q, sp_d, sp_l, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        n_sp = max(args.steps, n_sus // 4)
        g.set_profiling(2)
        g.reset_stage_ms()
        t_sp = time.perf_counter()
        for _ in range(n_sp):
            g.search_dev(nq, 1, d_q, sp_d, sp_l, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        el_sp = time.perf_counter() - t_sp
        st1 = g.stage_ms()
        g.set_profiling(1)      # ... and every stage of the one-part shape, in a few extra steps
        g.reset_stage_ms()
        for _ in range(10):
            g.search_dev(nq, 1, d_q, sp_d, sp_l, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        st1_all = g.stage_ms()
        g.set_profiling(False)
        g.set_batch_split(int(os.environ.get("IVFHNSW_SPLIT", "") or 1000))
        r1 = scan_roofline(g, M, st1, None, n_sp)
        one_part = {"steps": n_sp, "queries_per_s": round(nq * n_sp / el_sp, 1), "ms_per_batch": round(el_sp / n_sp * 1e3, 4),
                    "scan_avg_launch_ms": r1["avg_launch_ms"], "scan_gbps": r1["achieved"], "scan_frac_of_hbm_peak": r1["frac"],
                    "stage_ms_per_step": {k_: round(v_[0] / 10.0, 4) for k_, v_ in st1_all.items()},
                    "results_equal_to_default": bool(torch.equal(sp_l, d_lab)) and
                    bool(torch.equal(sp_d.view(torch.int32), d_dist.view(torch.int32)))}
        # leave the handle as the timed region left it: the last call a two-part one (scan counts, kernel name)
        g.search_dev(nq, 1, d_q, sp_d, sp_l, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()

    # Serving form (reported beside `value`, never as it): --in-flight batches on as many streams, each on its own
    # view of the index (ivfhnsw_gpu_create_view: same tables, own workspace).  The walk is ALU-bound and ends in a
    # tail of partly filled SIMDs, the scan is HBM-bound: batches in flight overlap the two.
    pipe = None
    if world == 1 and args.in_flight > 1:
        ctxs = [(g, torch.cuda.current_stream(), d_dist, d_lab)]
        for _ in range(args.in_flight - 1):
            v = g.view()
            st = torch.cuda.Stream(device=dev)
            v.set_stream(st.cuda_stream)
            ctxs.append((v, st, torch.empty_like(d_dist), torch.empty_like(d_lab)))
        torch.cuda.synchronize()
        for h, st, dd, ll in ctxs:   # warm-up: every view sizes its workspace
            h.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        n_pipe = max(args.steps, n_sus // 2)
        t_p = time.perf_counter()
        for i in range(n_pipe):
            h, st, dd, ll = ctxs[i % len(ctxs)]
            h.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t_p
        same = all(bool(torch.equal(ll, d_lab)) and bool(torch.equal(dd.view(torch.int32), d_dist.view(torch.int32)))
                   for _, _, dd, ll in ctxs[1:])
        pipe = {"in_flight": len(ctxs), "steps": n_pipe, "queries_per_s": round(nq * n_pipe / el_p, 1),
                "ms_per_batch": round(el_p / n_pipe * 1e3, 4), "results_equal_to_sequential": same}
        for h, _, _, _ in ctxs[1:]:
            h.close()
        g.set_stream(torch.cuda.current_stream().cuda_stream)

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    red = torch.tensor([elapsed, t_sus], dtype=torch.float64, device=red_dev)
    nc_sum = torch.tensor([float(ncodes)], dtype=torch.float64, device=red_dev)
    nc_max = nc_sum.clone()
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(nc_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(nc_max, op=dist.ReduceOp.MAX)
    elapsed, t_sus = float(red[0].item()), float(red[1].item())
    ncodes_all = float(nc_sum.item())

    if rank == 0 and args.dump:
        np.savez(args.dump, labels=lab_gpu, dist=dist_gpu)
    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq_job * args.steps / elapsed
        single = world == 1 and args.scale == 1 and args.batch == 0 and args.scaling == "weak"
        traffic, walk_traffic = pmc_traffic(args.workload) if single else (None, None)
        walk_ms, walk_n = stage["coarse"]
        walk_avg_ms = walk_ms / max(1, walk_n)
        if one_part is not None:   # two overlapping walk launches per step say nothing per launch: the one-part shape's walk
            walk_avg_ms = one_part["stage_ms_per_step"]["coarse"]
        walk_gbps = None if walk_traffic is None or walk_avg_ms <= 0 else round(walk_traffic / (walk_avg_ms * 1e-3), 1)
        if world == 1:
            sharding = "replicas=1"
        else:
            sharding = ("%s: %d replica group(s) x %d list shard(s) by owner table (%s partition), %s" % (
                        "ONE copy of the code arrays sharded list-wise over the %d GPUs" % world if R == 1 else "replica groups",
                        R, S, args.partition,
                        "RCCL min-merge" if backend == "nccl" else "%s min-merge (single-GPU rehearsal)" % backend))
        out = {
            "metric": METRIC,
            "value": round(qps, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload + ("" if args.scale == 1 else " x%d" % args.scale), "n_vectors": C.n_total,
                "nc": C.nc, "d": d, "code_size": M, "nprobe": nprobe, "max_codes": max_codes, "efSearch": ef,
                "batch": nq_job, "batch_per_group": nq, "k": 1, "coarse": "device HNSW walk", "codes_scored_per_query": round(ncodes_all / nq_job, 1),
                "sharding": sharding, "replica_groups": R, "list_shards": S,
            },
            "value_is": "device-resident rate (queries and results in HBM, bench contract); SURVEY 8d's end-to-end "
                        "figure incl. H2D/D2H is host_pointer_queries_per_s",
            "sustained": {"steps": n_sus, "seconds": round(t_sus, 3),
                          "queries_per_s": round(nq_job * n_sus / t_sus, 1) if t_sus > 0 else None,
                          "results_unchanged": sus_same},
            "roofline": scan_roofline(g, M, stage, traffic, args.steps),
            "roofline_lds": pmc_lds(args.workload) if single else None,
            # the kernel most of the step is spent in.  frac = its own HBM-side bytes (PMC) over its launch time, as a
            # fraction of the HBM peak; the reference's dist_calc accounting (dist_evals x 4d bytes, rows
            # the kernel's exact rejection filter mostly does not read) is reported separately by the cpu_baseline leg
            "roofline_walk": {
                "bound": "hbm, random kilobyte pieces (DESIGN.md 3.2: at 993 127 nodes the walk runs near the rate HBM serves them; on cache-resident graphs its instruction stream binds)", "kernel": "hnsw_walk_kernel",
                "shape": "the whole batch in one launch (one_part): the default path's two walk launches overlap each other",
                "traffic_calibration": "FETCH_SIZE x 2 + WRITE_SIZE; the x 2 measured on the walk's own access shapes "
                                       "(profiles/r03_fetch_calibration.md: 2.000 for link rows, byte rows and float rows alike)",
                "avg_launch_ms": round(walk_avg_ms, 4), "traffic": walk_traffic, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "traffic_gbps": walk_gbps, "frac": None if walk_gbps is None else round(walk_gbps / HBM_PEAK_GBPS, 4),
            },
            "stage_ms_per_step": {k: round(v[0] / max(1, args.steps), 4) for k, v in stage.items()},
            "stage_note": "sums of launch durations per step; with the batch split the two parts' launches OVERLAP, so the sum "
                          "exceeds the step (one_part.stage_ms_per_step has the one-launch-per-kernel shape).  scan: HIP "
                          "events over the timed region; the other stages: %d extra steps with every stage bracketed (six "
                          "event pairs per step cost 2.3 %% of it)" % n_aux,
            "host_pointer_queries_per_s": None if host_qps is None else round(host_qps, 1),
            "pipelined": pipe,
            "batch_split": {"parts": 2 if split_active else 1, "part_queries": list(parts_used) if split_active else None,
                            "note": "a plain ivfhnsw_gpu_search_dev call of >= 8192 queries runs as two uneven parts on two "
                                    "streams (default since ABI 9; IVFHNSW_SPLIT=0 = one part): `value`, `roofline` "
                                    "(17 B x ALL codes of the step over the SUMMED scan-launch time) and the stage "
                                    "times are this default path's"},
            "one_part": one_part,
        }
        if replica is not None:
            out["replica_groups"] = replica
        if world > 1:
            out["shard_balance"] = {"codes_per_step_max_rank": float(nc_max.item()),
                                    "codes_per_step_mean_rank": round(ncodes_all / world, 1),
                                    "max_over_mean": round(float(nc_max.item()) / max(1.0, ncodes_all / world), 4)}

        ox = None
        if world == 1 and args.scale == 1 and not args.no_cpu_baseline:
            # the oracle (a port of the reference's CPU path) on the same corpus, bounded sample of the same batch
            from oracle import orc
            t0 = time.time()
            ox = C.oracle(synth, orc)
            ox.set_params(nprobe, max_codes, ef, do_pruning=grouping)
            log("[bench] host corpus for the CPU baseline: %.1fs" % (time.time() - t0))
            # the GPU box gives one GPU's share of the host (16 cores); more OpenMP threads than that only spin
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            ncores = max(1, min(ncores, int(os.environ.get("IVFHNSW_BENCH_CPU_THREADS", "16"))))
            # serial, exactly how the reference drivers run it (tests/test_ivfhnsw_sift1b.cpp:193-208)
            ns = min(nq, 1000)
            ox.search_batch(queries[:32], 1, 1)
            t0 = time.perf_counter()
            rd, rl, _, _, st = ox.search_batch(queries[:ns], 1, 1)
            t_serial = time.perf_counter() - t0
            log("[bench] cpu serial: %d queries in %.2fs" % (ns, t_serial))
            # all host cores, OpenMP over queries (an extension: the reference has no parallel search path)
            reps = max(1, int(10.0 / max(1e-3, t_serial * (nq / ns) / ncores)))
            t0 = time.perf_counter()
            for _ in range(reps):
                rd_all, rl_all, _, _, _ = ox.search_batch(queries, 1, ncores)
            t_par = (time.perf_counter() - t0) / reps
            log("[bench] cpu openmp x%d: %d x %d queries, %.2fs each" % (ncores, reps, nq, t_par))
            out["cpu_baseline"] = {
                "value": round(nq / t_par, 1), "unit": "queries/s", "cores": ncores, "kind": "port",
                "sample": "%d x the full %d-query batch, OpenMP over queries on %d threads; serial (1 thread, first %d "
                          "queries, as the reference drivers loop): %.1f queries/s" % (reps, nq, ncores, ns, ns / t_serial),
                "serial_value": round(ns / t_serial, 1),
            }
            # reference accounting of the walk: every fstdistfunc call reads one 4d-byte row (hnswalg.cpp:57,91)
            evals_q = float(st.dist_evals) / ns
            walk_alg = evals_q * 4 * d * nq
            rw = out["roofline_walk"]
            rw["dist_evals_per_query"] = round(evals_q, 1)
            rw["reference_rows_gb_per_launch"] = round(walk_alg / 1e9, 4)
            rw["reference_rows_gbps"] = round(walk_alg / 1e9 / (walk_avg_ms * 1e-3), 1) if walk_avg_ms > 0 else 0.0
            rw["note"] = ("reference_rows_* counts the 4d-byte rows the reference's walk evaluates (its dist_calc); the "
                          "kernel's exact rejection filter settles most of them from byte rows, so this is not a "
                          "fraction of anything -- frac is traffic_gbps / peak")
            same_l = int((rl_all[:, 0] == lab_gpu).sum())
            same_d = int((rd_all[:, 0].view(np.uint32) == dist_gpu.view(np.uint32)).sum())
            out["parity"] = {"queries_checked": nq, "labels_equal": same_l, "distances_bit_equal": same_d}
            if same_l != nq or same_d != nq:
                log("[bench] PARITY FAILURE: %d/%d labels, %d/%d distances" % (same_l, nq, same_d, nq))
            # north_star's tolerance clause, measured on this batch: the same port with the float associations g++ 11.4
            # emits for the reference's loops under the reference's own flags (-Ofast -march=native, CMakeLists.txt:22;
            # oracle/liborc_ofast.so, tests/test_oracle_float_order.py) -- how many top-1 ids would differ from a
            # reference binary built today
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("orc_ofast", os.path.join(ROOT, "oracle", "orc.py"))
                orc_of = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(orc_of)
                orc_of.LIB_PATH = os.path.join(ROOT, "oracle", "liborc_ofast.so")
                if os.path.exists(orc_of.LIB_PATH):
                    oy = C.oracle(synth, orc_of)
                    oy.set_params(nprobe, max_codes, ef, do_pruning=grouping)
                    fd, fl, _, _, _ = oy.search_batch(queries, 1, ncores)
                    out["float_order"] = {
                        "what": "top-1 of the CPU port in source order vs the same port in the association of the "
                                "reference's -Ofast -march=native build (fma in the L2 loop, tree lane sum, pairwise ADC)",
                        "queries": nq, "top1_ids_differ": int((fl[:, 0] != rl_all[:, 0]).sum()),
                        "distances_differ_in_bits": int((fd[:, 0].view(np.uint32) != rd_all[:, 0].view(np.uint32)).sum()),
                        "north_star_tolerance": "Recall@1 within +-0.1 % for float ties"}
            except Exception as e:  # the measuring stick must never fail the bench
                log("[bench] float_order leg skipped: %r" % (e,))

        # secondary results in the same line: configs[2]'s operating point on the same corpus, configs[1], the metric's
        # shape on CLUSTERED centroids, and a recall-bearing index built by the library's own pipeline
        if single and args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
            out["secondary"] = []
            try:
                nthr = min(16, len(os.sched_getaffinity(0)))
            except AttributeError:
                nthr = 8

            def measure(gg, s_nq, s_dq, s_np, s_mc, s_ef, pruning=False):
                """(entry, labels, distances) of one workload on handle gg: the default call, every stage bracketed."""
                s_dd = torch.empty((s_nq, 1), dtype=torch.float32, device=dev)
                s_ll = torch.empty((s_nq, 1), dtype=torch.int64, device=dev)

                def s_step():
                    gg.search_dev(s_nq, 1, s_dq, s_dd, s_ll, s_np, s_mc, efSearch=s_ef, do_pruning=pruning)

                for _ in range(3 + args.warmup):
                    s_step()
                torch.cuda.synchronize()
                gg.set_profiling(True)
                gg.reset_stage_ms()
                t_s = timed_steps(torch, s_step, torch.cuda.synchronize, args.steps)
                s_stage = gg.stage_ms()
                gg.set_profiling(False)
                e = {"value": round(s_nq * args.steps / t_s, 1), "unit": "queries/s",
                     "ms_per_step": round(t_s / args.steps * 1e3, 4), "nprobe": s_np, "max_codes": s_mc, "efSearch": s_ef,
                     "batch": s_nq, "roofline": scan_roofline(gg, 16, s_stage, None, args.steps),
                     "stage_ms_per_step": {k: round(v[0] / max(1, args.steps), 4) for k, v in s_stage.items()}}
                return e, s_ll.cpu().numpy()[:, 0], s_dd.cpu().numpy()[:, 0]

            def parity_of(r_l, r_d, lg, dg, n_chk):
                return {"queries_checked": n_chk, "labels_equal": int((r_l[:n_chk, 0] == lg[:n_chk]).sum()),
                        "distances_bit_equal": int((r_d[:n_chk, 0].view(np.uint32) == dg[:n_chk].view(np.uint32)).sum())}

            # (1) the primary corpus at configs[2]'s point (its host copy is still there for the full-corpus oracle)
            name = "synthetic-1B-pq16-nc993127-nprobe64"
            _, _, _, _, s_np, s_mc, s_ef, s_nq = WORKLOADS[name]
            ent, lg, dg = measure(g, s_nq, d_q, s_np, s_mc, s_ef)
            ent["workload"] = name
            if ox is not None:
                ox.set_params(s_np, s_mc, s_ef)
                r_d, r_l, _, _, _ = ox.search_batch(queries[:2000], 1, nthr)
                ent["parity"] = parity_of(r_l, r_d, lg, dg, 2000)
            out["secondary"].append(ent)
            ox = None
            C._host = None   # 21 GB of host lists: the next corpora bring their own (sparse) views

            # (2) configs[1], (3) the metric's shape on clustered centroids: own corpora, oracle on a SPARSE host view (only
            # the lists its own walk probes for the sample are materialised, tests/test_gpu_configs_1b.py)
            for name in ("synthetic-100M-pq16-nc131072-nprobe32", "clustered-1B-pq16-nc993127-nprobe32",
                         "clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning"):
                CC = Corpus(pkg, synth, name, args.seed, dev, local_rank)
                CC.g.set_stream(torch.cuda.current_stream().cuda_stream)
                _, _, _, _, s_np, s_mc, s_ef, s_nq = WORKLOADS[name]
                sq = CC.queries(s_nq, args.seed + 1)
                ent, lg, dg = measure(CC.g, s_nq, torch.from_numpy(sq).to(dev), s_np, s_mc, s_ef, pruning=CC.grouping)
                ent["workload"] = name
                ent["centroids"] = CC.kind
                if not args.no_cpu_baseline:
                    from oracle import orc
                    n_chk = 2000
                    arrays = synth.synthetic_codes_sparse(CC.code_seed, CC.tb["offsets"], CC.M)
                    graph = orc.Hnsw.from_arrays(CC.counts, CC.links, CC.vectors, 16, 0)
                    kw = {}
                    if CC.grouping:
                        kw = dict(nsubc=64, alphas=CC.gt["alphas"], nn_centroid_idxs=CC.gt["nn_centroid_idxs"],
                                  subgroup_sizes=CC.gt["subgroup_sizes"], inter_centroid_dists=CC.gt["inter_centroid_dists"],
                                  opq_A=CC.opq_A)
                    oxx = orc.Index(CC.d, CC.M, graph, CC.tb["pq_centroids"], CC.tb["norm_table"], CC.tb["offsets"],
                                    arrays[0], arrays[1], arrays[2], CC.centroid_norms, **kw)
                    oxx.set_params(s_np, s_mc, s_ef, do_pruning=CC.grouping)
                    _, _, cid0, _, _ = oxx.search_batch(sq[:n_chk], 1, nthr)   # pass 1: which lists does its walk probe
                    probed = cid0.ravel()
                    synth.synthetic_codes_sparse(CC.code_seed, CC.tb["offsets"], CC.M, probed[probed < CC.nc], into=arrays)
                    r_d, r_l, _, _, _ = oxx.search_batch(sq[:n_chk], 1, nthr)
                    ent["parity"] = parity_of(r_l, r_d, lg, dg, n_chk)
                    graph.free()
                    del arrays, oxx
                out["secondary"].append(ent)
                CC.g.close()
                del CC

            # (4) Recall@1 itself (the metric is "queries/sec @ Recall@1"; the reference's drivers print it,
            # tests/test_ivfhnsw_sift1b.cpp:173-215): a 10 M-vector index of clustered data built by the library's own
            # pipeline on the device -- insertion-loop graph, Lloyd-trained code books, assignment + encoding, exact ground
            # truth (tests/synth.py make_recall_corpus) -- searched by the device path and by the CPU port
            name = "clustered-10M-pq16-nc16384-recall"
            t0 = time.time()
            rc = synth.make_recall_corpus(pkg, args.seed + 11, 16384, 10_000_000, nq=10000, device=local_rank,
                                          query_noise=24.0, log=log)
            gr_ = pkg.GpuIndex(local_rank)
            gr_.set_stream(torch.cuda.current_stream().cuda_stream)
            gr_.upload_ivf(rc["d"], rc["code_size"], rc["offsets"], rc["ids"], rc["codes"], rc["norm_codes"],
                           rc["centroid_norms"], rc["pq_centroids"], rc["norm_table"])
            gr_.upload_quantizer(rc["counts"], rc["links"], rc["centroids"], 0)
            ent, lg, dg = measure(gr_, 10000, torch.from_numpy(rc["queries"]).to(dev), 32, 10000, 80)
            ent["workload"] = name
            ent["built_in_s"] = round(time.time() - t0, 1)
            rec = {"device": round(float((lg == rc["gt"]).mean()), 4)}
            if not args.no_cpu_baseline:
                from oracle import orc
                graph = orc.Hnsw.from_arrays(rc["counts"], rc["links"], rc["centroids"], 16, 0)
                oxx = orc.Index(rc["d"], rc["code_size"], graph, rc["pq_centroids"], rc["norm_table"], rc["offsets"],
                                rc["ids"], rc["codes"], rc["norm_codes"], rc["centroid_norms"])
                oxx.set_params(32, 10000, 80)
                t0 = time.perf_counter()
                r_d, r_l, _, _, _ = oxx.search_batch(rc["queries"], 1, nthr)
                t_cpu = time.perf_counter() - t0
                ent["parity"] = parity_of(r_l, r_d, lg, dg, 10000)
                rec["cpu_port"] = round(float((r_l[:, 0] == rc["gt"]).mean()), 4)
                rec["equal"] = rec["cpu_port"] == rec["device"]
                ent["cpu_port_queries_per_s"] = round(10000 / t_cpu, 1)
                graph.free()
            rec["ground_truth"] = "exact nearest base vector of every query (ivfhnsw_gpu_knn, f32 MFMA brute force)"
            ent["recall_at_1"] = rec
            out["secondary"].append(ent)
            gr_.close()
        print(json.dumps(out), flush=True)

    if sharded is not None:
        sharded.close()
    g.close()
    if world > 1:
        dist.destroy_process_group()
    bad = False
    if out is not None:
        for ent in [out] + out.get("secondary", []):
            p = ent.get("parity")
            if p and (p["labels_equal"] != p["queries_checked"] or p["distances_bit_equal"] != p["queries_checked"]):
                bad = True
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
