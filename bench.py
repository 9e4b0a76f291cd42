#!/usr/bin/env python3
"""bench.py -- throughput of the IVFADC search hot path on MI355X.

One "step" = one pass of the whole hot path (rotate -> HNSW coarse walk -> PQ inner-product table ->
scan plan -> ADC list scan -> top-1 select) over one batch of queries that is already resident in HBM.

Workload (BASELINE.json configs[1]): synthetic 100M x 128-d, 2^17 centroids, PQ16, nprobe 32, 10 k-query
batch, at the paper operating point (nprobe, max_codes, efSearch) = (32, 10000, 80)
(reference examples/run_sift1b.sh:37-43).  SIFT1B-shaped synthetic data: see tests/synth.py.

N > 1 (WEAK scaling): the per-GPU work is fixed -- every GPU holds 100M codes and walks 10 k
queries -- so N GPUs search an N x 100M corpus with N x 2^17 centroids for N x 10 k queries per step (N = 8:
800M vectors, 2^20 centroids: the SIFT1B shape of BASELINE.json configs[3]/[4]).  The inverted lists are sharded
list-wise over the ranks (c % N == rank), every rank holds the replicated tables and graph; the coarse walk is
split over the ranks by query and all-gathered, every rank scans its shard for ALL queries, and the packed
(distance, scan position) keys are MIN-all-reduced over RCCL, labels MAX-all-reduced.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak (spec)

WORKLOADS = {
    # name: (n_total, nc, d, M, nprobe, max_codes, efSearch, nq)
    "synthetic-100M-pq16-nc131072-nprobe32": (100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000),
    "synthetic-10M-pq16-nc16384-nprobe32": (10_000_000, 1 << 14, 128, 16, 32, 10000, 80, 10000),
    "synthetic-1M-pq8-nc4096-nprobe8": (1_000_000, 4096, 128, 8, 8, 10 ** 9, 40, 2000),
    # Grouping + Pruning + OPQ at the reference's preset (examples/run_sift1b_grouping_OPQ.sh:7-53): nsubc 64
    "grouping-100M-pq16-nc131072-nsubc64-opq-pruning": (100_000_000, 1 << 17, 128, 16, 32, 10000, 80, 10000),
    "grouping-10M-pq16-nc16384-nsubc64-opq-pruning": (10_000_000, 1 << 14, 128, 16, 32, 10000, 80, 10000),
    # ... and at the reference's full size (BASELINE.json configs[3]) on ONE GPU; --no-cpu-baseline as for the 1B
    # shapes below (their CPU leg builds a 21-GB host copy of the lists: fine on the GPU box, not in a small container)
    "grouping-1B-pq16-nc993127-nsubc64-opq-pruning": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    # The 1B shapes of BASELINE.json configs[2] and [4] on ONE GPU (21 GB of lists; the CPU leg builds a host copy
    # of them -- pass --no-cpu-baseline where 25 GB of host memory are not to spare).  Parameters:
    # examples/run_sift1b.sh:37-43 (the two paper points) and examples/run_deep1b_OPQ.sh.
    "synthetic-1B-pq16-nc993127-nprobe32": (1_000_000_000, 993127, 128, 16, 32, 10000, 80, 10000),
    "synthetic-1B-pq16-nc993127-nprobe64": (1_000_000_000, 993127, 128, 16, 64, 30000, 100, 10000),
    "deep-1B-d96-opq-pq16-nc999973-nprobe128": (1_000_000_000, 999973, 96, 16, 128, 100000, 130, 10000),
}
WORKLOAD_FLAGS = {"deep-1B-d96-opq-pq16-nc999973-nprobe128": {"kind": "deep", "opq": True}}
DEFAULT_WORKLOAD = "synthetic-100M-pq16-nc131072-nprobe32"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("IVFHNSW_BENCH_WORKLOAD", DEFAULT_WORKLOAD))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale", type=int, default=0,
                    help="corpus / centroid / batch multiplier; default = number of GPUs (weak scaling)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight for the extra 'pipelined' figure (1 GPU only; 1 = skip it)")
    ap.add_argument("--dump", default=None, help="write rank 0's labels/distances of the last step to this .npz")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_total, nc, d, M, nprobe, max_codes, ef, nq = WORKLOADS[args.workload]
    if os.environ.get("IVFHNSW_BENCH_MAX_CODES"):   # experiment knob: longer / shorter scans per query
        max_codes = int(os.environ["IVFHNSW_BENCH_MAX_CODES"])
    scale = args.scale if args.scale > 0 else world
    n_total, nc, nq = n_total * scale, nc * scale, nq * scale
    if n_total >= 2 ** 32:
        raise SystemExit("corpus of %d vectors does not fit uint32 ids" % n_total)
    t0 = time.time()
    flags = WORKLOAD_FLAGS.get(args.workload, {})
    kind = flags.get("kind", "sift")
    tb = synth.make_throughput_tables(args.seed, nc, d, M, n_total, kind=kind)
    rng = np.random.default_rng(args.seed + 1)
    # queries: points near centroids, so that walks end in populated regions
    queries = (tb["centroids"][rng.choice(nc, nq)]
               + rng.normal(0, 12.0 if kind == "sift" else 0.03, size=(nq, d))).astype(np.float32)
    counts, links = synth.knn_graph_torch(tb["centroids"], 16, 32, device=dev)
    centroid_norms = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
    if rank == 0:
        log("[bench] tables + graph: %.1fs (avg degree %.1f)" % (time.time() - t0, counts.mean()))

    grouping = args.workload.startswith("grouping")
    opq_A = None
    vectors = tb["centroids"]
    if flags.get("opq") and not grouping:
        # OPQ: the graph holds rotated centroids at search time (rotate_quantizer, IndexIVF_HNSW.cpp:789-800);
        # for a throughput corpus the synthetic centroids simply ARE the rotated ones
        opq_A = synth.random_rotation(np.random.default_rng(args.seed + 4), d)
    if grouping:
        gt = synth.make_grouping_tables(args.seed + 3, tb, 64, device=dev)
        opq_A = synth.random_rotation(np.random.default_rng(args.seed + 4), d)
        # the graph holds rotated centroids at search time (rotate_quantizer, IndexIVF_HNSW.cpp:789-800)
        vectors = synth.rotated_vectors(tb["centroids"], opq_A)
    g = pkg.GpuIndex(local_rank)
    code_seed = args.seed + 7
    t0 = time.time()
    g.upload_ivf_synthetic(d, M, tb["offsets"], centroid_norms, tb["pq_centroids"], tb["norm_table"], code_seed,
                           opq_A=opq_A, shard_rank=rank, shard_world=world)
    g.upload_quantizer(counts, links, vectors, 0)
    if grouping:
        g.upload_grouping(64, gt["alphas"], gt["nn_centroid_idxs"], gt["subgroup_sizes"], gt["inter_centroid_dists"])
    if rank == 0:
        log("[bench] corpus on device: %.1fs, %.2f GB held" % (time.time() - t0, g.memory_bytes() / 1e9))

    # everything the timed region touches lives in HBM already
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_q = torch.from_numpy(queries).to(dev)
    d_dist = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    d_lab = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    sharded = pkg_dist.ShardedSearcher(g, rank, world, nq, nprobe, dev) if world > 1 else None

    def step():
        if world == 1:
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
    g.set_profiling(True)
    g.reset_stage_ms()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    if rank == 0:
        log("[bench] timed region: %d steps in %.3fs" % (args.steps, elapsed))
    stage = g.stage_ms()
    g.set_profiling(False)
    ncodes, nsegs = g.last_scan_counts()  # per step, this shard
    # the host-pointer entry point of the C ABI (queries in, distances and labels out over PCIe): never `value`,
    # reported beside it
    host_qps = None
    if world == 1:
        g.search(queries[:nq], 1, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        t_h = time.perf_counter()
        for _ in range(3):
            g.search(queries[:nq], 1, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        host_qps = 3 * nq / (time.perf_counter() - t_h)

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
        t_p = time.perf_counter()
        for i in range(args.steps):
            h, st, dd, ll = ctxs[i % len(ctxs)]
            h.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, efSearch=ef, do_pruning=grouping)
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t_p
        same = all(bool(torch.equal(ll, d_lab)) and bool(torch.equal(dd.view(torch.int32), d_dist.view(torch.int32)))
                   for _, _, dd, ll in ctxs[1:])
        pipe = {"in_flight": len(ctxs), "queries_per_s": round(nq * args.steps / el_p, 1),
                "ms_per_batch": round(el_p / args.steps * 1e3, 4), "results_equal_to_sequential": same}
        for h, _, _, _ in ctxs[1:]:
            h.close()
        g.set_stream(torch.cuda.current_stream().cuda_stream)

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    nc_t = torch.tensor([float(ncodes)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nc_t, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    ncodes_all = float(nc_t.item())

    lab_gpu = d_lab.cpu().numpy()[:, 0]
    dist_gpu = d_dist.cpu().numpy()[:, 0]

    if rank == 0 and args.dump:
        np.savez(args.dump, labels=lab_gpu, dist=dist_gpu)
    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        scan_ms, scan_n = stage["scan"]
        scan_avg_ms = scan_ms / max(1, scan_n)
        bytes_per_code = M + 1  # PQ code + norm code
        achieved = bytes_per_code * ncodes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
        # HBM bytes per launch of the scan kernel, measured by the rocprofv3 PMC passes committed under profiles/
        # (bench.py cannot collect counters itself); only reported for the workload they were taken on.
        traffic = None
        walk_traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
            if tj.get("workload") == args.workload and scale == 1 and world == 1:
                traffic = round(tj["bytes_per_launch"] / 1e9, 4)
                walk_traffic = round(tj["walk_bytes_per_launch"] / 1e9, 4)
        except (OSError, ValueError, KeyError):
            pass
        walk_ms, walk_n = stage["coarse"]
        walk_avg_ms = walk_ms / max(1, walk_n)
        out = {
            "metric": "queries/sec @ Recall@1, SIFT1B PQ16 nprobe=32; ADC scan HBM GB/s vs peak",
            "value": round(qps, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload + ("" if scale == 1 else " x%d" % scale), "n_vectors": n_total, "nc": nc, "d": d, "code_size": M,
                "nprobe": nprobe, "max_codes": max_codes, "efSearch": ef, "batch": nq, "k": 1,
                "coarse": "device HNSW walk", "codes_scored_per_query": round(ncodes_all / nq, 1),
                "sharding": "replicas=1" if world == 1 else "lists c%%%d, RCCL min-merge" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "scan_k1_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_unit": "GB per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/scan_traffic.json)",
                "algorithmic_gb_per_launch": round(bytes_per_code * ncodes / 1e9, 4),
                "bytes_per_code": bytes_per_code, "codes_per_launch": ncodes, "avg_launch_ms": round(scan_avg_ms, 4),
            },
            # the kernel most of the step is spent in: its own HBM bytes (PMC) over its launch time; the reference's
            # dist_calc count (dist_evals x 4d bytes) is added by the cpu_baseline leg, which counts it
            "roofline_walk": {
                "bound": "hbm", "kernel": "hnsw_walk_kernel", "avg_launch_ms": round(walk_avg_ms, 4),
                "traffic": walk_traffic, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "traffic_gbps": None if walk_traffic is None or walk_avg_ms <= 0
                else round(walk_traffic / (walk_avg_ms * 1e-3), 1),
            },
            "stage_ms_per_step": {k: round(v[0] / max(1, args.steps), 4) for k, v in stage.items()},
            "host_pointer_queries_per_s": None if host_qps is None else round(host_qps, 1),
            "pipelined": pipe,
        }

        if world == 1 and scale == 1 and not args.no_cpu_baseline:
            # the oracle (a port of the reference's CPU path) on the same corpus, bounded sample of the same batch
            from oracle import orc
            t0 = time.time()
            ids_h, codes_h, ncodes_h = synth.synthetic_codes(code_seed, tb["offsets"], M)
            graph = orc.Hnsw.from_arrays(counts, links, vectors, 16, 0)
            if grouping:
                ox = orc.Index(d, M, graph, tb["pq_centroids"], tb["norm_table"], tb["offsets"], ids_h, codes_h,
                               ncodes_h, centroid_norms, opq_A=opq_A, nsubc=64, alphas=gt["alphas"],
                               nn_centroid_idxs=gt["nn_centroid_idxs"], subgroup_sizes=gt["subgroup_sizes"],
                               inter_centroid_dists=gt["inter_centroid_dists"])
            else:
                ox = orc.Index(d, M, graph, tb["pq_centroids"], tb["norm_table"], tb["offsets"], ids_h, codes_h,
                               ncodes_h, centroid_norms, opq_A=opq_A)
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
            rw["algorithmic_gb_per_launch"] = round(walk_alg / 1e9, 4)
            rw["achieved"] = round(walk_alg / 1e9 / (walk_avg_ms * 1e-3), 1) if walk_avg_ms > 0 else 0.0
            rw["frac"] = round(rw["achieved"] / HBM_PEAK_GBPS, 4)
            rw["note"] = ("achieved counts the rows the reference's walk evaluates; the kernel's exact rejection "
                          "filter reads most of them as 32-byte rows instead, hence traffic < algorithmic")
            same_l = int((rl_all[:, 0] == lab_gpu).sum())
            same_d = int((rd_all[:, 0].view(np.uint32) == dist_gpu.view(np.uint32)).sum())
            out["parity"] = {"queries_checked": nq, "labels_equal": same_l, "distances_bit_equal": same_d}
            if same_l != nq or same_d != nq:
                log("[bench] PARITY FAILURE: %d/%d labels, %d/%d distances" % (same_l, nq, same_d, nq))
        print(json.dumps(out), flush=True)

    g.close()
    if world > 1:
        dist.destroy_process_group()
    if out is not None and "parity" in out and out["parity"]["labels_equal"] != out["parity"]["queries_checked"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
