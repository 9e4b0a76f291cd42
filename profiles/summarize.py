#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/..., scratch) into the small summaries kept under profiles/.

usage: summarize.py <kernel_stats.csv> <pmc counter_collection.csv>... > profiles/rNN_summary.md
Only this project's kernels (namespace ivfhnsw_gpu_impl) are kept; torch's data-generation kernels are dropped.
"""
import collections
import csv
import sys


def short(name):
    n = name.replace("void ", "").replace("ivfhnsw_gpu_impl::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    stats, pmcs = sys.argv[1], sys.argv[2:]
    print("## rocprofv3 --kernel-trace --stats (this project's kernels)\n")
    print("| kernel | calls | avg us | min us | max us | total ms |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats)):
        if "ivfhnsw_gpu_impl" not in r["Name"]:
            continue
        print("| %s | %s | %.1f | %.1f | %.1f | %.3f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / 1e6))
    for f in pmcs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ivfhnsw_gpu_impl" in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("\n## rocprofv3 --pmc (%s), average per dispatch\n" % f.split("/")[-2])
        print("| kernel | counter | dispatches | avg value |")
        print("|---|---|---|---|")
        for (k, c), v in sorted(agg.items()):
            print("| %s | %s | %d | %.6g |" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
