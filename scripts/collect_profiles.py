#!/usr/bin/env python3
"""Developer tool: copy the summaries of a scripts/round5_runs.sh (or bench_lines.sh) output directory into
profiles/<prefix>* (bench lines reduced to their JSON line, the rocprofv3 kernel-stats CSV, PMC traffic records,
the text records of the experiments).

  python scripts/collect_profiles.py gpurun_out/r03_final [--prefix r03_]
"""
import glob
import json
import os
import shutil
import sys

src = sys.argv[1]
prefix = sys.argv[sys.argv.index("--prefix") + 1] if "--prefix" in sys.argv else "r04_"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if not os.path.isfile(p) or f.endswith((".err", ".log")) and not f.startswith("train_"):
        continue
    if f.startswith("bench_line") and f.endswith(".json"):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]  # (gloo prints its rank census to stdout)
        if not lines:
            print("no JSON line in", f)
            continue
        json.loads(lines[-1])
        open(os.path.join(dst, prefix + f), "w").write(lines[-1] + "\n")
    elif f.endswith((".json", ".txt")) or f.startswith("train_"):
        shutil.copyfile(p, os.path.join(dst, prefix + f))
    else:
        continue
    print("->", prefix + f)
stats = glob.glob(os.path.join(src, "bench_stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copyfile(stats[0], os.path.join(dst, prefix + "bench_kernel_stats.csv"))
    print("->", prefix + "bench_kernel_stats.csv")
