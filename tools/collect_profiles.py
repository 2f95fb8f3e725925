#!/usr/bin/env python3
"""Copies what a visit left under gpurun_out/TAG (scratch) into profiles/ (tracked): bench lines, rocprofv3 kernel stats, PMC summaries,
the GPU test log.  usage: python tools/collect_profiles.py TAG"""
import glob
import os
import shutil
import sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"
n = 0
for pat in ("bench_*.json", "kernel_stats_*.csv", "pmc_*.txt", "pmc_traffic.json", "pytest_gpu.log", "smoke.log", "ablate_*.txt", "ab_*.json"):
    for f in glob.glob(os.path.join(src, pat)):
        if os.path.getsize(f) == 0:
            continue
        name = os.path.basename(f)
        shutil.copy(f, os.path.join(dst, name if name == "pmc_traffic.json" else f"{tag}_{name}"))
        n += 1
for d in glob.glob(os.path.join(src, "pmc_*")):
    s = os.path.join(d, "summary.txt")
    if os.path.isdir(d) and os.path.exists(s):
        shutil.copy(s, os.path.join(dst, f"{tag}_{os.path.basename(d)}.txt"))
        n += 1
print("copied", n, "files to profiles/ with prefix", tag)
