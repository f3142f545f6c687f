#!/usr/bin/env python3
"""Fold the counters.json of gpu_profile_round.sh runs (gpurun_out/<tag>/counters.json, one entry each)
into profiles/r06_counters.json and copy each run's kernel stats / PMC means next to it:
    python tools/merge_counters.py gpurun_out/r05_springs_1M gpurun_out/r05_cfg4 ..."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r06_counters.json")
try:
    every = json.load(open(dst))
except (OSError, ValueError):
    every = {}
for run in sys.argv[1:]:
    entry = json.load(open(os.path.join(run, "counters.json")))
    every.update(entry)
    (key,) = entry.keys()
    shutil.copy(os.path.join(run, "pmc_summary.txt"), os.path.join(ROOT, "profiles", f"r06_pmc_{key}.txt"))
    shutil.copy(os.path.join(run, "stats", "k_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"r06_kernel_stats_{key}.csv"))
    shutil.copy(os.path.join(run, "bench.json"), os.path.join(ROOT, "profiles", f"r06_bench_{key}.json"))
    print(key, "<-", run)
json.dump(every, open(dst, "w"), indent=1)
