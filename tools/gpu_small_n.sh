#!/bin/bash
# Small systems on the GPU box: bench.py lines for force_variant 2 (one lane per cell) and
# 3 (grid_force_coop: 16 / 8 / 4 lanes per cell, chosen from n) at 10^4 .. 3 * 10^5 cells.
#   gpurun -- bash tools/gpu_small_n.sh <out-tag>     ->  gpurun_out/<out-tag>/small_n.json
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
rm -f $out/lines.jsonl
for v in 2 3; do
  python3 bench.py --no-cpu-baseline --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --force-variant $v | tail -1 >> $out/lines.jsonl
  for n in 10000 30000 50000 100000 300000; do
    # 20 steps like the headline line: springs contract, a long run is a denser system
    python3 bench.py --no-cpu-baseline --cells-total $n --steps 20 --force-variant $v | tail -1 >> $out/lines.jsonl
  done
done
python3 - <<PY
import json
rows = []
for l in open("$out/lines.jsonl"):
    d = json.loads(l)
    c = d["config"]
    rows.append({"model": c["model"], "cells": c["total_cells"], "grid_size": c["grid_size"], "steps": d["steps"],
                 "force_variant": c["force_variant"], "kernel": d["roofline"]["kernel"].split("<")[0],
                 "cell_updates_per_s": d["value"], "us_per_step": d["ms_per_step"] * 1e3,
                 "force_launch_us": d["roofline"]["avg_launch_us"]})
    print(rows[-1])
json.dump(rows, open("$out/small_n.json", "w"), indent=1)
PY
