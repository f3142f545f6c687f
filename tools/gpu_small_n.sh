#!/bin/bash
# Small systems on the GPU box: bench.py lines for force_variant 2 (one lane per cell) and
# 3 (grid_force_coop, 16 lanes per cell) at 10^4 .. 10^5 cells, and a kernel trace of one of them.
#   gpurun -- bash tools/gpu_small_n.sh <out-tag>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in 2 3; do
  python3 bench.py --no-cpu-baseline --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --force-variant $v | tail -1 >> $out/lines.jsonl
  for n in 10000 30000 100000 300000; do
    python3 bench.py --no-cpu-baseline --cells-total $n --steps 20 --force-variant $v | tail -1 >> $out/lines.jsonl
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --force-variant 3 > $out/trace.log 2>&1
python3 - <<PY
import json
for l in open("$out/lines.jsonl"):
    d = json.loads(l)
    print(d["config"].get("workload", "")[:60], "variant", d["config"].get("force_variant"), "%.3g c-u/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "force %.1f us" % d["roofline"]["avg_launch_us"])
PY
head -20 $out/trace/*kernel_stats.csv | cut -c1-60,200-400
