#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2n; mkdir -p $out
cd $GRAFT_REPO_ROOT
true
for args in "--cells-total 100000 --graph 0" "--cells-total 100000 --graph 1" "--cells-total 10000 --model sorting_grid --dt 0.05 --graph 0" "--cells-total 10000 --model sorting_grid --dt 0.05 --graph 1" "--cells-total 1000 --graph 0" "--cells-total 1000 --graph 1" "--graph 1" ""; do
  timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $args > $out/bench.json 2> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/bench.json')); print('$args', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])" || tail -3 $out/bench.err
done
