#!/bin/bash
# usage: ab_pipeline.sh <tag> [cells]  -> bench json with / without the sorted-space second stage
tag=$1; cells=${2:-1000000}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for v in 0 1 0 1; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --cells $cells --sorted-pipeline $v > $out/p$v.json 2> $out/p$v.err
  python3 -c "import json; d=json.load(open('$out/p$v.json')); print('sorted_pipeline $v', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
