#!/bin/bash
# usage: ab.sh <tag> <variant> [variant...]  -> bench json per force variant
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-variant $v > $out/v$v.json 2> $out/v$v.err
  python3 -c "import json; d=json.load(open('$out/v$v.json')); print('variant $v', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
