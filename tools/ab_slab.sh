#!/bin/bash
# plain vs slab path on one GPU: ab_slab.sh <tag> [extra bench args]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for mode in "" "--slab"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $mode "$@" > $out/s.json 2> $out/s$mode.err
  python3 -c "import json; d=json.load(open('$out/s.json')); print('mode [$mode]', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
