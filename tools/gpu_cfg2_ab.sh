#!/bin/bash
# config 2 (sorting, 10 k cells) and small springs systems: plain launches / graph replay, folding updates on / off
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.4g c-u/s  %.2f us/step  force %.1f us' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))" "$@"; }
for extra in "" "--graph 1" "--sorted-pipeline 2" "--sorted-pipeline 2 --graph 1"; do
  one --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 $extra
done
for n in 10000 30000 100000; do
  for extra in "" "--graph 1"; do one --cells-total $n --steps 200 $extra; done
done
