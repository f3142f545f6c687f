#!/bin/bash
# the springs benchmark with the MODEL's opt-in renumbering (its functor indexes nothing by id): headline and 10 M
out=$GRAFT_REPO_ROOT/gpurun_out/r04_springs_renumber; mkdir -p $out
cd $GRAFT_REPO_ROOT
for args in "" "--cells-total 10000000" "--arith fast"; do
  for k in 0 20 0 20; do
    timeout 300 python bench.py --no-cpu-baseline $args --renumber-every $k > $out/b.json 2> $out/b.err
    python3 -c "import json; d=json.load(open('$out/b.json')); print('[$args] renumber-every $k', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
