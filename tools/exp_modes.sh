#!/bin/bash
# timing experiments on grid_force: variants built into scratch worktrees (default: . and _m1)
out=$GRAFT_REPO_ROOT/gpurun_out/modes; mkdir -p $out
for d in ${@:-. _m1}; do
  (cd $d && timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --time-every 1 > $out/m.json 2> $out/m.err)
  python3 -c "import json; d=json.load(open('$out/m.json')); print('$d', '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
