#!/bin/bash
# usage: ab_old.sh <tag>  -> same-box comparison of this tree against the worktree _old/
tag=$1
# _old/ is a scratch worktree of an earlier commit, built in place: git worktree add -f _old <commit>; make -C _old/yalla_amd/csrc
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
  (cd _old && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/old.json 2> $out/old.err)
  python3 -c "import json; d=json.load(open('$out/old.json')); print('old', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/new.json 2> $out/new.err
  python3 -c "import json; d=json.load(open('$out/new.json')); print('new', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
