#!/bin/bash
# old (worktree _old/) vs new at several system sizes: ab_sizes.sh <tag> <cells> [<cells> ...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for cells in "$@"; do
  for d in _old .; do
    (cd $d && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --cells $cells > $out/s.json 2> $out/s.err)
    python3 -c "import json; d=json.load(open('$out/s.json')); print('$cells cells', '$d'.ljust(5), '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"
  done
done
