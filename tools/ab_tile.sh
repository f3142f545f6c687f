#!/bin/bash
# Tile_solver old (worktree _old/) vs new: ab_tile.sh <tag> <cells>...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for cells in "$@"; do
  for d in _old .; do
    (cd $d && timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --model springs_tile --cells $cells > $out/s.json 2> $out/s.err) || tail -2 $out/s.err
    python3 -c "import json; d=json.load(open('$out/s.json')); print('$cells cells', '$d'.ljust(5), '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"
  done
done
