#!/bin/bash
# every built configuration at several system sizes: force_ab_sizes.sh <out-tag> <sizes...>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT/tools/micro/ab_bin
for n in "$@"; do
  for exe in force_ab_*; do
    timeout 120 ./$exe $n 10 40 2>>$out/err.log | tee -a $out/ab.jsonl
  done
done
