#!/bin/bash
# Run every built configuration on the GPU box: force_ab_run.sh <out-tag> [cells] [warm] [rounds] [dist]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT/tools/micro/ab_bin
for exe in force_ab_*; do
  timeout 120 ./$exe "$@" 2>>$out/err.log | tee -a $out/ab.jsonl
done
