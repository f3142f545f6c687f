#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06b_suite.txt 2>&1; tail -5 gpurun_out/r06b_suite.txt
for k in 1 2 3; do python bench.py --no-sustained-line --no-fast-tier-line --no-cpu-baseline > gpurun_out/r06b_bench_$k.json 2>> gpurun_out/r06b_bench.err; done
cp gpurun_out/r06_sum_order_gap.json gpurun_out/r06b_sum_order_gap.json 2>/dev/null
