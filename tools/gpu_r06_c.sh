#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06c_suite.txt 2>&1; tail -5 gpurun_out/r06c_suite.txt
cp gpurun_out/r06_sum_order_gap.json gpurun_out/r06c_sum_order_gap.json 2>/dev/null
for k in 1 2; do python bench.py --no-sustained-line --no-fast-tier-line --no-cpu-baseline --no-tail-ab-line > gpurun_out/r06c_bench_$k.json 2>> gpurun_out/r06c_bench.err; done
for k in 1 2 3; do python bench.py --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --no-cpu-baseline > gpurun_out/r06c_cfg2_$k.json 2>> gpurun_out/r06c_bench.err; done
for w in 1 2 4; do timeout 60 tools/micro/ab_bin/xcd_barrier $w 2000 >> gpurun_out/r06_xcd_barrier.jsonl; done
cat gpurun_out/r06_xcd_barrier.jsonl
