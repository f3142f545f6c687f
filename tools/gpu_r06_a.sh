#!/bin/bash
# round 6, first GPU call: the suite on the reference-order default, the bench line with preheat / tail_ab /
# headline clock, the clock-ramp diagnostic, the native whole-tiles vs tail A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_suite.txt 2>&1; tail -5 gpurun_out/r06a_suite.txt
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; tail -c 600 gpurun_out/r06a_bench.json
python bench.py --preheat-ms 0 --no-tail-ab-line --no-sustained-line --no-fast-tier-line --no-cpu-baseline > gpurun_out/r06a_bench_nopreheat.json 2>> gpurun_out/r06a_bench.err
python bench.py --no-tail-ab-line --no-sustained-line --no-fast-tier-line --no-cpu-baseline > gpurun_out/r06a_bench_preheat2.json 2>> gpurun_out/r06a_bench.err
python tools/diag/clock_ramp.py > gpurun_out/r06_clock_ramp.json 2> gpurun_out/r06_clock_ramp.err
for n in 1000000 1250000 150000 100000; do tools/micro/ab_bin/force_ab_tail $n 10 30 >> gpurun_out/r06a_tail_ab.jsonl; done
cat gpurun_out/r06a_tail_ab.jsonl
