#!/bin/bash
# the round's closing call: suite, profiles of the seven workloads, the default bench line, a fuzz sweep
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; grep -a -E "passed|failed" gpurun_out/r06_gpu_suite.txt | tail -1
bash tools/gpu_r06_final.sh $1 all > gpurun_out/r06_final.log 2>&1; tail -3 gpurun_out/r06_final.log
bash tools/gpu_r06_fuzz.sh ${2:-350000} 2000 100
