#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_dense; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o k -- python3 $GRAFT_REPO_ROOT/bench.py --cells 20000 --dist 0.02 --dt 0.000001 --grid-size 16 --steps 5 --warmup 2 --cpu-steps 0 --graph 0 > $out/bench.json 2> $out/err.txt
cut -d, -f1-4 $out/t/k_kernel_stats.csv | cut -c1-120 | head -14
tail -2 $out/err.txt
