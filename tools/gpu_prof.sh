#!/bin/bash
# usage: gpu_prof.sh <tag> "<bench args>"  -> kernel stats csv
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $1 > $out/bench.json 2> $out/err.txt
