#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script of this repo: gpu_prof_py.sh <tag> <script> [args]
tag=$1; shift; script=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python3 $GRAFT_REPO_ROOT/$script "$@" > $out/out.json 2> $out/err.txt
cat $out/out.json
