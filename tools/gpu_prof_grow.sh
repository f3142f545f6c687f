#!/bin/bash
# kernel stats of the passive-growth run to --target cells: gpu_prof_grow.sh <tag> <target>
tag=$1; target=${2:-300000}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python3 $GRAFT_REPO_ROOT/tools/grow_to.py --target $target > $out/grow.json 2> $out/err.txt
cat $out/grow.json
