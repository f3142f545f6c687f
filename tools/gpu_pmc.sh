#!/bin/bash
# rocprofv3 PMC passes over a short bench run.  usage: gpu_pmc.sh <tag> "<bench args>" "<counters pass 1>" ["<pass 2>" ...]
tag=$1; shift
bargs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $bargs > $out/pmc$i.json 2> $out/pmc$i.err
  ls $out/pmc$i
done
