#!/bin/bash
# VALU / wave-cycle counters of the A/B executables' force kernels
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for exe in $GRAFT_REPO_ROOT/tools/micro/ab_bin/force_ab_*; do
  tag=$(basename $exe)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/$tag -o p -- $exe 1000000 10 5 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$out/$tag/*counter_collection.csv" grid_force_bits > $out/$tag.txt
  rm -rf $out/$tag
  echo "== $tag"; cat $out/$tag.txt
done
