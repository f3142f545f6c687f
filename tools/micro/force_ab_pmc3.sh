#!/bin/bash
# Two PMC passes over every A/B executable (both kernels of each, told apart by their template arguments):
# force_ab_pmc3.sh <out-tag> [cells] [warm] [rounds]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for exe in $GRAFT_REPO_ROOT/tools/micro/ab_bin/force_ab_*; do
  tag=$(basename $exe)
  i=0
  for pass in \
    "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
    "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$tag.$i -o p -- $exe "$@" > /dev/null 2>$out/$tag.$i.err
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$out/$tag.*/*counter_collection.csv" grid_force > $out/$tag.txt
  rm -rf $out/$tag.1 $out/$tag.2
  echo "== $tag"; cat $out/$tag.txt
done
