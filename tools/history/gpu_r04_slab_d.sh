#!/bin/bash
# the 8-slab rehearsal with the slabs' turn order rotating (default) and left to the mutex; device time per slab
out=$GRAFT_REPO_ROOT/gpurun_out/r04_slab_d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for mode in rotate free; do
  if [ $mode = free ]; then export YALLA_REHEARSAL_FREE_ORDER=1; fi
  for plan in quantile planes; do
    if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; else unset YALLA_SLAB_PLAN; fi
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_$mode$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/traced_${mode}_$plan.json 2> $out/t_$mode$plan.err
    python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/t_$mode$plan/k_kernel_trace.csv 27 > $out/device_time_${mode}_$plan.json 2> /dev/null
    rm -rf $out/t_$mode$plan
  done
done
