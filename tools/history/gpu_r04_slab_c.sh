#!/bin/bash
# timelines of the end slabs (0, 7) and a middle one (4), quantile cuts and plane cuts
out=$GRAFT_REPO_ROOT/gpurun_out/r04_slab_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for plan in planes quantile; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/slab8_$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/slab8_traced_$plan.json 2> $out/slab8_$plan.err
  SLAB_TIMELINE_RANK=0,4,7 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8_$plan/k_kernel_trace.csv 27 > $out/slab8_device_time_$plan.json 2> $out/timelines_$plan.txt
  rm -rf $out/slab8_$plan
done
