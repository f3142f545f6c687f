#!/bin/bash
# traced rehearsal with and without a low-priority interior stream
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for mode in low same; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab5_$mode; mkdir -p $out
  if [ $mode = same ]; then export YA_INTERIOR_SAME_PRIORITY=1; fi
  rocprofv3 --kernel-trace --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json 2> $out/timeline_rank4.txt
  rm -rf $out/slab8
done
