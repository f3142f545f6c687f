#!/bin/bash
# the 8-slab rehearsal under the kernel trace: device time per slab, the rehearsal's transport set apart
out=$GRAFT_REPO_ROOT/gpurun_out/r04_slab_trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for rep in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/slab8_traced_$rep.json 2> $out/slab8.err
SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 27 > $out/slab8_device_time_$rep.json 2> $out/timeline_rank4_$rep.txt
cp $out/slab8/k_kernel_stats.csv $out/slab8_kernel_stats_$rep.csv
rm -rf $out/slab8
done
