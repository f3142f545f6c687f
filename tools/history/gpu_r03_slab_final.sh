#!/bin/bash
# The round's rehearsal of north_star's 10 M-cell configuration on one GPU: W = 1, 2, 4, 8 slabs through
# the native sequencing (24 timed steps after 3, migration every 8th), then the 8-slab run under the
# kernel trace: device time per slab and kernel, one slab's timeline, kernel statistics.
out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab_final; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 1 2 4 8; do
  timeout 900 tools/slab_rehearsal 10000000 $w 24 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err
SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json 2> $out/timeline_rank4.txt
cp $out/slab8/k_kernel_stats.csv $out/slab8_kernel_stats.csv
rm -rf $out/slab8
