#!/bin/bash
# Round 6: north_star's 10 M-cell configuration rehearsed on one GPU in the default (reference) summation order and,
# YALLA_SUM_ORDER=1, by plane (half tiles in both launches of a stage): W = 1, 2, 4, 8; two repeats of 8 each;
# the 8-slab run under the kernel trace (device time per slab).
out=$GRAFT_REPO_ROOT/gpurun_out/r06_rehearsal; rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 1 2 4 8 8; do
  timeout 900 tools/slab_rehearsal 10000000 $w 24 3 8 >> $out/rehearsal_10M_w$w.jsonl 2> $out/rehearsal_10M_w$w.err; echo "10M w=$w rc=$?"
done
for rep in 1 2; do YALLA_SUM_ORDER=1 timeout 900 tools/slab_rehearsal 10000000 8 24 3 8 >> $out/rehearsal_10M_w8_by_plane.jsonl 2> /dev/null; done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for order in 0 1; do
  YALLA_SUM_ORDER=$order timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/slab8_traced_order$order.json 2> $out/slab8.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 27 > $out/slab8_device_time_order$order.json 2> $out/timeline_rank4_order$order.txt
  cp $out/slab8/k_kernel_stats.csv $out/slab8_kernel_stats_order$order.csv
  rm -rf $out/slab8
done
ls $out
