#!/bin/bash
# traced rehearsal (10 M cells, 8 slabs): per-slab, per-kernel device time and one slab's timeline
out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
rocprofv3 --kernel-trace --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err
SLAB_TIMELINE_RANK=1 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json 2> $out/timeline_rank1.txt
SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 2> $out/timeline_rank4.txt > /dev/null
rm -rf $out/slab8
