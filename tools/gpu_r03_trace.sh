#!/bin/bash
# kernel trace of the slab rehearsal (10 M cells, 8 slabs, few steps): where a slab's step goes
out=$GRAFT_REPO_ROOT/gpurun_out/r03_trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 8 2 16 > $out/slab8.json 2> $out/slab8.err
ls $out/slab8
head -40 $out/slab8/*kernel_stats.csv
