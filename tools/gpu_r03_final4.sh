#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_final4; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 8 4 2 1; do
  timeout 600 tools/slab_rehearsal 10000000 $w 24 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
done
( cd /tmp && export TMPDIR=/tmp YALLA_REHEARSAL_MARKERS=1 && rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err )
python3 tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json
rm -f $out/slab8/k_kernel_trace.csv $out/slab8/k_agent_info.csv
