#!/bin/bash
# own cells put into cube order at every selection (the default) against the order they were adopted in
out=$GRAFT_REPO_ROOT/gpurun_out/r04_local_order; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_slab.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
for mode in 1 0; do
  YALLA_SLAB_LOCAL_ORDER=$mode timeout 600 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_order${mode}_$rep.json 2> /dev/null; echo "order=$mode rc=$?"
done
done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for mode in 1 0; do
  YALLA_SLAB_LOCAL_ORDER=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$mode -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/traced_order$mode.json 2> $out/t$mode.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/t$mode/k_kernel_trace.csv 27 > $out/device_time_order$mode.json 2> $out/timeline_rank4_order$mode.txt
  cp $out/t$mode/k_kernel_stats.csv $out/kernel_stats_order$mode.csv
  rm -rf $out/t$mode
done
