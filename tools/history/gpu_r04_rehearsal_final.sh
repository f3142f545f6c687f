#!/bin/bash
# The round's rehearsal of north_star's 10 M-cell configuration on one GPU: W = 1, 2, 4, 8 slabs through the
# native sequencing (24 timed steps after 3, caller's migration every 8th, the drift guard asking for more),
# the 8-slab run under the kernel trace (device time per slab, the rehearsal's own transport set apart),
# the 1 M-cell system in 2 / 4 / 8 slabs, then the slab fuzz sweep on the final sources.
out=$GRAFT_REPO_ROOT/gpurun_out/r04_rehearsal_final; rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 1 2 4 8; do
  timeout 900 tools/slab_rehearsal 10000000 $w 24 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "10M w=$w rc=$?"
done
for rep in 2 3; do timeout 900 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_10M_w8_rep$rep.json 2> /dev/null; done
for w in 2 4 8; do
  timeout 300 tools/slab_rehearsal 1000000 $w 24 3 8 > $out/rehearsal_1M_w$w.json 2> $out/rehearsal_1M_w$w.err; echo "1M w=$w rc=$?"
done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for rep in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/slab8_traced_$rep.json 2> $out/slab8.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 27 > $out/slab8_device_time_$rep.json 2> $out/timeline_rank4_$rep.txt
  cp $out/slab8/k_kernel_stats.csv $out/slab8_kernel_stats_$rep.csv
  rm -rf $out/slab8
done
unset YALLA_REHEARSAL_MARKERS
cd $GRAFT_REPO_ROOT
timeout 1500 python tests/fuzz_slab.py 150 100 > $out/fuzz_slab_150.log 2>&1; echo "fuzz_slab rc=$?"; tail -1 $out/fuzz_slab_150.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1; grep -E "passed|failed" $out/pytest.txt
