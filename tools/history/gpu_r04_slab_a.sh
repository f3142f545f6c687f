#!/bin/bash
# Round 4, first look on the GPU: the slab tests, then the 10 M-cell / 8-slab rehearsal with the round's
# changes (cuts on cube planes balanced by own + 0.6 mirrored cells, the prefix sum over the slab's cube
# range only, the drift guard) against the quantile cuts of round 3 (YALLA_SLAB_PLAN=quantile), wall
# clock and -- under the kernel trace -- device time per slab.
out=$GRAFT_REPO_ROOT/gpurun_out/r04_slab_a; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_slab.py tests/test_core_abi_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.txt
timeout 900 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_10M_w8.json 2> $out/rehearsal_10M_w8.err; echo "planes rc=$?"
YALLA_SLAB_PLAN=quantile timeout 900 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_10M_w8_quantile.json 2> $out/rehearsal_10M_w8_quantile.err; echo "quantile rc=$?"
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for plan in planes quantile; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8_$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced_$plan.json 2> $out/slab8_$plan.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8_$plan/k_kernel_trace.csv 16 > $out/slab8_device_time_$plan.json 2> $out/timeline_rank4_$plan.txt
  cp $out/slab8_$plan/k_kernel_stats.csv $out/slab8_kernel_stats_$plan.csv
  rm -rf $out/slab8_$plan
done
unset YALLA_SLAB_PLAN YALLA_REHEARSAL_MARKERS
cd $GRAFT_REPO_ROOT
timeout 900 python tools/diag/springs_drift.py > $out/springs_drift.json 2> $out/springs_drift.err; echo "drift rc=$?"
ls -la $out
