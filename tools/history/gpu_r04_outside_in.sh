#!/bin/bash
# Round 4: ranges of tiles visited outside in (the shipped mapping) against storage order
# (yalla_amd/ab/libyalla_models_inorder.so, -DYA_XCD_OUTSIDE_IN=0): headline, 3e5, 10 M, fast tier; then the
# 8-slab rehearsal with device time per slab.
out=$GRAFT_REPO_ROOT/gpurun_out/r04_outside_in; mkdir -p $out
cd $GRAFT_REPO_ROOT
inorder=$GRAFT_REPO_ROOT/yalla_amd/ab/libyalla_models_inorder.so
for rep in 1 2 3; do
  for which in outside_in storage_order; do
    lib=""; [ $which = storage_order ] && lib=$inorder
    YALLA_MODELS_LIB=$lib timeout 300 python bench.py --no-cpu-baseline > $out/b_1M_${which}_$rep.json 2> $out/b.err
    python3 -c "import json; d=json.load(open('$out/b_1M_${which}_$rep.json')); print('1M $which rep $rep', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
for cells in 100000 300000 3000000 10000000; do
  for which in outside_in storage_order outside_in storage_order; do
    lib=""; [ $which = storage_order ] && lib=$inorder
    YALLA_MODELS_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --cells-total $cells > $out/b_${cells}_$which.json 2> $out/b.err
    python3 -c "import json; d=json.load(open('$out/b_${cells}_$which.json')); print('$cells $which', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for plan in quantile planes; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; else unset YALLA_SLAB_PLAN; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/traced_$plan.json 2> $out/t_$plan.err
  python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/t_$plan/k_kernel_trace.csv 27 > $out/device_time_$plan.json 2> /dev/null
  rm -rf $out/t_$plan
done
