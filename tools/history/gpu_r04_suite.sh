#!/bin/bash
# the whole GPU suite, the headline, the sustained figure, the rehearsal with the balanced cuts
out=$GRAFT_REPO_ROOT/gpurun_out/r04_suite; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --sustained --no-cpu-baseline > $out/bench_sustained.json 2> $out/bench_sustained.err; echo "sustained rc=$?"
timeout 600 python bench.py --sustained --no-cpu-baseline --arith fast > $out/bench_sustained_fast.json 2> $out/bench_sustained_fast.err
python3 -c "
import json
for f in ('bench','bench_sustained','bench_sustained_fast'):
    d=json.load(open('$out/%s.json'%f)); print(f, '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'], d.get('sustained',{}).get('shader_clock_mhz'))
"
for plan in balanced quantile; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; else unset YALLA_SLAB_PLAN; fi
  timeout 600 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_10M_w8_$plan.json 2> $out/rehearsal_$plan.err; echo "rehearsal $plan rc=$?"
done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for plan in balanced quantile; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; else unset YALLA_SLAB_PLAN; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/traced_$plan.json 2> $out/t_$plan.err
  SLAB_TIMELINE_RANK=4 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/t_$plan/k_kernel_trace.csv 27 > $out/device_time_$plan.json 2> $out/timeline_rank4_$plan.txt
  cp $out/t_$plan/k_kernel_stats.csv $out/kernel_stats_$plan.csv
  rm -rf $out/t_$plan
done
