#!/bin/bash
# Round 4: configs 4 and 3 with and without the model's renumbering (Solution::renumber every 10th step),
# on the same saved state; the renumbering tests; kernel statistics of both config-4 runs.
out=$GRAFT_REPO_ROOT/gpurun_out/r04_cfg34; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_growth.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.txt
timeout 600 python tools/make_state.py 4 /tmp/state4.npz > $out/state4.log 2>&1
timeout 300 python tools/make_state.py 3 /tmp/state3.npz > $out/state3.log 2>&1
for k in 0 10; do
  timeout 600 python bench.py --model passive_growth_grid --state /tmp/state4.npz --renumber-every $k --cpu-steps 2 > $out/bench_cfg4_renumber$k.json 2> $out/bench_cfg4_renumber$k.err; echo "cfg4 k=$k rc=$?"
  timeout 600 python bench.py --model branching_grid --state /tmp/state3.npz --renumber-every $k --cpu-steps 4 > $out/bench_cfg3_renumber$k.json 2> $out/bench_cfg3_renumber$k.err; echo "cfg3 k=$k rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for k in 0 10; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$k -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --model passive_growth_grid --state /tmp/state4.npz --renumber-every $k > $out/stats_bench$k.json 2> $out/stats$k.err
  cp $out/stats$k/k_kernel_stats.csv $out/kernel_stats_cfg4_renumber$k.csv; rm -rf $out/stats$k
done
python3 -c "
import json,glob
for f in sorted(glob.glob('$out/bench_cfg*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], '%.3e c-u/s'%d['value'], '%.3f ms/step'%d['ms_per_step'], 'force %.0f us'%d['roofline']['avg_launch_us'], 'cpu %.3e'%d['cpu_baseline']['value'])
    except Exception as e: print(f, 'unreadable', e)
"
