#!/bin/bash
# round 2, first GPU call: force-kernel A/B, end-to-end bench of both variants, self-launched 2-rank rehearsal
out=$GRAFT_REPO_ROOT/gpurun_out/r2a; mkdir -p $out
cd $GRAFT_REPO_ROOT
tools/micro/force_ab_run.sh r2a 1000000 10 30
for v in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-variant $v > $out/bench_v$v.json 2> $out/bench_v$v.err
  python3 -c "import json; d=json.load(open('$out/bench_v$v.json')); print('variant $v', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
timeout 600 python -m pytest tests/test_bench_cli.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $out/pytest.log
