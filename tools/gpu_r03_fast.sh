#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_fast; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fast_arith_gpu.py -x -q 2>&1 | tail -15
for a in exact fast exact fast; do
  timeout 300 python bench.py --no-cpu-baseline --arith $a > $out/bench_$a.json 2> $out/bench_$a.err
  python3 -c "import json; d=json.load(open('$out/bench_$a.json')); print('$a', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
for a in exact fast; do
  timeout 300 python bench.py --no-cpu-baseline --arith $a --cells-total 10000000 > $out/bench10M_$a.json 2> $out/bench10M_$a.err
  python3 -c "import json; d=json.load(open('$out/bench10M_$a.json')); print('10M $a', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
