#!/bin/bash
# full GPU suite + bench of the three force variants + a fuzz sweep with its log
out=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/pytest.log
for v in 2 1; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-variant $v > $out/bench_v$v.json 2> $out/bench_v$v.err
  python3 -c "import json; d=json.load(open('$out/bench_v$v.json')); print('variant $v', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
FUZZ_LOG=$out/fuzz_parity_400.jsonl timeout 1200 python tests/fuzz_parity.py 400 5000 2>&1 | tail -3 | tee $out/fuzz.log
