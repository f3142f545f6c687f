#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2m; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $out/pytest.log
for args in "" "--cells-total 10000000" "--cells-total 100000" "--cells-total 10000 --model sorting_grid --dt 0.05"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $args > $out/bench.json 2> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/bench.json')); print('$args', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
