#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2o; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_slab.py tests/test_bench_cli.py -m gpu -x -q 2>&1 | tail -6 | tee $out/pytest.log
for args in "" "--slab" "--slab --sequencing python" "--gpus 2 --backend gloo --cells-total 2000000" "--gpus 2 --backend gloo --cells-total 2000000 --sequencing python"; do
  YALLA_BENCH_DEVICE=0 timeout 600 python bench.py --no-cpu-baseline $args > $out/bench.json 2> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/bench.json')); print('$args', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])" || tail -5 $out/bench.err
done
