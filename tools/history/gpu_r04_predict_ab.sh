#!/bin/bash
# the predictor applied by the second build's binning kernel (this tree) against a launch of its own (_old/ = the
# commit before): parity suite, then the headline, 10 M cells and 1e5 cells, same box, interleaved
out=$GRAFT_REPO_ROOT/gpurun_out/r04_predict_ab; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_slab.py tests/test_fast_arith_gpu.py tests/test_growth.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
for args in "" "--cells-total 10000000" "--cells-total 100000"; do
  for rep in 1 2 3; do
    for which in old new; do
      dir=$GRAFT_REPO_ROOT; [ $which = old ] && dir=$GRAFT_REPO_ROOT/_old
      (cd $dir && timeout 300 python bench.py --no-cpu-baseline $args > $out/b.json 2> $out/b.err)
      python3 -c "import json; d=json.load(open('$out/b.json')); print('$which [$args]', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
    done
  done
done
