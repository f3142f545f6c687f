#!/bin/bash
# The models' functors inlined at the grid kernels' call sites (YA_CALL_INLINED, include/solvers.cuh) and
# d_type as the enumeration the reference declares, against the builds without either (yalla_amd/ab/):
# configs 4 (renumbered) and 3.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_inline_ab; mkdir -p $out
timeout 600 python tools/make_state.py 4 /tmp/s4.npz > /dev/null 2>&1
timeout 600 python tools/make_state.py 3 /tmp/s3.npz > /dev/null 2>&1
for rep in 1 2; do
for tag in ${TAGS:-shipped}; do
  lib=$GRAFT_REPO_ROOT/yalla_amd/ab/libyalla_models_$tag.so
  [ $tag = shipped ] && lib=$GRAFT_REPO_ROOT/yalla_amd/libyalla_models.so
  YALLA_MODELS_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --model passive_growth_grid --state /tmp/s4.npz --renumber-every 10 > $out/cfg4_$tag.json 2>$out/cfg4_$tag.err
  python3 -c "import json,sys; d=json.load(open('$out/cfg4_$tag.json')); print('cfg4 $tag renumber-every 10', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  YALLA_MODELS_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --model branching_grid --state /tmp/s3.npz > $out/cfg3_$tag.json 2>$out/cfg3_$tag.err
  python3 -c "import json,sys; d=json.load(open('$out/cfg3_$tag.json')); print('cfg3 $tag', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
done
YALLA_MODELS_LIB=$GRAFT_REPO_ROOT/yalla_amd/ab/libyalla_models_base.so timeout 300 python bench.py --no-cpu-baseline > $out/springs_base.json 2>$out/springs_base.err
timeout 300 python bench.py --no-cpu-baseline > $out/springs_shipped.json 2>$out/springs_shipped.err
for t in base shipped; do python3 -c "import json,sys; d=json.load(open('$out/springs_$t.json')); print('springs $t', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"; done
timeout 900 python -m pytest tests/test_growth.py tests/test_parity_gpu.py tests/test_fast_arith_gpu.py tests/test_model_functors_independent.py -x -q -m gpu 2>&1 | tail -3
