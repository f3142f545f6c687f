#!/bin/bash
# config 4 with the model's renumbering: wavefronts per SIMD asked of the LOCAL_IDS build of grid_force_bits
# (yalla_amd/ab/libyalla_models_l<N>.so; the shipped library asks for 4), and the un-renumbered run
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_cfg4ab; mkdir -p $out
timeout 600 python tools/make_state.py 4 /tmp/s4.npz > /dev/null 2>&1
for rep in 1 2; do
for tag in l4 l3 l5 l6; do
  lib=$GRAFT_REPO_ROOT/yalla_amd/ab/libyalla_models_$tag.so
  [ $tag = l4 ] && lib=$GRAFT_REPO_ROOT/yalla_amd/libyalla_models.so
  for k in 10 0; do
    [ $k = 0 ] && [ $tag != l4 ] && continue
    YALLA_MODELS_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --model passive_growth_grid --state /tmp/s4.npz --renumber-every $k > $out/cfg4_${tag}_k$k.json 2>$out/cfg4_${tag}_k$k.err
    python3 -c "import json,sys; d=json.load(open('$out/cfg4_${tag}_k$k.json')); print('cfg4 $tag renumber-every $k', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
done
timeout 600 python -m pytest tests/test_growth.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
