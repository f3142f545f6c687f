#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_cfg4ab; mkdir -p $out
python tools/make_state.py 4 /tmp/s4.npz > /dev/null 2>&1
python tools/make_state.py 3 /tmp/s3.npz > /dev/null 2>&1
for tag in p2w1 p2w4 p1w1 p1w4 p2w1 p2w4; do
  lib=$GRAFT_REPO_ROOT/yalla_amd/ab/libyalla_models_$tag.so
  [ $tag = p1w4 ] && lib=$GRAFT_REPO_ROOT/yalla_amd/libyalla_models.so
  YALLA_MODELS_LIB=$lib python bench.py --no-cpu-baseline --model passive_growth_grid --state /tmp/s4.npz > $out/cfg4_$tag.json 2>$out/cfg4_$tag.err
  YALLA_MODELS_LIB=$lib python bench.py --no-cpu-baseline --model branching_grid --state /tmp/s3.npz --force-variant 2 > $out/cfg3_$tag.json 2>$out/cfg3_$tag.err
  for c in cfg4 cfg3; do python3 -c "import json,sys; d=json.load(open('$out/${c}_$tag.json')); print('$c $tag', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"; done
done
