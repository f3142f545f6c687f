#!/bin/bash
# Rebuild libyalla_models.so on the GPU box with extra -D flags and bench each.
# usage: sweep.sh <tag> "<flags1>" "<flags2>" ...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT/yalla_amd/csrc
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -DYALLA_NO_THRUST $flags -I../../include -I. -shared -o ../libyalla_models.so models.hip -L.. -lyalla_hip -Wl,-rpath,'$ORIGIN' -Wl,-Bsymbolic 2> $out/build$i.err || { echo "build failed: $flags"; tail -3 $out/build$i.err; continue; }
  (cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/b$i.json 2> $out/b$i.err)
  python3 -c "import json; d=json.load(open('$out/b$i.json')); print('[$flags]', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
