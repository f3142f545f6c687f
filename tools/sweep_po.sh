#!/bin/bash
# Rebuild libyalla_models.so on the GPU box with extra -D flags; bench a Po_cell model and the growth run.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT/yalla_amd/csrc
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -DYALLA_NO_THRUST $flags -I../../include -I. -shared -o ../libyalla_models.so models.hip -L.. -lyalla_hip -Wl,-rpath,'$ORIGIN' -Wl,-Bsymbolic 2> $out/build.err || { echo "build failed: $flags"; continue; }
  (cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --no-cpu-baseline --model relu_po_grid --dt 0.01 > $out/b.json 2> $out/b.err)
  python3 -c "import json; d=json.load(open('$out/b.json')); print('[$flags] relu_po 1M rho9.8', '%.3f ms'%d['ms_per_step'])"
  (cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --no-cpu-baseline --model relu_po_grid --dt 0.01 --dist 0.75 > $out/b.json 2> $out/b.err)
  python3 -c "import json; d=json.load(open('$out/b.json')); print('[$flags] relu_po 1M rho2.9', '%.3f ms'%d['ms_per_step'])"
  (cd $GRAFT_REPO_ROOT && timeout 300 python tools/grow_to.py --target 400000 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('[$flags] growth 400k', '%.3f ms/step'%d['ms_per_step'], 'grow %.2f s'%d['growth_seconds'])")
done
