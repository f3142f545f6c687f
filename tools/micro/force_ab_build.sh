#!/bin/bash
# Build one force_ab executable per configuration: force_ab_build.sh tag:"-Dflags" ...
# (run here, on the CPU container; the binaries travel to the GPU box with gpurun)
cd /root/repo/tools/micro
mkdir -p ab_bin
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DYALLA_NO_THRUST \
    -I../../include -I../ab -I../../yalla_amd/csrc $flags -DAB_TAG="\"$tag\"" force_ab.hip -o ab_bin/force_ab_$tag \
    -L../../yalla_amd -lyalla_hip -Wl,-rpath,'$ORIGIN/../../../yalla_amd' 2>&1 | grep -E "error|warning: (?!unused)" -A3 &
done
wait
ls -la ab_bin
