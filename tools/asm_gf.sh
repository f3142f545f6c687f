#!/bin/bash
# dump the gfx950 assembly of the springs grid_force kernel to /tmp/gf.s
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -DYALLA_NO_THRUST -I/root/repo/include -I/root/repo/yalla_amd/csrc -S --cuda-device-only /root/repo/yalla_amd/csrc/models.hip -o /tmp/models.s 2>&1 | grep -E "error" -A3
start=$(grep -n "^_ZN2ya10grid_forceI15HIP_vector_typeIfLj3EETnPFT_S3_S3_fiiEXadL_ZN6models6springE" /tmp/models.s | head -1 | cut -d: -f1)
tail -n +$start /tmp/models.s | awk '{print} /s_endpgm/{exit}' > /tmp/gf.s
wc -l /tmp/gf.s
echo "b128: $(grep -c ds_read_b128 /tmp/gf.s)  b96: $(grep -c ds_read_b96 /tmp/gf.s)"
tail -n +$start /tmp/models.s | grep -m4 -E "^; (NumVgprs|ScratchSize|Occupancy|LDSByteSize)"
