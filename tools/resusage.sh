#!/bin/bash
# usage: resusage.sh <kernel-name-substring> [extra hipcc flags]  -> registers / LDS / scratch of the matching kernels of models.hip
pat=$1; shift
cd /root/repo/yalla_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -DYALLA_NO_THRUST -I../../include -I. "$@" -c models.hip -o /tmp/models.o -Rpass-analysis=kernel-resource-usage 2>/tmp/res.txt
grep -E "error" -A5 /tmp/res.txt | head -30
grep -A11 "Function Name: .*$pat" /tmp/res.txt | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" | sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - - - | sed -e 's/Name: _ZN2ya[0-9]*//' | cut -c1-75,200-420
