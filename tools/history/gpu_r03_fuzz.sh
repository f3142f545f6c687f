#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_fuzz3; mkdir -p $out
FUZZ_LOG=$out/fuzz_parity_6000.jsonl timeout 3000 python tests/fuzz_parity.py 6000 70000 2>&1 | tail -3
gzip -f $out/fuzz_parity_6000.jsonl
bash tools/gpu_r03_slab_repeat.sh
