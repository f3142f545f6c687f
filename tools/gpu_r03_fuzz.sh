#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_fuzz; mkdir -p $out
FUZZ_LOG=$out/fuzz_parity_4000.jsonl timeout 2400 python tests/fuzz_parity.py 4000 50000 2>&1 | tail -3
timeout 1500 python tests/fuzz_slab.py 150 3000 > $out/fuzz_slab_150.log 2>&1; tail -2 $out/fuzz_slab_150.log
gzip -f $out/fuzz_parity_4000.jsonl
