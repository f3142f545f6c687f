#!/bin/bash
# randomised parity sweeps with the round's final sources: new seeds
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_fuzz2; mkdir -p $out
FUZZ_LOG=$out/fuzz_parity_6000.jsonl timeout 3000 python tests/fuzz_parity.py 6000 70000 2>&1 | tail -3
timeout 2400 python tests/fuzz_slab.py 300 5000 > $out/fuzz_slab_300.log 2>&1; tail -2 $out/fuzz_slab_300.log
gzip -f $out/fuzz_parity_6000.jsonl
