#!/bin/bash
# the randomised parity sweeps on the round's final build (new seeds each time: first seed = $1)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_fuzz; mkdir -p $out
first=${1:-90000}
FUZZ_LOG=$out/fuzz_parity.jsonl timeout 3000 python tests/fuzz_parity.py 4000 $first > $out/fuzz_parity.log 2>&1; tail -2 $out/fuzz_parity.log
gzip -f $out/fuzz_parity.jsonl
timeout 2400 python tests/fuzz_slab.py 200 $first > $out/fuzz_slab.log 2>&1; tail -2 $out/fuzz_slab.log
