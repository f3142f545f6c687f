#!/bin/bash
# the randomised parity sweeps on round 6's sources (the grid summation order is drawn too: first seed = $1)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_fuzz; mkdir -p $out
first=${1:-310000}
FUZZ_LOG=$out/fuzz_parity.jsonl timeout 3000 python tests/fuzz_parity.py ${2:-4000} $first > $out/fuzz_parity.log 2>&1; tail -2 $out/fuzz_parity.log
gzip -f $out/fuzz_parity.jsonl
timeout 2400 python tests/fuzz_slab.py ${3:-200} $first > $out/fuzz_slab.log 2>&1; tail -2 $out/fuzz_slab.log
