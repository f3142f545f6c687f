#!/bin/bash
# a longer randomised sweep on the round's final build, in ONE process each (new seeds: first seed = $1)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_fuzz_long; mkdir -p $out
first=${1:-130000}
FUZZ_LOG=$out/fuzz_parity.jsonl timeout 4000 python tests/fuzz_parity.py 10000 $first > $out/fuzz_parity.log 2>&1; tail -1 $out/fuzz_parity.log
gzip -f $out/fuzz_parity.jsonl
timeout 3000 python tests/fuzz_slab.py 400 $first > $out/fuzz_slab.log 2>&1; tail -1 $out/fuzz_slab.log
