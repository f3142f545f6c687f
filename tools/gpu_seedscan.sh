#!/bin/bash
# which YALLA_SEED values make the reference's statistical test_inits pass?
out=$GRAFT_REPO_ROOT/gpurun_out/seedscan; mkdir -p $out; cd $out
for s in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 32; do
  r=$(YALLA_SEED=$s timeout 120 $GRAFT_REPO_ROOT/oracle/_ref/test_inits 2>&1 | grep -E "ALL TESTS PASSED|not relaxed|wrong|too" | head -1)
  echo "seed $s: $r"
done | tee $out/scan.txt
