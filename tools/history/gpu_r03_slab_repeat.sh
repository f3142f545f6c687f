#!/bin/bash
# five repeats of the 8-slab rehearsal (24 timed steps each): the slowest slab of a run is a different one each time
out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab_repeat; mkdir -p $out
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 5; do
  timeout 900 tools/slab_rehearsal 10000000 8 24 3 8 > $out/rehearsal_10M_w8_run$k.json 2> $out/run$k.err; echo "run $k rc=$?"
done
