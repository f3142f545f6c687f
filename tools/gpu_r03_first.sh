#!/bin/bash
# Round 3, first GPU call: VALU calibration probe, slab rehearsal (10 M cells, 1/2/4/8 slabs), baseline bench.
out=$GRAFT_REPO_ROOT/gpurun_out/r03_first; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 tools/micro/ab_bin/valu_probe 20000 > $out/valu_probe.jsonl 2> $out/valu_probe.err
tail -n 60 $out/valu_probe.jsonl
timeout 120 tools/slab_rehearsal 200000 4 8 2 4 > $out/rehearsal_small.json 2> $out/rehearsal_small.err; echo "small rc=$?"
cat $out/rehearsal_small.json
for w in 8 4 2 1; do
  timeout 600 tools/slab_rehearsal 10000000 $w 16 3 16 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
  cat $out/rehearsal_10M_w$w.json
done
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/bench.json
