#!/bin/bash
# bench.py at the sizes between the cooperative kernel and launches that fill the chip (round 5: every tile as
# half tiles while all of them are resident), plus configs 2 and 3
out=$GRAFT_REPO_ROOT/gpurun_out/r05_mid_sizes; mkdir -p $out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for args in "--cells-total 50000" "--cells-total 80000" "--cells-total 100000" "--cells-total 150000" "--cells-total 200000" "--cells-total 300000" \
              "--model branching_grid" "--model sorting_grid --cells-total 10000 --dt 0.05 --steps 300"; do
    timeout 400 python bench.py $args --no-cpu-baseline --no-fast-tier-line --no-sustained-line > $out/b.json 2> $out/b.err || { echo "bench failed: $args"; tail -3 $out/b.err; continue; }
    python3 -c "import json; d=json.load(open('$out/b.json')); print('$args', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'], d['roofline']['kernel'][:60])" | tee -a $out/lines.txt
  done
done
