#!/bin/bash
# link_forces: does the order of the link array matter?
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r03_links; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for order in by-first shuffled by-min; do
  for cells in 100000 1000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/$order.$cells -o k -- python3 $GRAFT_REPO_ROOT/bench.py --model springs_links_grid --cells $cells --links-per-cell 3 --link-order $order --steps 40 --warmup 5 --cpu-steps 0 > $out/$order.$cells.json 2> $out/$order.$cells.err
    grep -i "link" $out/$order.$cells/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
    rm -rf $out/$order.$cells/*trace.csv
  done
done
