#!/bin/bash
# Links::link_forces under load: 100 k cells / 300 k links (and 1 M / 3 M), with and without links,
# plus the kernel-trace stats of the run with links.
out=$GRAFT_REPO_ROOT/gpurun_out/r2links; mkdir -p $out
cd $GRAFT_REPO_ROOT
for n in 100000 1000000; do
  for model in springs_grid springs_links_grid; do
    timeout 300 python bench.py --no-cpu-baseline --cells-total $n --model $model > $out/${model}_$n.json 2> $out/${model}_$n.err
    python3 -c "import json; d=json.load(open('$out/${model}_$n.json')); print('$model', $n, 'links', d['config']['links'], '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --cells-total 100000 --model springs_links_grid > $out/stats_bench.json 2> $out/stats.err
python3 - <<PY
import csv, glob
for p in glob.glob("$out/stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
