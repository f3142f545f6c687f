#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command: gpu_kstats.sh <tag> [bench.py arguments ...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast-tier-line "$@" > $out/stats_bench.json 2> $out/stats.err
rm -f $out/stats/*agent_info.csv $out/stats/*kernel_trace.csv
python3 - $out/stats/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-70s %5s  avg %9.1f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
