#!/bin/bash
# every workgroup of a 1 M-cell (and 1.25 M-cell) force launch stamped, whole tiles only and with the tail of
# half tiles: how many are in flight over the launch's twenty twentieths (profiles/r05_force_trace.jsonl)
out=$GRAFT_REPO_ROOT/gpurun_out/r05_force_trace; mkdir -p $out
cd $GRAFT_REPO_ROOT
for n in 1000000 1250000; do
  for tail in 0 -1; do
    tools/micro/ab_bin/force_trace $n 3 $tail | gzip > $out/stamps_${n}_tail${tail}.csv.gz
  done
done
python3 tools/force_trace_summary.py $out/stamps_*.csv.gz > $out/summary.jsonl
python3 -c "
import json
for l in open('$out/summary.jsonl'):
    d=json.loads(l); print(d['file'], d['workgroups'], d['launch_us_hip_events'], d['mean_lifetime'], d['in_flight'])"
