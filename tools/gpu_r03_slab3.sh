#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab3; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 8; do
  timeout 600 tools/slab_rehearsal 10000000 $w 16 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
  python3 -c "
import json; d=json.load(open('$out/rehearsal_10M_w$w.json'))
print({k:d[k] for k in ('undivided_ms_per_step','slowest_slab_ms_per_step','critical_path_ms_per_step','projected_speedup_compute_only','parity')})
print('segment max', d['segment_max_ms'])"
done
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
rocprofv3 --kernel-trace --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err
python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json
cat $out/slab8_device_time.json | head -80
rm -f $out/slab8/k_kernel_trace.csv
