#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_slab2; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_slab.py tests/test_native.py -x -q -m gpu 2>&1 | tail -8
for w in 8 4 2; do
  timeout 600 tools/slab_rehearsal 10000000 $w 16 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
  python3 -c "
import json; d=json.load(open('$out/rehearsal_10M_w$w.json'))
print({k:d[k] for k in ('undivided_ms_per_step','slowest_slab_ms_per_step','critical_path_ms_per_step','projected_speedup_compute_only','parity')})
print([ (s['n_own'],s['n_ghost'],round(s['ms_per_step'],3)) for s in d['slabs']])
print('segment max', d['segment_max_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 8 2 8 > $out/slab8_traced.json 2> $out/slab8.err
