#!/bin/bash
# Round 4: tiles started dearest first inside every XCD (Grid_computer::tile_order_every) against storage order:
# the headline (1 M cells), 3e5 and 10 M cells, same box, interleaved; parity of the two orders; then the
# slab rehearsal's per-stage force spans for the end slabs, and the springs drift table.
out=$GRAFT_REPO_ROOT/gpurun_out/r04_lpt; mkdir -p $out
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for k in 0 1 8; do
    timeout 300 python bench.py --no-cpu-baseline --tile-order-every $k > $out/b_1M_k${k}_$rep.json 2> $out/b.err
    python3 -c "import json; d=json.load(open('$out/b_1M_k${k}_$rep.json')); print('1M order-every $k rep $rep', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
for cells in 300000 10000000; do
  for k in 0 8 0 8; do
    timeout 300 python bench.py --no-cpu-baseline --cells-total $cells --tile-order-every $k > $out/b_${cells}_k$k.json 2> $out/b.err
    python3 -c "import json; d=json.load(open('$out/b_${cells}_k$k.json')); print('$cells order-every $k', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
  done
done
for k in 0 8; do
  timeout 300 python bench.py --no-cpu-baseline --arith fast --tile-order-every $k > $out/b_1M_fast_k$k.json 2> $out/b.err
  python3 -c "import json; d=json.load(open('$out/b_1M_fast_k$k.json')); print('1M fast order-every $k', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'])"
done
timeout 300 python - <<'PY'
import numpy as np
from yalla_amd.solution import Solution
runs = []
for k in (0, 1, 3):
    with Solution("springs_grid", 400000, 50, 1.0) as s:
        s.random_sphere(0.5, 7)
        s.set_param("tile_order_every", k)
        s.take_step(0.001, 6)
        runs.append(s.positions())
print("orders bit-identical:", all(np.array_equal(runs[0].view(np.uint32), r.view(np.uint32)) for r in runs[1:]))
PY
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1
for plan in planes quantile; do
  if [ $plan = quantile ]; then export YALLA_SLAB_PLAN=quantile; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/slab8_$plan -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 24 3 8 > $out/slab8_traced_$plan.json 2> $out/slab8_$plan.err
  SLAB_TIMELINE_RANK=0,7 python3 $GRAFT_REPO_ROOT/tools/slab_trace_summary.py $out/slab8_$plan/k_kernel_trace.csv 27 > $out/slab8_device_time_$plan.json 2> $out/timelines_$plan.txt
  rm -rf $out/slab8_$plan
done
unset YALLA_SLAB_PLAN YALLA_REHEARSAL_MARKERS
cd $GRAFT_REPO_ROOT
timeout 900 python tools/diag/springs_drift.py > $out/springs_drift.json 2> $out/springs_drift.err; echo "drift rc=$?"
