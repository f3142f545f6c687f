#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_final3; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 8 4 2 1; do
  timeout 600 tools/slab_rehearsal 10000000 $w 16 3 8 > $out/rehearsal_10M_w$w.json 2> $out/rehearsal_10M_w$w.err; echo "w=$w rc=$?"
done
( cd /tmp && export TMPDIR=/tmp YALLA_REHEARSAL_MARKERS=1 && rocprofv3 --kernel-trace --stats --output-format csv -d $out/slab8 -o k -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 16 0 8 > $out/slab8_traced.json 2> $out/slab8.err )
python3 tools/slab_trace_summary.py $out/slab8/k_kernel_trace.csv 16 > $out/slab8_device_time.json
rm -f $out/slab8/k_kernel_trace.csv $out/slab8/k_agent_info.csv
for n in 10000 30000 50000 100000 300000; do for a in exact fast; do
  python bench.py --no-cpu-baseline --arith $a --cells-total $n > $out/small_${n}_$a.json 2>/dev/null
done; done
for n in 500 800 2000; do
  python bench.py --model springs_tile --cells-total $n --steps 100 --cpu-steps 100 > $out/tile_$n.json 2>/dev/null
done
python bench.py --model springs_links_grid --cells-total 100000 --no-cpu-baseline > $out/links_100k.json 2>/dev/null
python bench.py --model springs_links_grid --cells-total 1000000 --no-cpu-baseline > $out/links_1M.json 2>/dev/null
python bench.py > $out/bench.json 2> $out/bench.err
for f in $out/small_*.json $out/tile_*.json $out/links_*.json $out/bench.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'], d['roofline']['kernel'][:70], d.get('cpu_baseline',{}).get('value'))"; done
