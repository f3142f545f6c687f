#!/bin/bash
# quick lines of configs 4, 3, 2 (no profiler)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_cfgq; mkdir -p $out
python tools/make_state.py 4 /tmp/s4.npz > /dev/null 2>&1
python tools/make_state.py 3 /tmp/s3.npz > /dev/null 2>&1
for a in exact fast; do
python bench.py --no-cpu-baseline --arith $a --model passive_growth_grid --state /tmp/s4.npz > $out/cfg4_$a.json 2>$out/cfg4_$a.err
python bench.py --no-cpu-baseline --arith $a --model branching_grid --state /tmp/s3.npz > $out/cfg3_$a.json 2>$out/cfg3_$a.err
python bench.py --no-cpu-baseline --arith $a --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 > $out/cfg2_$a.json 2>$out/cfg2_$a.err
python bench.py --no-cpu-baseline --arith $a --cells-total 100000 > $out/s100k_$a.json 2>$out/s100k_$a.err
done
for f in $out/*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'], d['roofline']['kernel'][:60])"; done
timeout 300 tests/native/test_stateless
timeout 300 tests/native/test_links_big_ids
