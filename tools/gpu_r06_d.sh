#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06d_suite.txt 2>&1; tail -4 gpurun_out/r06d_suite.txt
for k in 1 2 3; do python bench.py --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --no-cpu-baseline > gpurun_out/r06d_cfg2_$k.json 2>> gpurun_out/r06d_bench.err; done
python bench.py --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --no-cpu-baseline --sorted-pipeline 3 > gpurun_out/r06d_cfg2_noinline.json 2>> gpurun_out/r06d_bench.err
python bench.py --cells-total 10000 --steps 300 --no-cpu-baseline > gpurun_out/r06d_springs10k.json 2>> gpurun_out/r06d_bench.err
for k in 1 2; do python bench.py --no-sustained-line --no-fast-tier-line --no-cpu-baseline --no-tail-ab-line > gpurun_out/r06d_bench_$k.json 2>> gpurun_out/r06d_bench.err; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06d_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --model sorting_grid --cells-total 10000 --dt 0.05 --steps 60 --no-cpu-baseline --preheat-ms 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06d_trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 400 kernels: durations and gaps
tail=rows[-400:]
out=open('gpurun_out/r06d_cfg2_timeline.txt','w')
prev=None
for r in tail[-60:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write('%-50s dur %6.2f us gap %6.2f us\n'%(r['Kernel_Name'][:50],(e-s)/1e3,(s-prev)/1e3 if prev else 0)); prev=e
import statistics
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(tail[:-1],tail[1:])]
durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in tail]
out.write('median gap %.2f mean gap %.2f  mean dur %.2f  sum(dur)/span %.3f\n'%(statistics.median(gaps),sum(gaps)/len(gaps),sum(durs)/len(durs),sum(durs)/((int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp']))/1e3)))
out.close()
print(open('gpurun_out/r06d_cfg2_timeline.txt').read()[-2500:])
PY
rm -rf gpurun_out/r06d_trace
