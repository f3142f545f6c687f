#!/usr/bin/env python3
"""profiles/r01_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes."""
import csv, glob, json, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(p)):
        if 'grid_force' in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
out = {"grid_force_1M_springs": {"FETCH_SIZE_KiB": sum(agg['FETCH_SIZE'])/len(agg['FETCH_SIZE']),
                                 "WRITE_SIZE_KiB": sum(agg['WRITE_SIZE'])/len(agg['WRITE_SIZE']),
                                 "launches": len(agg['FETCH_SIZE']),
                                 "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline (tools/gpu_profile_round.sh)"}}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(out)
