#!/usr/bin/env python3
"""Counters of the force launches per slab from a rocprofv3 --pmc run of tools/slab_rehearsal with
YALLA_REHEARSAL_MARKERS=1: slab_pmc_by_rank.py <counter_collection.csv>.  Rows are dispatches in order;
a marker kernel slab_takes_the_gpu<r> says whose launches follow."""
import collections, csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
rank, seen_marker = None, False
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"slab_takes_the_gpu<(\d+)>", name) or re.search(r"slab_takes_the_gpuILi(\d+)E", name)
    if m:
        rank, seen_marker = int(m.group(1)), True
        continue
    if not seen_marker or "grid_force" not in name:
        continue
    a = agg[rank][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
out = {str(k): {c: round(v[0] / max(v[1], 1), 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())}
       for k, cs in sorted(agg.items())}
print(json.dumps(out, indent=1))
