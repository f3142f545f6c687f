#!/usr/bin/env python3
"""Print the kernel timeline of one mid-run step from a rocprofv3 kernel trace csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(k_\w+|grid_force\w*|tile_force|heun_step\w*|euler_step\w*|copyBuffer|fillBuffer|make_fix|link\w*)', n)
    return m.group(1) if m else n[:24]
seq = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
last = sys.argv[2] if len(sys.argv) > 2 else 'heun_step_raw'
idx = [i for i, s in enumerate(seq) if s[0] == last]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = seq[a][2]; prev = seq[a][1]
for nm, s, e in seq[a:b + 1]:
    print(f"{nm:20s} start {(s - t0) / 1000:8.1f}  gap {(s - prev) / 1000:6.1f}  dur {(e - s) / 1000:7.1f}")
    prev = e
print("step", (seq[b][2] - seq[a][2]) / 1000)
