#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel."""
import csv, sys, collections, glob
def main(paths, filt):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r['Kernel_Name']
            if filt and filt not in k: continue
            # template arguments that tell two builds of one kernel apart sit at the END of the name
            # (demangled: before the parameter list, i.e. the last top-level '('; mangled: before the 'EEv'
            # that opens the parameter list)
            head = k
            if head.startswith('_Z'):
                if 'EEv' in head: head = head[:head.rindex('EEv')]
            elif head.endswith(')'):
                depth = 0
                for pos in range(len(head) - 1, -1, -1):
                    depth += head[pos] == ')'
                    depth -= head[pos] == '('
                    if depth == 0:
                        head = head[:pos]
                        break
            key = head if len(head) <= 60 else head[:40] + ' ... ' + head[-24:]
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print('   %-28s mean %.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
if __name__ == '__main__':
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    main(sorted(glob.glob(sys.argv[1])), filt)
