#!/bin/bash
# HBM-side traffic of the force kernel: FETCH_SIZE and WRITE_SIZE in separate passes, at 1 M and 10 M cells
out=$GRAFT_REPO_ROOT/gpurun_out/r2traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 1000000 10000000; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${c}_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --cells-total $n > $out/${c}_$n.json 2> $out/${c}_$n.err
  done
done
python3 - <<PY
import csv, glob, collections
for n in (1000000, 10000000):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for p in glob.glob("$out/%s_%d/*counter_collection.csv" % (c, n)):
            for r in csv.DictReader(open(p)):
                agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
            print(n, c, k, "mean KiB %.0f" % (sum(v) / len(v)), "n", len(v))
PY
