#!/bin/bash
# The tail of half tiles against launches of whole tiles only (bench.py --tail-tiles 0), same box, interleaved:
# both arithmetic tiers at 1 M cells, the exact tier at 10 M and 6e5 cells, with the sustained figure.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_tail_ab; mkdir -p $out
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for cfg in ${TAIL_AB_CASES:-"exact:1000000:0 exact:1000000:-1 fast:1000000:0 fast:1000000:-1 exact:600000:0 exact:600000:-1 exact:10000000:0 exact:10000000:-1"}; do
    IFS=: read tier cells tail <<< "$cfg"
    set -- $tier $cells
    {
      timeout 400 python bench.py --arith $1 --cells-total $2 --no-cpu-baseline --no-fast-tier-line --sustained-steps 1500 \
        --tail-tiles $tail > $out/b.json 2> $out/b.err || { echo "bench failed"; tail -3 $out/b.err; continue; }
      python3 -c "import json; d=json.load(open('$out/b.json')); s=d.get('sustained'); print('$1 $2 tail $tail', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], 'force %.1f us'%d['roofline']['avg_launch_us'], ('sustained %.4g, force %.1f us'%(s['value'], s['force_us'])) if s else '')" | tee -a $out/lines.txt
    }
  done
done
