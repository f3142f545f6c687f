#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 1 16 64; do for n in 800 3000; do
  python bench.py --no-cpu-baseline --model springs_tile --cells-total $n --tile-lanes $l --steps 100 --time-every 1 2>/dev/null > /tmp/t.json
  python3 -c "import json; d=json.load(open('/tmp/t.json')); print('lanes $l n $n', '%.4g c-u/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'force %.1f us' % d['roofline']['avg_launch_us'])"
done; done
