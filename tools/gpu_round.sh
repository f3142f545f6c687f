#!/bin/bash
# Run on the GPU box via gpurun: gpu tests, A/B bench of the two force kernels,
# and a rocprofv3 kernel-trace summary of the default bench.  Outputs under
# gpurun_out/<tag>/.
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log
for v in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-variant $v > $out/bench_v$v.json 2> $out/bench_v$v.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err
ls -R $out/prof | head -20
