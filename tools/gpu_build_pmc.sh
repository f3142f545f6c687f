#!/bin/bash
# PMC passes over the grid build's kernels of the headline run (which gathers miss):
# gpu_build_pmc.sh <tag>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast-tier-line --no-sustained-line --steps 10 --warmup 2 > $out/pmc$i.json 2> $out/pmc$i.err
done
for k in k_order k_bin k_scatter k_scan k_reduce euler_step heun_step; do
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$out/pmc*/*counter_collection.csv" $k
done > $out/build_pmc.txt
rm -rf $out/pmc?
cat $out/build_pmc.txt
