#!/bin/bash
# why are the last slab's force launches dearer than the first slab's?  counters per slab
out=$GRAFT_REPO_ROOT/gpurun_out/r04_slab_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export YALLA_REHEARSAL_MARKERS=1 YALLA_SLAB_PLAN=quantile
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/p$i -o p -- $GRAFT_REPO_ROOT/tools/slab_rehearsal 10000000 8 10 2 8 > $out/p$i.json 2> $out/p$i.err
  python3 $GRAFT_REPO_ROOT/tools/slab_pmc_by_rank.py $out/p$i/p_counter_collection.csv > $out/by_rank_$i.json 2> $out/by_rank_$i.err
  rm -rf $out/p$i
done
cat $out/by_rank_*.json | head -150
