#!/bin/bash
# Everything one entry of the profiles/ directory is made from, in one gpurun call:
#   gpu_profile_round.sh <tag> <counters key> <commit> [bench.py arguments ...]
# the bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, and the PMC
# passes (traffic: FETCH_SIZE and WRITE_SIZE in separate passes; SQ / cache counters).
# The counters' summary (tools/roofline_json.py -> $out/counters.json, one entry) is made on the box and
# the raw per-dispatch CSVs are dropped there: gpurun carries at most 64 MiB back.
tag=${1:-round}; key=${2:-none}; head=${3:-unknown}; shift $(( $# < 3 ? $# : 3 ))
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
state=""
case " $* " in
  *" passive_growth_grid "*) timeout 600 python tools/make_state.py 4 /tmp/state.npz > $out/state.log 2>&1; state="--state /tmp/state.npz";;
  *" branching_grid "*) timeout 600 python tools/make_state.py 3 /tmp/state.npz > $out/state.log 2>&1; state="--state /tmp/state.npz";;
esac
timeout 900 python bench.py "$@" $state > $out/bench.json 2> $out/bench.err
cat $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast-tier-line --no-sustained-line --no-tail-ab-line "$@" $state > $out/stats_bench.json 2> $out/stats.err
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_WAVES GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast-tier-line --no-sustained-line --no-tail-ab-line --preheat-ms 50 --steps 10 --warmup 2 "$@" $state > $out/pmc$i.json 2> $out/pmc$i.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$out/pmc*/*counter_collection.csv" grid_force > $out/pmc_summary.txt
python3 $GRAFT_REPO_ROOT/tools/roofline_json.py $out $out/counters.json $key "$*" $head > $out/counters.log 2>&1
rm -rf $out/pmc? $out/stats/*agent_info.csv $out/stats/*kernel_trace.csv
ls $out
