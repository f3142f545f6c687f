#!/bin/bash
# The round's profiles: one gpu_profile_round per workload (bench line + kernel stats + PMC passes).
#   gpu_r06_final.sh <commit> [which: all | springs | configs]
cd $GRAFT_REPO_ROOT
which=${2:-all}
if [ $which = all ] || [ $which = springs ]; then
bash tools/gpu_profile_round.sh r06_springs_1M springs_1M $1 2>&1 | tail -2
bash tools/gpu_profile_round.sh r06_springs_1M_fast springs_1M_fast $1 --arith fast 2>&1 | tail -2
bash tools/gpu_profile_round.sh r06_springs_10M springs_10M $1 --cells-total 10000000 --cpu-steps 1 2>&1 | tail -2
fi
if [ $which = all ] || [ $which = configs ]; then
bash tools/gpu_profile_round.sh r06_cfg4 cfg4_passive_growth_1M $1 --model passive_growth_grid --cpu-steps 3 2>&1 | tail -2
bash tools/gpu_profile_round.sh r06_cfg4_renumbered cfg4_passive_growth_1M_renumbered $1 --model passive_growth_grid --renumber-every 10 --cpu-steps 3 2>&1 | tail -2
bash tools/gpu_profile_round.sh r06_cfg3 cfg3_branching_100k $1 --model branching_grid --cpu-steps 20 2>&1 | tail -2
bash tools/gpu_profile_round.sh r06_cfg2 cfg2_sorting_10k $1 --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --cpu-steps 300 2>&1 | tail -2
fi
du -sh gpurun_out/r06_*
