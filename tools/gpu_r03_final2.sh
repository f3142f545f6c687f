#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile_round.sh r03_cfg3 cfg3_branching_100k $1 --model branching_grid --cpu-steps 20 2>&1 | tail -2
bash tools/gpu_profile_round.sh r03_cfg2 cfg2_sorting_10k $1 --model sorting_grid --cells-total 10000 --dt 0.05 --steps 300 --cpu-steps 300 2>&1 | tail -2
