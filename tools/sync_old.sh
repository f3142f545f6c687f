#!/bin/bash
# copy the working tree's sources into the scratch worktree _old/ and build it there
# with extra compiler flags: sync_old.sh "-DYA_SOME_KNOB=1"  (same-box A/B via tools/ab_old.sh)
cd /root/repo
[ -d _old ] || git worktree add -f _old HEAD >/dev/null 2>&1   # scratch tree (git-ignored)
for f in include/*.cuh include/*.h bench.py yalla_amd/*.py yalla_amd/csrc/*.hip yalla_amd/csrc/*.inc yalla_amd/csrc/*.h yalla_amd/csrc/Makefile; do cp $f _old/$f; done
touch _old/yalla_amd/csrc/models.hip
make -C _old/yalla_amd/csrc EXTRA="$1" 2>&1 | grep -E "error|warning: unused" -A3 | head -20
ls -la _old/yalla_amd/*.so
