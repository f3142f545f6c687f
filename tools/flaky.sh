#!/bin/bash
# run a reference test binary N times in two trees, count failures: flaky.sh <name> <N>
name=$1; N=${2:-20}
for d in _old .; do
  bad=0
  for i in $(seq 1 $N); do
    (cd $d && ./oracle/_ref/$name > /tmp/fl.log 2>&1) || true
    grep -q "ALL TESTS PASSED" /tmp/fl.log || { bad=$((bad+1)); tail -2 /tmp/fl.log | head -1; }
  done
  echo "$d: $bad / $N failed"
done
