#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/r04_suite_soak3.log; : > $L
for i in 1 2 3 4 5 6 7 8; do
  timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=long --durations=3 > /tmp/run_$i.log 2>&1
  tail -1 /tmp/run_$i.log >> $L
  if ! tail -1 /tmp/run_$i.log | grep -q "214 passed"; then echo "=== FAILURE in run $i" >> $L; grep -v "^$" /tmp/run_$i.log | tail -150 | cut -c1-300 >> $L; break; fi
done
