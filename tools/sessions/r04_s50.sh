#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s50.log; : > $L
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/base/libmerizo_search_amd.so; else unset MS_LIB_OVERRIDE; fi
  echo "== $v" >> $L
  for s in "1000000 32" "4000000 32" "1000000 8"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
  timeout 300 python tools/sample_sweep.py 1000000,256,10 2>&1 | grep "^n=" >> $L
done; done
