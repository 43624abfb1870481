#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s51.log; : > $L
timeout 1200 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/base/libmerizo_search_amd.so; else unset MS_LIB_OVERRIDE; fi
  echo "== $v" >> $L
  for s in "1000000 32" "4000000 32" "1000000 8" "1000000 1" "1000000 4" "1000000 64"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
done; done
