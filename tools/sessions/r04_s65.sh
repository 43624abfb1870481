#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s65.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $L
timeout 900 python tools/stress_search.py 8 300 2>&1 | tail -1 >> $L
for rep in 1 2 3; do
for v in main fence; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  for s in "500000 1" "1000000 1" "1000000 2" "4000000 1"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
done; done
