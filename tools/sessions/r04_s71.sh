#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s71.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_prefilter_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $L
for rep in 1 2; do
for v in main bv512; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  timeout 300 python tools/sample_sweep.py 1000000,256,10 1000000,1024,10 250000,256,10 2>&1 | grep "^n=" >> $L
  for s in "1000000 32" "4000000 32" "1000000 8"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
done; done
