#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s61.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $L
for rep in 1 2; do
for v in main nont; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  for s in "1000000 1" "1000000 32" "4000000 1" "4000000 32" "16000000 1" "45625000 1 60" "45625000 32 60"; do timeout 200 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
done; done
