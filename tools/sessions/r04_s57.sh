#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s57.log; : > $L
timeout 900 python -m pytest tests/test_prefilter_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
S="1000000,256,10 4000000,256,10 1000000,1024,10 16000000,1024,10 1000000,160,10 1000000,128,10"
for rep in 1 2 3; do
for v in main waitc; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  timeout 300 python tools/pf2_try.py $S 2>&1 | grep "^n=" | cut -c1-110 >> $L
done; done
