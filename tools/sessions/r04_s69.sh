#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s69.log; : > $L
timeout 900 python -m pytest tests/test_prefilter_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $L
for rep in 1 2; do
for v in main fence; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  timeout 300 python tools/sample_sweep.py 1000000,256,10 1000000,1024,10 4000000,256,32 2>&1 | grep "^n=" >> $L
  timeout 300 python tools/prof_c3.py 40 prefiltered 2>&1 | tail -1 >> $L
done; done
