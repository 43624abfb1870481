#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s62.log; : > $L
S="1000000,256,10 4000000,256,10 16000000,256,10 1000000,1024,10 16000000,1024,10 1000000,128,10"
for rep in 1 2; do
for v in main pfnt; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  timeout 400 python tools/pf2_try.py $S 2>&1 | grep "^n=" | cut -c1-110 >> $L
  timeout 300 python tools/pf_loop.py 45625000 4096 10 3 2>&1 | tail -1 >> $L
done; done
