#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s59.log; : > $L
timeout 900 python -m pytest tests/test_egnn_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
for rep in 1 2; do
for v in main waitc; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  echo "== $v" >> $L
  timeout 300 python tools/c5_latency.py 2>&1 | grep -v amdgpu.ids >> $L
done; done
