#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s44.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_scan.py 1000000,256,64 >> $L 2>&1
for T in 5 9 14; do echo "== MS_PREPASS_TILES=$T" >> $L; MS_PREPASS_TILES=$T timeout 200 python tools/ksweep.py 64 2>&1 | grep "^k=" >> $L; done
echo "== default" >> $L; timeout 200 python tools/ksweep.py 10 48 64 2>&1 | grep "^k=" >> $L
