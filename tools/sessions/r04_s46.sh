#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s46.log; : > $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_body.py 1000000,32 4000000,32 1000000,1 >> $L 2>&1
echo "== inkernel norm up to 64" >> $L
for s in "1000000 32" "4000000 32" "1000000 8"; do MS_INKERNEL_NORM_MAX_NQ=64 timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
echo "== default" >> $L
for s in "1000000 32" "4000000 32" "1000000 8" "1000000 64"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
echo "== PF sample coef" >> $L
for C in 0.3 0.6 1.2; do echo "coef $C" >> $L; MS_PF_SAMPLE_COEF=$C timeout 300 python tools/sample_sweep.py 1000000,256,10 250000,256,10 4000000,256,10 2>&1 | grep "^n=" >> $L; done
