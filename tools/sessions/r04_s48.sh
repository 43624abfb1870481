#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s48.log; : > $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_body.py 1000000,32 2>&1 | head -12 >> $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
for s in "1000000 32" "2000000 32" "4000000 32" "8000000 32" "1000000 8" "1000000 64" "4000000 64" "16000000 32" "500000 32"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L;  MS_SELF_SAMPLE=0 timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= | sed 's/^/   off: /' >> $L; done
