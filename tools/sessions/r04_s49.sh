#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s49.log; : > $L
timeout 1200 python -m pytest tests/test_search_gpu.py tests/test_prefilter_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $L
for s in "1000000 32" "4000000 32" "1000000 8" "1000000 64"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
timeout 300 python tools/sample_sweep.py 1000000,256,10 1000000,256,64 2>&1 | grep "^n=" >> $L
