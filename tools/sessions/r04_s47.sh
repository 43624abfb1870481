#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s47.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $L
echo "== self-sample (default)" >> $L
for s in "1000000 32" "4000000 32" "1000000 8" "1000000 64" "16000000 32" "250000 32"; do timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
echo "== MS_SELF_SAMPLE=0" >> $L
for s in "1000000 32" "4000000 32" "1000000 8" "1000000 64" "16000000 32" "250000 32"; do MS_SELF_SAMPLE=0 timeout 100 python tools/hbm_shape.py $s 2>&1 | grep rows= >> $L; done
