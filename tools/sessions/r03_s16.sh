#!/bin/bash
# small-nq one-call path: tests, then sweep of the thresholds
mkdir -p gpurun_out/s16
timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -x -q > gpurun_out/s16/tests.log 2>&1
echo "default (fused<=8, norm<=16)" > gpurun_out/s16/small_nq.log
timeout 300 python tools/small_nq.py >> gpurun_out/s16/small_nq.log 2>&1
echo "fused<=2 norm<=16" >> gpurun_out/s16/small_nq.log
MS_FUSED_MERGE_MAX_NQ=2 timeout 300 python tools/small_nq.py >> gpurun_out/s16/small_nq.log 2>&1
echo "fused<=0 norm<=16" >> gpurun_out/s16/small_nq.log
MS_FUSED_MERGE_MAX_NQ=0 timeout 300 python tools/small_nq.py >> gpurun_out/s16/small_nq.log 2>&1
echo "fused<=2 norm<=0" >> gpurun_out/s16/small_nq.log
MS_FUSED_MERGE_MAX_NQ=2 MS_INKERNEL_NORM_MAX_NQ=0 timeout 300 python tools/small_nq.py >> gpurun_out/s16/small_nq.log 2>&1
tail -3 gpurun_out/s16/tests.log
