#!/bin/bash
mkdir -p gpurun_out/s20
timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -x -q > gpurun_out/s20/tests.log 2>&1
tail -3 gpurun_out/s20/tests.log
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so python tools/stamp_body.py 1000000,1 1000000,32 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/small_nq.py 2>&1 | grep rows=
