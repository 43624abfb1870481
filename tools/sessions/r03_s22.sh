#!/bin/bash
mkdir -p gpurun_out/s22
timeout 1200 python -m pytest tests/test_search_gpu.py -m gpu -x -q > gpurun_out/s22/tests.log 2>&1
tail -3 gpurun_out/s22/tests.log
timeout 600 python tools/ksweep.py 2>&1 | grep -v amdgpu.ids | tail -12
