#!/bin/bash
timeout 900 python tools/pf_repro2.py 300 2>&1 | grep -v amdgpu.ids | grep "k=48"
timeout 600 python tools/ksweep.py 10 64 2>&1 | grep "^k="
timeout 1200 python -m pytest tests/test_search_gpu.py tests/test_prefilter_gpu.py -m gpu -x -q 2>&1 | tail -2
