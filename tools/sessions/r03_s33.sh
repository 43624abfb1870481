#!/bin/bash
timeout 1200 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/pf_try.py 1000000,256,10 1000000,256,20 1000000,256,32 4000000,256,10 2>&1 | grep "^n="
