#!/bin/bash
timeout 300 python tools/ksweep.py 10 32 33 48 64 2>&1 | grep "^k="
timeout 600 python -m pytest tests/test_search_gpu.py -m gpu -x -q 2>&1 | tail -2
