#!/bin/bash
timeout 300 python tools/small_nq.py 2>&1 | grep rows=
timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -x -q 2>&1 | tail -2
