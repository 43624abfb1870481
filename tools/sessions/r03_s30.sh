#!/bin/bash
timeout 1200 python -m pytest tests/test_prefilter_gpu.py -m gpu -x -q 2>&1 | tail -15
