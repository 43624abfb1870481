#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -x -q -k "large_k" 2>&1 | tail -30
