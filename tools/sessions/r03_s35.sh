#!/bin/bash
STAMP_PF=1 MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_scan.py 1000000,256,10 2>&1 | grep "^n=\|loader cycles\|insert path"
