#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s43.log; : > $L
STAMP_FLUSH=1 MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stampf/libmerizo_search_amd.so timeout 300 python tools/stamp_scan.py 1000000,256,64 >> $L 2>&1
