#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s42.log; : > $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_scan.py 1000000,256,64 >> $L 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_k64 -o kt -- python3 $GRAFT_REPO_ROOT/tools/ksweep.py 64 > /dev/null 2>&1
f=$(find /tmp/kt_k64 -name '*kernel_stats.csv' | head -1); cut -d, -f1-4 $f | grep ms_ | cut -c1-150 >> $L
