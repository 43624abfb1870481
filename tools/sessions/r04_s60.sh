#!/bin/bash
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gpurun_out/s60.log; : > $L
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c5 -o kt -- python3 $GRAFT_REPO_ROOT/tools/c5_latency.py > /dev/null 2>&1
f=$(find /tmp/kt_c5 -name '*kernel_stats.csv' | head -1); cut -d, -f1-4 $f | grep -i "egnn\|Name" | cut -c1-170 >> $L
