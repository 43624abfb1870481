#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s28
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p28 -o t -- python3 $R/tools/pf_loop.py 1000000 256 10 > $R/gpurun_out/s28/run.log 2>&1
f=$(find /tmp/p28 -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/s28/stats.csv
tail -1 $R/gpurun_out/s28/run.log
