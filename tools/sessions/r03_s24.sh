#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s24
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s24/p -o t -- python3 $R/tools/ksweep.py 64 > $R/gpurun_out/s24/run.log 2>&1
f=$(find $R/gpurun_out/s24/p -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/s24/stats_k64.csv
find $R/gpurun_out/s24/p -type f ! -name '*stats.csv' -delete
