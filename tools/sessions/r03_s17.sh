#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s17
for shape in "1000000 1" "4000000 1" "1000000 32" "16000000 1"; do
  set -- $shape
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s17/p_$1_$2 -o t -- python3 $R/tools/hbm_shape.py $1 $2 > $R/gpurun_out/s17/run_$1_$2.log 2>&1
  f=$(find $R/gpurun_out/s17/p_$1_$2 -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/s17/stats_$1_$2.csv
  find $R/gpurun_out/s17/p_$1_$2 -type f ! -name '*stats.csv' -delete
done
