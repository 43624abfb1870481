#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04final; mkdir -p $O
timeout 1500 python bench.py > $O/r04_bench.json 2> $O/r04_bench_stderr.log
cut -c1-200 $O/r04_bench.json
