#!/bin/bash
# GPU session 2 of round 2: few-query kernel parity, whole GPU suite, bench, driver-vs-bench timing
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -x -q -k "few_query" > $OUT/pytest_few.log 2>&1; echo "few rc=$?" >> $OUT/pytest_few.log
tail -15 $OUT/pytest_few.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cat $OUT/bench_default.err | grep bench
timeout 600 python tools/cli_vs_bench.py 4000000 256 10 > $OUT/cli_vs_bench.log 2>&1; cat $OUT/cli_vs_bench.log | tail -4
timeout 300 python tools/cli_vs_bench.py 4000000 1 10 >> $OUT/cli_vs_bench.log 2>&1; tail -3 $OUT/cli_vs_bench.log
