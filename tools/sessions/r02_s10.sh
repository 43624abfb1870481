#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_stress_gpu.py -m gpu -x -q > $OUT/pytest_search.log 2>&1; echo "search rc=$?" >> $OUT/pytest_search.log
tail -4 $OUT/pytest_search.log
timeout 600 python tools/stress_search.py > $OUT/stress.log 2>&1; tail -2 $OUT/stress.log
for k in 10 20 25 32 64; do python tools/diag_search.py 1000000,256,$k 2>/dev/null | tail -1; done | tee $OUT/diag_k.log
DIAG_COSINE=1 python tools/diag_search.py 1000000,256,25 500000,1000,10 2>/dev/null | tail -2 | tee -a $OUT/diag_k.log
