#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_egnn_gpu.py tests/test_stress_gpu.py tests/test_drivers_gpu.py -m gpu -x -q > $OUT/pytest_egnn.log 2>&1; echo "egnn rc=$?" >> $OUT/pytest_egnn.log
tail -5 $OUT/pytest_egnn.log
python tools/diag_egnn.py 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_egnn.log
