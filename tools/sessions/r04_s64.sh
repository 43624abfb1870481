#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s64.log; : > $L
timeout 1500 python -m pytest tests/test_sharded_drivers.py -q -m gpu 2>&1 | tail -4 >> $L
