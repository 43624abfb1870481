#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s68.log; : > $L
for i in 1 2; do timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1 >> $L; done
