#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/s41.log; : > $L
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu  2>&1 | tail -5 >> $L
for T in 5 7 9 12 16 22; do echo "== MS_PREPASS_TILES=$T" >> $L; MS_PREPASS_TILES=$T timeout 200 python tools/ksweep.py 64 2>&1 | grep "^k=" >> $L; done
echo "== default" >> $L; timeout 200 python tools/ksweep.py 10 32 48 64 2>&1 | grep "^k=" >> $L
