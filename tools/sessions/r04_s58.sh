#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/r04_stress_more_seeds.log; : > $L
for sd in 5 6 7; do timeout 900 python tools/stress_search.py $sd 300 2>&1 | tail -3 >> $L; done
for sd in 5 6; do timeout 1500 python tools/stress_prefilter.py $sd 3000 2>&1 | tail -2 >> $L; done
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_prefilter_gpu.py -q -m gpu 2>&1 | tail -2 >> $L
