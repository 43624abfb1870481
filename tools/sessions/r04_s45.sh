#!/bin/bash
# sample-size rule with the shared bound: sweep of its constant over a few shapes (fp32 and prefiltered)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/r04_sample_size_sweep.log; : > $L
timeout 600 python -m pytest tests/test_search_gpu.py tests/test_prefilter_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $L
S="1000000,256,10 1000000,256,64 4000000,256,10 1000000,1024,10 250000,256,10 16000000,512,10 1000000,256,32"
for C in 0.01 0.02 0.035 0.07 0.15 0.3; do echo "== MS_SAMPLE_COEF=$C" >> $L; MS_SAMPLE_COEF=$C timeout 400 python tools/sample_sweep.py $S 2>&1 | grep "^n=" >> $L; done
