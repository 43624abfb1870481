#!/bin/bash
# Round 4, final build: the whole GPU suite, the randomised sweeps (fp32 search vs the oracle, prefiltered search vs the fp32 scan)
# and the bench line; logs into gpurun_out/r04final (copied to profiles/ afterwards).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04final; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/r04_pytest_gpu.log
timeout 1500 python tools/stress_search.py 4 300 > $O/r04_stress_search_300.log 2>&1
timeout 2400 python tools/stress_prefilter.py 4 3000 > $O/r04_stress_prefilter_3000.log 2>&1
timeout 1500 python bench.py > $O/r04_bench.json 2> $O/r04_bench_stderr.log
tail -3 $O/r04_pytest_gpu.log; tail -2 $O/r04_stress_search_300.log; tail -2 $O/r04_stress_prefilter_3000.log; cut -c1-300 $O/r04_bench.json
