#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu > gpurun_out/s1_search.log 2>&1; echo "search rc=$?"; tail -5 gpurun_out/s1_search.log
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench rc=$?"; cat gpurun_out/s1_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['recall_at_k'], d['topk_identical_to_torch_bruteforce'])"
