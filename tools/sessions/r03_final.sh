#!/bin/bash
# Final checks of round 3: the GPU test suite, smoke(), the default bench line
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest_gpu.log
tail -3 gpurun_out/final/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 1500 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
tail -c 300 gpurun_out/final/bench.json
