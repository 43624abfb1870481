#!/bin/bash
# Randomised parity sweep of the final build of round 3 (indices and score bits against the oracle; cosine modes near-tie aware)
mkdir -p gpurun_out/stress
for seed in 31 32 33 34; do timeout 1500 python tools/stress_search.py $seed 300 2>&1 | grep -v amdgpu.ids | tail -4; done > gpurun_out/stress/r03_stress_sweep.log
timeout 600 python tools/extreme_shapes.py 2>&1 | grep -v amdgpu.ids | tail -12 >> gpurun_out/stress/r03_stress_sweep.log
cat gpurun_out/stress/r03_stress_sweep.log
