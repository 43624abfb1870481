#!/bin/bash
mkdir -p gpurun_out/stress
for seed in 41 42 43; do timeout 1200 python tools/stress_prefilter.py $seed 1000 2>&1 | grep -v amdgpu.ids | tail -4; done > gpurun_out/stress/r03_stress_prefilter.log
cat gpurun_out/stress/r03_stress_prefilter.log
