#!/bin/bash
mkdir -p gpurun_out/s32
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/s32/bench.json 2> gpurun_out/s32/bench.err; echo "rc=$?"
grep "^\[bench\]" gpurun_out/s32/bench.err | tail -30
