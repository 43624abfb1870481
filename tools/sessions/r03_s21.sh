#!/bin/bash
mkdir -p gpurun_out/s21
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s21/tests.log 2>&1
tail -3 gpurun_out/s21/tests.log
timeout 900 python bench.py > gpurun_out/s21/bench.json 2> gpurun_out/s21/bench.err
tail -c 600 gpurun_out/s21/bench.err | tail -5
