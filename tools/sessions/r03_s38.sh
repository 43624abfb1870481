#!/bin/bash
for v in 8 16 33; do echo "MS_SAMPLE_MIN_NQ=$v"; MS_SAMPLE_MIN_NQ=$v timeout 300 python tools/small_nq.py 2>&1 | grep "rows=" | grep "nq=8\|nq=32\|nq=64"; done
echo "PF sample size:"; for t in 4 6 9 14; do echo "T0=$t"; MS_PREPASS_TILES=$t timeout 300 python tools/pf_try.py 1000000,256,10 2>&1 | grep "^n="; done
