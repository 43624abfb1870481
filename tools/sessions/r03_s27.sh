#!/bin/bash
for c in 1000000,256,1 300000,256,1 1000000,256,5 4000000,512,3 100000,200,10 2000000,300,16; do echo "== $c"; timeout 600 python tools/pf_debug.py $c 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-120 | grep "garbage\|fell"; done
timeout 600 python tools/pf_try.py 2>&1 | grep "^n="
