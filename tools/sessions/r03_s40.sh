#!/bin/bash
timeout 900 python tools/pf_repro2.py 400 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/ksweep.py 10 32 64 2>&1 | grep "^k=\|c3 search"
timeout 600 python tools/pf_try.py 1000000,256,10 1000000,256,32 2>&1 | grep "^n="
