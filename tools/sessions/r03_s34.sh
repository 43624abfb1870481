#!/bin/bash
for v in ps1 ps2; do echo $v; MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so timeout 600 python tools/pf_try.py 1000000,256,10 4000000,256,10 2>&1 | grep "^n="; done
