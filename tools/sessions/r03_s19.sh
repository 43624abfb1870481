#!/bin/bash
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so python tools/stamp_body.py 1000000,1 1000000,32 2>&1 | grep -v amdgpu.ids
