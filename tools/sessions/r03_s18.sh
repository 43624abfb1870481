#!/bin/bash
for n in 1000000 4000000; do for nq in 1 8 32; do
python tools/hbm_shape.py $n $nq 2>&1 | grep rows=
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/noins/libmerizo_search_amd.so python tools/hbm_shape.py $n $nq 2>&1 | grep rows= | sed 's/^/   no-insert: /'
done; done
