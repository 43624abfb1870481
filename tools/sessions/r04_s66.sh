#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/r04_suite_soak.log; : > $L
for i in 1 2 3; do timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1 >> $L; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $L
