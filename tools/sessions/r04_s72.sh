#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s72.log; : > $L
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $L
timeout 1700 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1 >> $L
timeout 600 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-330 >> $L
