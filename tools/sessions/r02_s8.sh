#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep bench $OUT/bench_default.err
timeout 600 python tools/cli_vs_bench.py 4000000 256 10 > $OUT/cli_vs_bench.log 2>&1; tail -3 $OUT/cli_vs_bench.log
python -c "
import sys; sys.path.insert(0,'.')
from merizo_search_amd.foldclass.engine import HipEngine
HipEngine.COPY_THREADS=1
exec(open('tools/cli_vs_bench.py').read())
" 4000000 256 10 2>&1 | tail -3 | sed 's/^/1-thread copy: /' | tee -a $OUT/cli_vs_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
