#!/bin/bash
# Final checks of round 2: whole GPU suite, smoke, default bench line, 2-rank self-test of the bench (gloo, both ranks on cuda:0)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep bench $OUT/bench_default.err
MS_BENCH_SAME_DEVICE=1 MS_BENCH_BACKEND=gloo MS_BENCH_ROWS_PER_GPU=4000000 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 2> $OUT/bench_2rank_selftest.err | grep '^{' > $OUT/bench_2rank_selftest.json; echo "bench2 rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_2rank_selftest.json')); print(d['config']['workload'], d['value'], d['recall_at_k'], d['planted_recall'], d['scaling'])"
