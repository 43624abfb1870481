#!/bin/bash
# GPU session 1 of round 2: full GPU test-suite, default bench line, 2-rank self-test of the bench, kernel trace.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -3 $OUT/bench_default.err
MS_BENCH_SAME_DEVICE=1 MS_BENCH_BACKEND=gloo MS_BENCH_ROWS_PER_GPU=2000000 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2rank_selftest.json 2> $OUT/bench_2rank_selftest.err; echo "bench2 rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_default -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/kt_default.log 2>&1; echo "rocprof rc=$?"
ls $OUT/kt_default | head
