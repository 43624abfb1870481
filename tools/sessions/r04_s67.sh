#!/bin/bash
# repro loop for the one-in-many hang of `bench.py --gpus 8` with eight ranks on one GPU (tests/test_sharded_drivers.py)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/r04_bench8_repro.log; : > $L
export MS_BENCH_SAME_DEVICE=1 MS_BENCH_BACKEND=gloo MS_BENCH_FAULT_DUMP=100
for i in $(seq 1 60); do
  t0=$(date +%s)
  timeout 170 python bench.py --gpus 8 --rows 800000 --nq 96 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/b8.out 2> /tmp/b8.err; rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 )) s" >> $L
  if [ $rc -ne 0 ]; then echo "=== stderr of run $i" >> $L; grep -v "^$" /tmp/b8.err | tail -200 | cut -c1-260 >> $L; break; fi
done
