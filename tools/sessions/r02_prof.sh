#!/bin/bash
# Profiles of round 2: kernel traces + PMC passes for the bench headline (C2) and the extra entries
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
cd /tmp
kt() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -o kt -- python3 "$@" > $OUT/kt_$name.log 2>&1; echo "kt $name rc=$?"; }
pmc() { name=$1; ctr=$2; shift; shift; timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${name} -o pmc -- python3 "$@" > $OUT/pmc_${name}.log 2>&1; echo "pmc $name rc=$?"; }
# C2 headline: the default bench step
kt c2 $R/bench.py --no-extras --no-cpu-baseline
pmc c2_fetch FETCH_SIZE $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
pmc c2_write WRITE_SIZE $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
pmc c2_busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
# HBM-bound regime
for shape in "1000000 1" "1000000 32" "4000000 1" "4000000 32" "45625000 1" "45625000 32"; do set -- $shape
  kt hbm_$1_$2 $R/tools/prof_scan.py $1 $2 10 12
  pmc hbm_$1_$2_fetch FETCH_SIZE $R/tools/prof_scan.py $1 $2 10 4
done
# C4 per-GPU shard
kt c4 $R/tools/prof_scan.py 45625000 4096 10 2
pmc c4_busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" $R/tools/prof_scan.py 45625000 4096 10 1
pmc c4_fetch FETCH_SIZE $R/tools/prof_scan.py 45625000 4096 10 1
# encoder
kt egnn $R/tools/prof_egnn.py 1000 3
pmc egnn_busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" $R/tools/prof_egnn.py 1000 2
ls $OUT | head -80
du -sh $OUT
