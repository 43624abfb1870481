#!/bin/bash
# Profiles of the prefiltered search (round 3): kernel trace + PMC passes of the default bench step (C2) and of one rank's share of C4
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03prof_pf
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BUSY="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
kt() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o kt -- python3 "$@" > $OUT/kt_$name.log 2>&1; echo "kt $name rc=$?"
       f=$(find /tmp/kt_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/r03_${name}_kernel_stats.csv; }
pmc() { name=$1; ctr=$2; shift; shift; timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name} -o pmc -- python3 "$@" > $OUT/pmc_${name}.log 2>&1; echo "pmc $name rc=$?"; }
tojson() { python3 $R/tools/pmc_to_json.py "$@" > /dev/null; }
kt pf_c2 $R/bench.py --no-extras --no-cpu-baseline
pmc pf_c2_fetch FETCH_SIZE $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
pmc pf_c2_write WRITE_SIZE $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
pmc pf_c2_busy "$BUSY" $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3
tojson $OUT/r03_pf_c2_pmc.json "ms_scan_loader_kernel<10, 0, false, true>" "bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 3 (C2, prefiltered search: 1,000,000 x 128 rows, 256 queries, top-10)" 512000000 196608000000 /tmp/pmc_pf_c2_fetch /tmp/pmc_pf_c2_write /tmp/pmc_pf_c2_busy
kt pf_c4 $R/tools/pf_loop.py 45625000 4096 10 2
pmc pf_c4_busy "$BUSY" $R/tools/pf_loop.py 45625000 4096 10 1
pmc pf_c4_fetch FETCH_SIZE $R/tools/pf_loop.py 45625000 4096 10 1
tojson $OUT/r03_pf_c4_pmc.json "ms_scan_loader_kernel<10, 0, false, true>" "tools/pf_loop.py 45625000 4096 10 1 (one rank's share of C4, prefiltered search)" 23360000000 143523840000000 /tmp/pmc_pf_c4_fetch /tmp/pmc_pf_c4_busy
rm -f $OUT/pmc_*.log
ls $OUT
