#!/bin/bash
# Profiles of the prefiltered search over the split image (round 4): kernel trace + PMC passes (separate runs per counter set)
# of C2, of one rank's share of C4 and of the c3_search workload; summaries only travel back (tools/pmc_to_json.py).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04prof_pf
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BUSY="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
kt() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o kt -- python3 "$@" > $OUT/kt_$name.log 2>&1; echo "kt $name rc=$?"
       f=$(find /tmp/kt_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/r04_${name}_kernel_stats.csv; }
pmc() { name=$1; ctr=$2; shift; shift; timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name} -o pmc -- python3 "$@" > $OUT/pmc_${name}.log 2>&1; echo "pmc $name rc=$?"; }
tojson() { python3 $R/tools/pmc_to_json.py "$@" > /dev/null; }
kt pf_c2 $R/tools/pf_loop.py 1000000 256 10 200
pmc pf_c2_fetch FETCH_SIZE $R/tools/pf_loop.py 1000000 256 10 5
pmc pf_c2_write WRITE_SIZE $R/tools/pf_loop.py 1000000 256 10 5
pmc pf_c2_busy "$BUSY" $R/tools/pf_loop.py 1000000 256 10 5
tojson $OUT/r04_pf_c2_pmc.json "ms_scan_pf2_kernel<10, 8, false, false>" "tools/pf_loop.py 1000000 256 10 5 (C2, prefiltered search over the split image: 1,000,000 x 128 rows, 256 queries, top-10)" 512000000 65536000000 /tmp/pmc_pf_c2_fetch /tmp/pmc_pf_c2_write /tmp/pmc_pf_c2_busy
kt pf_c4 $R/tools/pf_loop.py 45625000 4096 10 2
pmc pf_c4_busy "$BUSY" $R/tools/pf_loop.py 45625000 4096 10 1
pmc pf_c4_fetch FETCH_SIZE $R/tools/pf_loop.py 45625000 4096 10 1
tojson $OUT/r04_pf_c4_pmc.json "ms_scan_pf2_kernel<10, 8, false, false>" "tools/pf_loop.py 45625000 4096 10 1 (one rank's share of C4, prefiltered search over the split image)" 23360000000 47841280000000 /tmp/pmc_pf_c4_fetch /tmp/pmc_pf_c4_busy
kt pf_c3 $R/tools/prof_c3.py 20 prefiltered
pmc pf_c3_busy "$BUSY" $R/tools/prof_c3.py 5 prefiltered
pmc pf_c3_fetch FETCH_SIZE $R/tools/prof_c3.py 5 prefiltered
tojson $OUT/r04_pf_c3_pmc.json "ms_scan_pf2_kernel<10, 8, false, true>" "tools/prof_c3.py 5 prefiltered (c3_search: 500,000 unit rows + lengths, 1000 queries, mincov 0.7, top-10)" 258000000 128000000000 /tmp/pmc_pf_c3_fetch /tmp/pmc_pf_c3_busy
rm -f $OUT/pmc_*.log
ls $OUT
for f in $OUT/*_pmc.json; do echo $f; grep -E "traffic_over|matrix_pipe" $f; done
for f in $OUT/*kernel_stats.csv; do echo $f; head -4 $f; done
