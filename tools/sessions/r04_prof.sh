#!/bin/bash
# Profiles of round 4: kernel traces + PMC passes (separate runs per counter set) for the bench's top-level step (C2, fp32 scan), one
# rank's share of C4, the c3_search workload, the HBM-bound shapes, k = 64 and the encoder (split-bf16 edge GEMM, and the fp32 form);
# the prefiltered search: tools/sessions/r04_prof_pf.sh.  Summaries are built here (tools/pmc_to_json.py) and only they travel back.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BUSY="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
kt() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o kt -- python3 "$@" > $OUT/kt_$name.log 2>&1; echo "kt $name rc=$?"
       f=$(find /tmp/kt_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/r04_${name}_kernel_stats.csv; }
pmc() { name=$1; ctr=$2; shift; shift; timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name} -o pmc -- python3 "$@" > $OUT/pmc_${name}.log 2>&1; echo "pmc $name rc=$?"; }
tojson() { python3 $R/tools/pmc_to_json.py "$@" > /dev/null; }
# C2: the bench's top-level step (fp32 scan)
kt c2 $R/bench.py --no-extras --no-cpu-baseline --no-prefilter
pmc c2_fetch FETCH_SIZE $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --steps 5 --warmup 3
pmc c2_write WRITE_SIZE $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --steps 5 --warmup 3
pmc c2_busy "$BUSY" $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --steps 5 --warmup 3
tojson $OUT/r04_c2_pmc.json "ms_scan_loader_kernel<5, 0, false, false>" "bench.py --no-extras --no-cpu-baseline --no-prefilter --steps 5 --warmup 3 (C2: 1,000,000 x 128 rows, 256 queries, top-10, fp32 scan)" 512000000 65536000000 /tmp/pmc_c2_fetch /tmp/pmc_c2_write /tmp/pmc_c2_busy
# one rank's share of C4 (fp32 scan)
kt c4 $R/tools/prof_scan.py 45625000 4096 10 2
pmc c4_busy "$BUSY" $R/tools/prof_scan.py 45625000 4096 10 1
pmc c4_fetch FETCH_SIZE $R/tools/prof_scan.py 45625000 4096 10 1
pmc c4_write WRITE_SIZE $R/tools/prof_scan.py 45625000 4096 10 1
tojson $OUT/r04_c4_pmc.json "ms_scan_loader_kernel<5, 0, false, false>" "tools/prof_scan.py 45625000 4096 10 1 (one rank's share of C4, fp32 scan)" 23360000000 47841280000000 /tmp/pmc_c4_fetch /tmp/pmc_c4_write /tmp/pmc_c4_busy
# c3_search: cosine + length mask on unit rows (fp32 scan)
kt c3 $R/tools/prof_c3.py 20
pmc c3_busy "$BUSY" $R/tools/prof_c3.py 5
pmc c3_fetch FETCH_SIZE $R/tools/prof_c3.py 5
pmc c3_write WRITE_SIZE $R/tools/prof_c3.py 5
tojson $OUT/r04_c3_pmc.json "ms_scan_loader_kernel<5, 2, false, false>" "tools/prof_c3.py 5 (c3_search: 500,000 unit rows + lengths, 1000 queries, mincov 0.7, top-10, fp32 scan)" 258000000 128000000000 /tmp/pmc_c3_fetch /tmp/pmc_c3_write /tmp/pmc_c3_busy
# HBM-bound regime (one call per search)
for shape in "1000000 1" "1000000 32" "4000000 1" "4000000 32" "45625000 1" "45625000 32"; do set -- $shape
  kt hbm_$1_$2 $R/tools/hbm_shape.py $1 $2 40
  pmc hbm_$1_$2_fetch FETCH_SIZE $R/tools/hbm_shape.py $1 $2 4
  tojson $OUT/r04_hbm_$1_$2_pmc.json "ms_scan_kernel<5, false, false>" "tools/hbm_shape.py $1 $2 4" $(( $1 * 512 )) $(( $1 * 256 * 32 * (($2 + 31) / 32) )) /tmp/pmc_hbm_$1_$2_fetch
done
# k = 64 in the loader-wave form
kt k64 $R/tools/ksweep.py 64
# encoder: split-bf16 edge GEMM (default) and the fp32 form
kt egnn $R/tools/prof_egnn.py 1000 3
pmc egnn_busy "$BUSY" $R/tools/prof_egnn.py 1000 2
tojson $OUT/r04_egnn_pmc.json "ms_egnn_edge_kernel<true>" "tools/prof_egnn.py 1000 2 (1000 TED-length domains, split-bf16 edge GEMM)" 0 0 /tmp/pmc_egnn_busy
export MS_EGNN_SPLIT=0
kt egnn_fp32 $R/tools/prof_egnn.py 1000 3
unset MS_EGNN_SPLIT
rm -f $OUT/pmc_*.log
ls $OUT; du -sh $OUT
for f in $OUT/*_pmc.json; do echo $f; grep -E "traffic_over|matrix_pipe" $f; done
