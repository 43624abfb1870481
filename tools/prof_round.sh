#!/bin/bash
# Profiles of one round on the GPU box: rocprofv3 kernel traces (--kernel-trace --stats) and PMC passes (separate runs per counter set:
# FETCH_SIZE; WRITE_SIZE; busy counters) of the bench's workloads; summaries (tools/pmc_to_json.py) and kernel-stats CSVs travel back
# under gpurun_out/<tag>prof/ and are copied into profiles/ from there.
#   usage: bash tools/prof_round.sh TAG [part ...]     parts: pf c2 c4 c3 hbm k64 egnn   (default: all)
set -u
TAG=${1:-r06}; shift || true
PARTS=${*:-"pf c2 c4 c3 hbm k64 egnn"}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BUSY="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
kt() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -o kt -- python3 "$@" > $OUT/kt_$name.log 2>&1; echo "kt $name rc=$?"
       f=$(find /tmp/kt_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_${name}_kernel_stats.csv; }
pmc() { name=$1; ctr=$2; shift; shift; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name} -o pmc -- python3 "$@" > $OUT/pmc_${name}.log 2>&1; echo "pmc $name rc=$?"; }
tojson() { python3 $R/tools/pmc_to_json.py "$@" > /dev/null; }
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
PFK=${PFK:-"ms_scan_pf16_kernel<10, 8, false, false, 1>"}      # the image scan of the chosen arithmetic (F16X1 on i.i.d. data)
PFKM=${PFKM:-"ms_scan_pf16_kernel<10, 8, false, true, 1>"}     # ... with the length mask (c3_search)
if has pf; then
  kt pf_c2 $R/tools/pf_loop.py 1000000 256 10 200
  pmc pf_c2_fetch FETCH_SIZE $R/tools/pf_loop.py 1000000 256 10 5
  pmc pf_c2_write WRITE_SIZE $R/tools/pf_loop.py 1000000 256 10 5
  pmc pf_c2_busy "$BUSY" $R/tools/pf_loop.py 1000000 256 10 5
  tojson $OUT/${TAG}_pf_c2_pmc.json "$PFK" "tools/pf_loop.py 1000000 256 10 5 (C2, prefiltered search over the fp16 image: 1,000,000 x 128 rows, 256 queries, top-10)" 256000000 65536000000 /tmp/pmc_pf_c2_fetch /tmp/pmc_pf_c2_write /tmp/pmc_pf_c2_busy
  # (MS_PF_FORMAT=f16x1: no search of 256 of the database's own rows to choose the arithmetic -- its launches of the same kernel name polluted round 5's average)
  MS_PF_FORMAT=f16x1 kt pf_c4 $R/tools/pf_loop.py 45625000 4096 10 3
  pmc pf_c4_busy "$BUSY" $R/tools/pf_loop.py 45625000 4096 10 1
  pmc pf_c4_fetch FETCH_SIZE $R/tools/pf_loop.py 45625000 4096 10 1          # (4 launches: 3 warm-up + 1 -- every one is a pass over the shard)
  pmc pf_c4_write WRITE_SIZE $R/tools/pf_loop.py 45625000 4096 10 1
  tojson $OUT/${TAG}_pf_c4_pmc.json "$PFK" "tools/pf_loop.py 45625000 4096 10 1 (one rank's share of C4, prefiltered search over the fp16 image)" 11680000000 47841280000000 /tmp/pmc_pf_c4_fetch /tmp/pmc_pf_c4_write /tmp/pmc_pf_c4_busy
  # every launch of the fetch pass, not their mean: the traffic of this shape varies from pass to pass
  python3 - $OUT/${TAG}_pf_c4_pmc.json "$PFK" /tmp/pmc_pf_c4_fetch <<'PY'
import csv, glob, json, sys
out, kern, d = sys.argv[1:4]
vals = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
        if kern in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
doc = json.load(open(out))
doc["hbm_read_bytes_corrected_every_launch"] = [2.0 * v * 1024 for v in vals]
doc["read_over_algorithmic_every_launch"] = [2.0 * v * 1024 / doc["algorithmic_bytes_per_launch"] for v in vals]
json.dump(doc, open(out, "w"), indent=1)
print("pf_c4 read traffic over algorithmic, every launch:", ["%.2f" % x for x in doc["read_over_algorithmic_every_launch"]])
PY
  kt pf_c3 $R/tools/prof_c3.py 20 prefiltered
  pmc pf_c3_busy "$BUSY" $R/tools/prof_c3.py 5 prefiltered
  pmc pf_c3_fetch FETCH_SIZE $R/tools/prof_c3.py 5 prefiltered
  pmc pf_c3_write WRITE_SIZE $R/tools/prof_c3.py 5 prefiltered
  tojson $OUT/${TAG}_pf_c3_pmc.json "$PFKM" "tools/prof_c3.py 5 prefiltered (c3_search: 500,000 unit rows + lengths, 1000 queries, mincov 0.7, top-10)" 130000000 128000000000 /tmp/pmc_pf_c3_fetch /tmp/pmc_pf_c3_write /tmp/pmc_pf_c3_busy
fi
if has c2; then   # C2: the bench's top-level step (fp32 scan)
  kt c2 $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --no-pipelined --no-live-traffic
  pmc c2_fetch FETCH_SIZE $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --no-pipelined --no-live-traffic --steps 5 --warmup 3
  pmc c2_write WRITE_SIZE $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --no-pipelined --no-live-traffic --steps 5 --warmup 3
  pmc c2_busy "$BUSY" $R/bench.py --no-extras --no-cpu-baseline --no-prefilter --no-pipelined --no-live-traffic --steps 5 --warmup 3
  tojson $OUT/${TAG}_c2_pmc.json "ms_scan_loader_kernel<5, 0, false, false>" "bench.py --no-extras --no-cpu-baseline --no-prefilter --no-pipelined --no-live-traffic --steps 5 --warmup 3 (C2: 1,000,000 x 128 rows, 256 queries, top-10, fp32 scan)" 512000000 65536000000 /tmp/pmc_c2_fetch /tmp/pmc_c2_write /tmp/pmc_c2_busy
fi
if has c4; then   # one rank's share of C4 (fp32 scan)
  kt c4 $R/tools/prof_scan.py 45625000 4096 10 2
  pmc c4_busy "$BUSY" $R/tools/prof_scan.py 45625000 4096 10 1
  pmc c4_fetch FETCH_SIZE $R/tools/prof_scan.py 45625000 4096 10 1
  pmc c4_write WRITE_SIZE $R/tools/prof_scan.py 45625000 4096 10 1
  tojson $OUT/${TAG}_c4_pmc.json "ms_scan_loader_kernel<5, 0, false, false>" "tools/prof_scan.py 45625000 4096 10 1 (one rank's share of C4, fp32 scan)" 23360000000 47841280000000 /tmp/pmc_c4_fetch /tmp/pmc_c4_write /tmp/pmc_c4_busy
fi
if has c3; then   # c3_search: cosine + length mask on unit rows (fp32 scan)
  kt c3 $R/tools/prof_c3.py 20
  pmc c3_busy "$BUSY" $R/tools/prof_c3.py 5
  pmc c3_fetch FETCH_SIZE $R/tools/prof_c3.py 5
  pmc c3_write WRITE_SIZE $R/tools/prof_c3.py 5
  tojson $OUT/${TAG}_c3_pmc.json "ms_scan_loader_kernel<5, 2, false, false>" "tools/prof_c3.py 5 (c3_search: 500,000 unit rows + lengths, 1000 queries, mincov 0.7, top-10, fp32 scan)" 258000000 128000000000 /tmp/pmc_c3_fetch /tmp/pmc_c3_write /tmp/pmc_c3_busy
fi
if has hbm; then  # HBM-bound regime (one call per search)
  for shape in "1000000 1" "1000000 32" "4000000 1" "4000000 32" "45625000 1" "45625000 32"; do set -- $shape
    kt hbm_$1_$2 $R/tools/hbm_shape.py $1 $2 40
    pmc hbm_$1_$2_fetch FETCH_SIZE $R/tools/hbm_shape.py $1 $2 4
    tojson $OUT/${TAG}_hbm_$1_$2_pmc.json "ms_scan_kernel<5, false, false>" "tools/hbm_shape.py $1 $2 4" $(( $1 * 512 )) $(( $1 * 256 * 32 * (($2 + 31) / 32) )) /tmp/pmc_hbm_$1_$2_fetch
  done
fi
if has k64; then kt k64 $R/tools/ksweep.py 64; fi
if has egnn; then # encoder: split-bf16 edge GEMM (default) and the fp32 form
  kt egnn $R/tools/prof_egnn.py 1000 3
  pmc egnn_busy "$BUSY" $R/tools/prof_egnn.py 1000 2
  tojson $OUT/${TAG}_egnn_pmc.json "ms_egnn_edge_kernel<true>" "tools/prof_egnn.py 1000 2 (1000 TED-length domains, split-bf16 edge GEMM)" 0 0 /tmp/pmc_egnn_busy
  MS_EGNN_SPLIT=0 kt egnn_fp32 $R/tools/prof_egnn.py 1000 3
fi
rm -f $OUT/pmc_*.log
ls $OUT; du -sh $OUT
for f in $OUT/*_pmc.json; do echo $f; grep -E "traffic_over|matrix_pipe|hbm_write_bytes" $f; done
