set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for pace in 1 0; do
  for f in f16x1 f16x2; do
    echo "== pace=$pace $f"; MS_PF_PACE=$pace MS_PF_FORMAT=$f timeout 150 python3 $R/tools/pf_scan_only.py 16000000,1024,10 45625000,4096,10 2>&1 | grep "^n="
  done
  MS_PF_PACE=$pace MS_PF_FORMAT=f16x1 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_pace$pace -o pmc -- python3 $R/tools/pf_loop.py 45625000 4096 10 1 > $OUT/pmc_pace$pace.log 2>&1
  python3 - /tmp/pmc_pace$pace <<'PY'
import csv, glob, sys
vals = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
        if "ms_scan_pf16_kernel<10, 8, false, false, 1>" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("   HBM reads per launch / image bytes (11.68 GB):", ["%.2f" % (2 * v * 1024 / 11.68e9) for v in vals])
PY
done
cd $R; timeout 600 python3 -m pytest tests/test_prefilter_gpu.py -m gpu -x -q 2>&1 | tail -3
