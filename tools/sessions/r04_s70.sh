#!/bin/bash
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gpurun_out/s70.log; : > $L
for v in main fence; do
  if [ $v = main ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so; fi
  rm -rf /tmp/kt_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python3 $GRAFT_REPO_ROOT/tools/pf_loop.py 1000000 256 10 100 > /dev/null 2>&1
  f=$(find /tmp/kt_$v -name '*kernel_stats.csv' | head -1); echo "== $v" >> $L; python3 -c "import csv,sys; [print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1)) for r in csv.DictReader(open(sys.argv[1])) if 'rescore' in r['Name'] or 'pf2_kernel<10' in r['Name']]" $f >> $L
done
