#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/gpurun_out/s52.log; : > $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_pf2.py 1000000,256,10 4000000,256,10 16000000,1024,10 >> $L 2>&1
export TMPDIR=/tmp; cd /tmp
for i in 1 2; do
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_c4f_$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/pf_loop.py 45625000 4096 10 1 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_to_json.py /tmp/c4f_$i.json "ms_scan_pf2_kernel<10, 8, false, false>" "x" 23360000000 47841280000000 /tmp/pmc_c4f_$i | grep traffic_over >> $L
done
