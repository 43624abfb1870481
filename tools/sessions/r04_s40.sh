#!/bin/bash
# fact finding: k = 64 insertion share, sample-size sweep at C2 (fp32), HBM-regime variants
cd $GRAFT_REPO_ROOT
L=gpurun_out/s40.log; : > $L
echo "== stamps k=64/32/10" >> $L
MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/stamp/libmerizo_search_amd.so timeout 300 python tools/stamp_scan.py 1000000,256,64 1000000,256,32 1000000,256,10 >> $L 2>&1
for T in 2 3 5 7; do echo "== MS_PREPASS_TILES=$T" >> $L; MS_PREPASS_TILES=$T timeout 200 python tools/ksweep.py 10 64 2>&1 | grep "^k=" >> $L; done
echo "== default" >> $L; timeout 200 python tools/ksweep.py 10 64 2>&1 | grep "^k=" >> $L
echo "== hbm regime default" >> $L
for s in "1000000 32" "4000000 32" "1000000 16"; do timeout 100 python tools/hbm_shape.py $s >> $L 2>&1; done
echo "== hbm regime no sample" >> $L
for s in "1000000 32" "4000000 32" "1000000 16"; do MS_SAMPLE_MIN_NQ=100 timeout 100 python tools/hbm_shape.py $s >> $L 2>&1; done
echo "== hbm regime T=1,2" >> $L
for T in 1 2; do for s in "1000000 32" "4000000 32"; do MS_PREPASS_TILES=$T timeout 100 python tools/hbm_shape.py $s >> $L 2>&1; done; done
echo "== hbm regime fused merge up to 32" >> $L
for s in "1000000 32" "4000000 32"; do MS_FUSED_MERGE_MAX_NQ=32 MS_INKERNEL_NORM_MAX_NQ=32 MS_SAMPLE_MIN_NQ=100 timeout 100 python tools/hbm_shape.py $s >> $L 2>&1; done
