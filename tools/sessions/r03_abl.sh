cd $GRAFT_REPO_ROOT
for v in noins abl_EXEC1 abl_NOFLAG abl_NOCMP; do echo "== $v"; MS_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build/$v/libmerizo_search_amd.so timeout 200 python tools/stamp_scan.py 4000000,256,10 2>&1 | grep "^n=\|loader cycles\|flag misses"; done
