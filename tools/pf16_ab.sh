#!/bin/bash
# Round 6: same-box A/B of builds of the fp16-image scan (tools/build_variant.sh NAME -D...): per build, tools/pf2_try.py on the shapes given
# (bit equality with the fp32 scan, flagged queries, scan launch and whole call per arithmetic).   usage: tools/pf16_ab.sh "shape ..." build [build ...]
# build = a directory name under build/ or "default" (the shipped library)
shapes=$1; shift
for rep in 1 2; do
for b in "$@"; do
  if [ "$b" = default ]; then unset MS_LIB_OVERRIDE; else export MS_LIB_OVERRIDE=$PWD/build/$b/libmerizo_search_amd.so; fi
  echo "=== build $b (pass $rep)"
  timeout 600 python tools/pf2_try.py $shapes 2>&1 | grep -v "^$"
done
done
