#!/bin/bash
# Diagnostic builds of the library with extra -D flags, into build/<name>/ (git-ignored, travels to the GPU box);
# load one with MS_LIB_OVERRIDE=build/<name>/libmerizo_search_amd.so.   usage: tools/build_variant.sh NAME -DFLAG [...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$R/build/$name
mkdir -p $out/obj
pids=()
for f in $R/merizo_search_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c -o $out/obj/$b.o $f &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmerizo_search_amd.so $out/obj/*.o
rm -rf $out/obj
echo built $out/libmerizo_search_amd.so
