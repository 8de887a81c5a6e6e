#!/bin/bash
# Builds a second copy of the library with extra compile flags (instrumentation such as -DSVIN_CHOL_TIMING) into
# build/variants/<name>.so (git-ignored, travels with gpurun) without touching the product build; use with SVIN_BA_LIB=<path>.
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/build/variants"; mkdir -p "$out/obj_$name"
cd "$root/svin_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $*"
for f in kernels.hip marg.hip posegraph.hip resident.hip; do hipcc $FLAGS -c $f -o "$out/obj_$name/$f.o" & done
for f in window.cpp capi.cpp host_eval.cpp; do hipcc $FLAGS -x hip -c $f -o "$out/obj_$name/$f.o" & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$out/obj_$name"/*.o
echo "built $out/$name.so"
