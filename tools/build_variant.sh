#!/bin/bash
# Builds a second copy of the library with extra compile flags (instrumentation such as -DSVIN_CHOL_TIMING) into
# build/variants/<name>.so (git-ignored, travels with gpurun) without touching the product build; use with SVIN_BA_LIB=<path>.
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/build/variants"; mkdir -p "$out/obj_$name"
cd "$root/svin_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $*"
pids=()
for f in kernels.hip marg.hip posegraph.hip resident.hip; do hipcc $FLAGS -c $f -o "$out/obj_$name/$f.o" & pids+=($!); done
for f in window.cpp capi.cpp host_eval.cpp; do hipcc $FLAGS -x hip -c $f -o "$out/obj_$name/$f.o" & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done   # (set -e: a failed compile stops here instead of linking stale objects)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$out/obj_$name"/*.o
echo "built $out/$name.so"
