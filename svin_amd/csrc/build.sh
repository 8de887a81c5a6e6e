#!/bin/bash
# Builds libsvin_ba.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $SVIN_EXTRA_FLAGS"
mkdir -p obj
# every object depends on every header of the library (a stale object would travel to the GPU box inside the .so)
HEADERS="kernels.hpp options.hpp dmath.hpp window.hpp resident.hpp trust_region.hpp symeig.hpp tile16.hpp flat_map.hpp ../../include/svin_ba.h ../../include/svin_pg.h build.sh"
stale() {  # stale <object> <source>
  [ ! -f "$1" ] && return 0
  [ "$2" -nt "$1" ] && return 0
  for h in $HEADERS; do [ "$h" -nt "$1" ] && return 0; done
  return 1
}
for f in kernels.hip marg.hip posegraph.hip resident.hip; do
  if stale obj/$f.o $f; then hipcc $FLAGS -c $f -o obj/$f.o; fi
done
for f in window.cpp capi.cpp host_eval.cpp; do
  if stale obj/$f.o $f; then hipcc $FLAGS -x hip -c $f -o obj/$f.o; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvin_ba.so obj/kernels.hip.o obj/marg.hip.o obj/posegraph.hip.o obj/resident.hip.o obj/window.cpp.o obj/capi.cpp.o obj/host_eval.cpp.o
echo "built $(cd .. && pwd)/libsvin_ba.so"
