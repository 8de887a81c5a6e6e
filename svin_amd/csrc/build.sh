#!/bin/bash
# Builds libsvin_ba.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $SVIN_EXTRA_FLAGS"
mkdir -p obj
for f in kernels.hip marg.hip posegraph.hip; do
  if [ ! -f obj/$f.o ] || [ $f -nt obj/$f.o ] || [ ../../include/svin_pg.h -nt obj/$f.o ] || [ kernels.hpp -nt obj/$f.o ] || [ dmath.hpp -nt obj/$f.o ] || [ window.hpp -nt obj/$f.o ]; then
    hipcc $FLAGS -c $f -o obj/$f.o
  fi
done
for f in window.cpp capi.cpp host_eval.cpp; do
  if [ ! -f obj/$f.o ] || [ $f -nt obj/$f.o ] || [ kernels.hpp -nt obj/$f.o ] || [ dmath.hpp -nt obj/$f.o ] || [ window.hpp -nt obj/$f.o ] || [ ../../include/svin_ba.h -nt obj/$f.o ]; then
    hipcc $FLAGS -x hip -c $f -o obj/$f.o
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvin_ba.so obj/kernels.hip.o obj/marg.hip.o obj/posegraph.hip.o obj/window.cpp.o obj/capi.cpp.o obj/host_eval.cpp.o
echo "built $(cd .. && pwd)/libsvin_ba.so"
