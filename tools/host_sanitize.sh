#!/bin/bash
# Builds the library with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side only (device code is
# compiled as usual) and runs the host-only entry points -- octree build / export, OBJ / MTL loading with good and bad
# files, PPM output, name lookup -- from a small C++ driver.  No GPU needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/rt_host_sanitize
mkdir -p $OUT
cd "$ROOT/raytracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -pthread \
  -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  -o $OUT/libraytracer_amd.so rt_capi.hip host_octree.cpp scene_io.cpp
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -g -fsanitize=address,undefined -I "$ROOT/include" "$ROOT/tools/host_sanitize_main.cc" \
  -L$OUT -lraytracer_amd -Wl,-rpath,$OUT -o $OUT/main
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 $OUT/main
