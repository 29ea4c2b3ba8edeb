#!/bin/bash
# Builds a profiling variant of the library (HERE, before the GPU call) or runs the profile with it (on the GPU box).
# usage: tools/lane_profile.sh build | tools/lane_profile.sh run [workload]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  make -C "$ROOT/raytracer_amd/csrc" OUT=../lib/libraytracer_amd_prof.so EXTRA="-DRT_LANE_PROFILE $EXTRA"
else
  RT_LIB_PATH="$ROOT/raytracer_amd/lib/libraytracer_amd_prof.so" python3 "$ROOT/tools/lane_profile.py" $2
fi
