#!/bin/bash
# A/B of library builds: tools/ab.sh <steps> <lib.so> [<lib.so> ...]  -- runs the three single-GPU workloads with each library.
STEPS=$1; shift
for lib in "$@"; do
  echo "== $lib"
  for w in cornell_original_blinn_spt16_1080p cornell_sphere_blinn_smooth_spp4_1080p synthetic_ico8_phong_4k; do
    RT_LIB_PATH=$(pwd)/raytracer_amd/lib/$lib timeout -k 10 150 bash tools/runb.sh $w $STEPS || exit 1
  done
done
