#!/bin/bash
# A/B of environment settings: tools/ab_env.sh <steps> "<ENV=.. ENV=..>" ["<ENV=..>" ...] -- runs the three single-GPU workloads under each setting.
STEPS=$1; shift
WLS=${AB_WORKLOADS:-cornell_original_blinn_spt16_1080p cornell_sphere_blinn_smooth_spp4_1080p synthetic_ico8_phong_4k}
for setting in "$@"; do
  echo "== $setting"
  for w in $WLS; do
    env $setting timeout -k 10 150 bash tools/runb.sh $w $STEPS || exit 1
  done
done
