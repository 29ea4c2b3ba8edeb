#!/bin/bash
# (measurement helper, GPU box, repository root) 1/8 share of the 1.31 M-triangle frame with frames in flight under several settings
for s in "RT_X=1" "RT_TAIL_WAVES=256" "RT_TAIL_WAVES=64" "RT_TAIL_MAX_RAYS=1048576" "RT_TAIL_MAX_RAYS=1048576 RT_TAIL_WAVES=256" "RT_TAIL_MAX_RAYS=65536" "RT_LANE_WAVES=512" "RT_LANE_WAVES=512 RT_TAIL_WAVES=256 RT_TAIL_MAX_RAYS=1048576" "RT_TAIL_MAX_RAYS=0"; do
  echo "== $s"
  env $s INFLIGHT_WORLDS=${W:-8} INFLIGHT_DEPTHS=${D:-2,3,4,5,6} INFLIGHT_FRAMES=60 timeout -k 10 100 python3 tools/inflight_time.py ${WL:-synthetic_ico8_phong_4k} 2>/dev/null | cut -d' ' -f3-8 || exit 1
done
