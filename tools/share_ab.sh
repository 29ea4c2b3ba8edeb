#!/bin/bash
# (measurement helper; run on the GPU box from the repository root: the time of one rank's 1/8 share of the 1.31 M-triangle frame under several settings of the fused tail, the per-lane wave target and concurrent sub-frames)
# 1/8 share of the synthetic frame under several tail settings
for s in "RT_X=1" "RT_TAIL_WAVES=2048" "RT_TAIL_WAVES=4096" "RT_TAIL_WAVES=512" "RT_TAIL_MAX_RAYS=65536" "RT_TAIL_MAX_RAYS=65536 RT_TAIL_WAVES=2048" "RT_TAIL_MAX_RAYS=20000" "RT_LANE_WAVES=4096" "RT_RENDER_STREAMS=2" "RT_RENDER_STREAMS=3"; do
  echo "== $s"
  env $s BAND_WORLDS=8 BAND_FRAMES=20 timeout -k 10 100 python3 tools/band_time.py synthetic_ico8_phong_4k 2>/dev/null | cut -c1-120 || exit 1
done
