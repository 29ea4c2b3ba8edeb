#!/bin/bash
# Run on the GPU box from the repository root: everything behind profiles/<tag>_*: the counter passes of the three
# single-GPU workloads (tools/pmc_profile.sh -> gpurun_out/profiles_new/), one default bench run
# (gpurun_out/bench_default.json) and one rank's share of 1/2/4/8-way splits of every workload (gpurun_out/band_times.txt).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 bash tools/pmc_profile.sh r03 > gpurun_out/pmc_profile.log 2>&1
echo pmc done
RT_BENCH_COUNTERS=gpurun_out/profiles_new/r03_pmc_counters.json timeout -k 10 300 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo bench done
: > gpurun_out/band_times.txt
for w in cornell_original_blinn_spt16_1080p synthetic_ico8_phong_4k cornell_sphere_blinn_smooth_spp4_1080p; do
  timeout -k 10 200 python3 tools/band_time.py $w >> gpurun_out/band_times.txt 2>/dev/null
done
BAND_WORLDS=1,8 BAND_FRAMES=2 timeout -k 10 200 python3 tools/band_time.py cornell_original_blinn_spp64_spt32_4k >> gpurun_out/band_times.txt 2>/dev/null
echo band done
# the 1.31 M-triangle frame with 3 concurrent sub-frames per call (rt_ads_set_render_streams)
RT_RENDER_STREAMS=3 timeout -k 10 200 python3 tools/band_time.py synthetic_ico8_phong_4k > gpurun_out/band_times_streams3.txt 2>/dev/null
echo band streams done
# one rank's share with 4 frames in flight (rt_render_submit / rt_render_wait) beside the blocking call (depth 0)
: > gpurun_out/inflight_times.txt
for w in cornell_original_blinn_spt16_1080p synthetic_ico8_phong_4k cornell_sphere_blinn_smooth_spp4_1080p; do
  INFLIGHT_WORLDS=1,2,4,8 INFLIGHT_DEPTHS=0,4 timeout -k 10 200 python3 tools/inflight_time.py $w >> gpurun_out/inflight_times.txt 2>/dev/null
done
echo in-flight done
