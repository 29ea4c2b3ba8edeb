#!/bin/bash
# Copies what tools/refresh_round.sh left under gpurun_out/ into profiles/ (run here, after the GPU call).
# usage: tools/install_profiles.sh [tag]
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
cp gpurun_out/profiles_new/${TAG}_pmc_counters.json gpurun_out/profiles_new/${TAG}_rocprofv3_bench_line.json gpurun_out/profiles_new/${TAG}_rocprofv3_kernel_stats.csv profiles/
cp gpurun_out/bench_default.json profiles/${TAG}_bench_n1.json
{
  echo "# One rank's share of a K-way band split, rendered on one MI355X (tools/band_time.py; no gather). ms/frame = host clock over 20 frames."
  cat gpurun_out/band_times.txt
  echo
  echo "# the 1.31 M-triangle frame with 3 concurrent sub-frames per rt_render call (RT_RENDER_STREAMS=3, what bench.py uses for this workload's blocking loop); 'kernels' sums overlapping intervals here"
  cat gpurun_out/band_times_streams3.txt
  echo
  echo "# the same shares with frames in flight (tools/inflight_time.py: rt_render_submit / rt_render_wait, 8 hardware queues); depth 0 = the blocking rt_render, depth 4 = what bench.py keeps in flight; ms per frame(-share) = host clock / 40 frames"
  cat gpurun_out/inflight_times.txt
} > profiles/${TAG}_band_share_times.txt
ls -la profiles/${TAG}_*
