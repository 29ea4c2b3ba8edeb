#!/bin/bash
# Run on the GPU box from the repository root: collects the rocprofv3 kernel statistics and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel-trace only) of the default bench command into gpurun_out/prof/,
# then writes the summaries that are kept under profiles/ into gpurun_out/profiles_new/.
# usage: tools/refresh_profiles.sh <tag>        e.g. r01
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles_new"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass done"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1
echo "write pass done"
cd "$ROOT"
# the bench line of the profiled run itself: its HIP-event kernel times are the ones to compare with the stats table
grep -o '{"metric.*' "$OUT/stats.log" > "$ROOT/gpurun_out/profiles_new/${TAG}_rocprofv3_bench_line.json" || true
python3 tools/summarize_profiles.py "$OUT" "$ROOT/gpurun_out/profiles_new" "$TAG"
