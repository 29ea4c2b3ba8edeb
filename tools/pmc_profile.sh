#!/bin/bash
# Counter evidence for the three single-GPU workloads: separate rocprofv3 passes (kernel trace + one counter group each,
# the program directly after `--`) of a short bench run, condensed by tools/summarize_pmc.py into
# gpurun_out/profiles_new/<tag>_pmc_counters.json (+ the --stats table of the default workload).
# usage (on the GPU box, from the repository root): tools/pmc_profile.sh <tag> [workload ...]
TAG=${1:-r03}; shift
export RT_BENCH_STREAMS=1  # counters and durations per kernel: no concurrent sub-frames
WLS=${@:-cornell_original_blinn_spt16_1080p cornell_sphere_blinn_smooth_spp4_1080p synthetic_ico8_phong_4k}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/profiles_new"
export TMPDIR=/tmp
cd /tmp
for WL in $WLS; do
  ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload $WL"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$WL/sq1" -- python3 $ARGS > "$OUT/$WL.sq1.log" 2>&1 || { echo "pass sq1 of $WL failed"; exit 1; }
  echo "$WL sq1 done"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/$WL/sq2" -- python3 $ARGS > "$OUT/$WL.sq2.log" 2>&1 || { echo "pass sq2 of $WL failed"; exit 1; }
  echo "$WL sq2 done"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/$WL/tcc" -- python3 $ARGS > "$OUT/$WL.tcc.log" 2>&1 || { echo "pass tcc of $WL failed"; exit 1; }
  echo "$WL tcc done"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$WL/fetch" -- python3 $ARGS > "$OUT/$WL.fetch.log" 2>&1 || { echo "pass fetch of $WL failed"; exit 1; }
  echo "$WL fetch done"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$WL/write" -- python3 $ARGS > "$OUT/$WL.write.log" 2>&1 || { echo "pass write of $WL failed"; exit 1; }
  echo "$WL write done"
done
# the --stats table and the bench line of the same profiled run (default command, default workload)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; exit 1; }
echo "stats done"
cd "$ROOT"
grep -o '{"metric.*' "$OUT/stats.log" > "$ROOT/gpurun_out/profiles_new/${TAG}_rocprofv3_bench_line.json" || true
python3 tools/summarize_pmc.py "$OUT" "$ROOT/gpurun_out/profiles_new" "$TAG" $WLS
