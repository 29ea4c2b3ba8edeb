#!/bin/bash
# Per-dispatch kernel durations of a few frames of one workload (rocprofv3 kernel trace), printed as a table.
# usage: tools/trace_frame.sh [workload]            (bench.py, 2 steps)
#        tools/trace_frame.sh --cmd <python script and arguments>
WL=${1:-cornell_original_blinn_spt16_1080p}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ "$1" = "--cmd" ]; then
  shift
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$@" > "$OUT/log.txt" 2>&1
else
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload $WL > "$OUT/log.txt" 2>&1
fi
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-int(os.environ.get('TRACE_ROWS', '40')):]:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "").replace("rt::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  grid {r.get('Grid_Size', '?'):>10}  wg {r.get('Workgroup_Size', '?'):>4}  {short}")
PY
