#!/bin/bash
# Kernel timeline (with queue ids) of the last frames of a frames-in-flight run of one rank's share.
# usage: tools/trace_inflight.sh [workload]   (INFLIGHT_WORLDS, INFLIGHT_DEPTHS: one value each)
WL=${1:-synthetic_ico8_phong_4k}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_inflight
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
INFLIGHT_FRAMES=${INFLIGHT_FRAMES:-12} timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $ROOT/tools/inflight_time.py $WL > "$OUT/log.txt" 2>&1
cd "$ROOT"
tail -2 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(os.environ.get('TRACE_ROWS', '120')):]
t0 = int(rows[0]["Start_Timestamp"])
busy_end = t0; idle = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end: idle += s - busy_end
    busy_end = max(busy_end, e)
    short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} q{r.get('Queue_Id','?'):>3} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9} {short}")
print(f"span {(busy_end - t0) / 1e3:.1f} us, no kernel running for {idle / 1e3:.1f} us")
PY
