#!/bin/bash
# Kernel-by-kernel timeline of the last frame of a short bench run (rocprofv3 kernel trace).
# usage: tools/trace_levels.sh [workload]
WL=${1:-synthetic_ico8_phong_4k}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_levels
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload $WL > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rt::" in r["Kernel_Name"] and "build" not in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if "k_resolve_pixels" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size')):>10} {r['Kernel_Name'].split('(')[0].replace('void rt::', '')[:40]}")
PY
