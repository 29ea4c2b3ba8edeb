#!/bin/bash
# One rocprofv3 counter pass over any python script; prints per-kernel totals and means of the requested counters.
# usage: tools/pmc_cmd.sh <counter> [<counter> ...] -- <script.py> [args]     (at most 8 SQ_, 4 TCC_ per pass)
CNT=()
while [ "$1" != "--" ]; do CNT+=("$1"); shift; done
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_cmd
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "${CNT[@]}" --output-format csv -d "$OUT" -- python3 $SCRIPT "$@" > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "rt::" not in name or "build" in name: continue
    short = name.split("(")[0].replace("void ", "").replace("rt::", "") + " grid " + r["Grid_Size"]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}  total {sum(v):16.1f}")
PY
