#!/bin/bash
# One rocprofv3 counter pass over a short bench run; prints per-kernel averages of the requested counters.
# usage: tools/pmc_pass.sh <workload> <counter> [<counter> ...]      (at most 8 SQ_, 4 TCC_, 4 TCP_ per pass)
WL=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_pass
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --workload $WL > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "rt::" not in name: continue
    short = name.split("(")[0].replace("void ", "").replace("rt::", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}")
PY
