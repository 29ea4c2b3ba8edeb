#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/refresh_profiles.sh into the two files kept under profiles/:
<tag>_rocprofv3_kernel_stats.csv (the --stats table as rocprofv3 wrote it) and <tag>_pmc_traffic.json
(FETCH_SIZE / WRITE_SIZE per launch of the three frame kernels, gfx950 correction applied to the reads).
usage: summarize_profiles.py <rocprof dir> <out dir> <tag>"""
import csv
import glob
import json
import os
import shutil
import sys


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {root}")
    return hits[0]


def per_kernel(path, counter):
    """kernel short name -> list of counter values (KiB) in dispatch order"""
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for short in ("k_shadow", "k_closest", "k_shade", "k_resolve_pixels", "k_resolve"):
                if "rt::" + short in name:
                    out.setdefault(short, []).append(float(row["Counter_Value"]))
                    break
    return out


def main():
    src, dst, tag = sys.argv[1:4]
    os.makedirs(dst, exist_ok=True)
    shutil.copy(find(os.path.join(src, "stats"), "*kernel_stats.csv"), os.path.join(dst, f"{tag}_rocprofv3_kernel_stats.csv"))
    fetch = per_kernel(find(os.path.join(src, "fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(find(os.path.join(src, "write"), "*counter_collection.csv"), "WRITE_SIZE")
    res = {
        "_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of "
                    "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (default workload), one MI355X. The counters "
                    "are in KiB; reads are doubled (gfx950 tallies 128-B read requests at 64 B), writes are taken as "
                    "reported. All values are bytes per launch, averaged over the launches of the run; 'largest' is the "
                    "level-0 launch (all camera-ray hits of a frame).",
        "workload": sys.argv[4] if len(sys.argv) > 4 else "cornell_original_blinn_spt16_1080p",
        "kernels": {},
    }
    for k in sorted(set(fetch) | set(write)):
        fr = [2.0 * 1024.0 * v for v in fetch.get(k, [])]
        wr = [1024.0 * v for v in write.get(k, [])]
        n = min(len(fr), len(wr))
        if n == 0:
            continue
        res["kernels"][k] = {
            "launches": n,
            "read_bytes_per_launch": sum(fr[:n]) / n,
            "write_bytes_per_launch": sum(wr[:n]) / n,
            "bytes_per_launch": (sum(fr[:n]) + sum(wr[:n])) / n,
            "largest_launch_bytes": max(a + b for a, b in zip(fr[:n], wr[:n])),
        }
    if "k_shadow" in res["kernels"]:
        res["k_shadow_bytes_per_launch_corrected"] = {"mean_over_launches": res["kernels"]["k_shadow"]["bytes_per_launch"],
                                                      "level0": res["kernels"]["k_shadow"]["largest_launch_bytes"]}
        res["note"] = ("algorithmic bytes of a level-0 k_shadow launch at 1080p: 2.07 M hits x (4 B work index + 4 B t + "
                       "4 B triangle index) read = 24.9 MB and 2.07 M x 2 lights x 4 B visibility words written = 16.6 MB")
    with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
