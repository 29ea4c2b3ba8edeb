#!/usr/bin/env python3
"""Condenses the rocprofv3 passes of tools/pmc_profile.sh into <tag>_pmc_counters.json: per workload and kernel class
the per-launch mean of every collected counter plus a few derived figures, and copies the --stats table.
usage: summarize_pmc.py <pass dir> <out dir> <tag> <workload> [...]"""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("k_shadow", "k_closest", "k_shade", "k_tail", "k_resolve_pixels", "k_resolve", "k_roll_counters", "k_frame_begin")
N_SIMD = 1024  # 256 CUs x 4 SIMDs


def per_kernel(path):
    """kernel class -> counter -> list of per-dispatch values"""
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            for short in KERNELS:
                if "rt::" + short in name:
                    out.setdefault(short, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    break
    return out


def main():
    src, dst, tag = sys.argv[1:4]
    workloads = sys.argv[4:]
    os.makedirs(dst, exist_ok=True)
    res = {"_comment": "rocprofv3 --kernel-trace --pmc <one counter group per run> -- python3 bench.py --steps 2 --warmup 1 "
                       "--no-cpu-baseline --no-extras --workload <w>, one MI355X.  Every value is the mean over the launches of that "
                       "kernel class in the run (all recursion levels; 'largest' = the level-0 launch).  Derived: "
                       "valu_lane_ops_per_launch = SQ_INSTS_VALU x 64 (a wave64 instruction occupies 64 lanes whatever its exec mask); "
                       "salu_per_valu = SQ_INSTS_SALU / SQ_INSTS_VALU; lanes_active = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU ... per 64; "
                       "(no VALU-busy fraction is derived from SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE: the latter is summed over the 8 XCDs and the "
                       "former's unit could not be pinned on this chip -- the figure to read is valu_issue_frac, executed instructions over kernel time); "
                       "wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                       "l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); bytes_per_launch = 2 x 1024 x FETCH_SIZE (gfx950 tallies "
                       "128-B read requests at 64 B) + 1024 x WRITE_SIZE.",
           "workloads": {}}
    for wl in workloads:
        merged = {}
        for p in ("sq1", "sq2", "tcc", "fetch", "write"):
            files = sorted(glob.glob(os.path.join(src, wl, p, "**", "*counter_collection.csv"), recursive=True))
            if not files:
                continue
            for k, d in per_kernel(files[0]).items():
                merged.setdefault(k, {}).update(d)
        kernels = {}
        for k, d in merged.items():
            e = {"launches": max(len(v) for v in d.values())}
            for c, v in sorted(d.items()):
                e[c] = sum(v) / len(v)
            if "SQ_INSTS_VALU" in e:
                e["valu_lane_ops_per_launch"] = e["SQ_INSTS_VALU"] * 64.0
                e["largest_launch_valu_lane_ops"] = max(d["SQ_INSTS_VALU"]) * 64.0
                if e.get("SQ_INSTS_SALU") is not None and e["SQ_INSTS_VALU"]:
                    e["salu_per_valu"] = e["SQ_INSTS_SALU"] / e["SQ_INSTS_VALU"]
                if e.get("SQ_THREAD_CYCLES_VALU") and e.get("SQ_ACTIVE_INST_VALU"):
                    e["lanes_active"] = e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"]
                if e.get("SQ_WAVE_CYCLES"):
                    e["wait_inst_frac"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
                if e.get("SQ_WAVES"):
                    e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
            if e.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
            if "TCC_HIT_sum" in e and (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]) > 0:
                e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["read_bytes_per_launch"] = 2.0 * 1024.0 * e["FETCH_SIZE"]
                e["write_bytes_per_launch"] = 1024.0 * e["WRITE_SIZE"]
                e["bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
            kernels[k] = e
        res["workloads"][wl] = {"kernels": kernels}
    with open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w") as f:
        json.dump(res, f, indent=1)
    stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats.csv"))
    for wl, d in res["workloads"].items():
        for k in ("k_shadow", "k_closest"):
            e = d["kernels"].get(k, {})
            print(wl, k, {c: (round(v, 4) if isinstance(v, float) and v < 10 else v) for c, v in e.items() if c in (
                "launches", "valu_lane_ops_per_launch", "salu_per_valu", "lanes_active", "wait_inst_frac", "lds_bank_conflict_frac", "l2_hit_rate", "bytes_per_launch", "valu_insts_per_wave")})


if __name__ == "__main__":
    main()
