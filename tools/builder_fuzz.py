#!/usr/bin/env python3
"""Randomised check of the device-side Octree::Build against the host builder (GPU box): random triangle soups (slivers,
duplicates, coplanar layers, shared vertices), random octree parameters; node records, leaf slots, root link, stack bound
and PrintStats figures must be equal byte for byte, and a soup one builder rejects must be rejected by the other too.
usage: builder_fuzz.py [N=300] [first seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad, rejected, tris_total = [], 0, 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(np.exp(rs.uniform(np.log(1), np.log(120000))))
    mb, md = int(rs.choice([1, 2, 4, 8, 16, 64])), int(rs.randint(0, 13))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    lo = 0.01 if n_tri > 20000 else 0.03
    size = np.exp(rs.uniform(np.log(lo), np.log(4.0 if n_tri < 20000 else 0.5), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 3 == 0:
        pos[::3, :, int(rs.randint(0, 3))] = np.round(pos[::3, :, 0])
    if case % 5 == 0 and n_tri > 40:
        pos[::17, 2] = pos[::17, 1] + (pos[::17, 1] - pos[::17, 0]) * 1e-3 + 1e-4
        pos[5::40] = pos[4::40][: len(pos[5::40])]
        pos[7::23, 0] = pos[6::23, 0][: len(pos[7::23])]
    res = {}
    for builder in ("host", "device"):
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md, "builder": builder})
        try:
            g.Build(pos)
            nodes, slots, root, max_stack, on_device = g.export_tree()
            ts = g.tree_stats()
            res[builder] = (nodes.tobytes(), slots.tobytes(), root, max_stack, tuple(ts[k] for k in ("max_depth", "nodes", "leaves", "max_triangles_in_leaf", "avg_leaf_depth", "avg_triangles_in_leaf", "leaf_refs", "internal_nodes")))
            assert on_device == (builder == "device")
        except RuntimeError as e:
            res[builder] = "rejected: " + ("no octant" if "no octant" in str(e) else str(e)[:60])
        g.close()
    tris_total += n_tri
    if isinstance(res["host"], str) or isinstance(res["device"], str):
        rejected += 1
        same = isinstance(res["host"], str) and isinstance(res["device"], str) and res["host"].split(":")[0] == res["device"].split(":")[0]
    else:
        h, d = res["host"], res["device"]
        # (a root-leaf tree has no leaves in the sense of PrintStats: its averages are 0 / 0 = NaN in both builders)
        same = h[:4] == d[:4] and np.array_equal(np.asarray(h[4], np.float64), np.asarray(d[4], np.float64), equal_nan=True)
        if not same:
            print("   differs in:", [k for k, (x, y) in zip(("nodes", "slots", "root", "max_stack"), zip(h[:4], d[:4])) if x != y], h[4], d[4], flush=True)
    if not same:
        bad.append(seed0 + case)
        print(f"seed {seed0 + case} n_tri {n_tri} mb {mb} md {md}: builders differ ({res['host'] if isinstance(res['host'], str) else 'tree'} / "
              f"{res['device'] if isinstance(res['device'], str) else 'tree'})", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {len(bad)} differ, {rejected} rejected", flush=True)
print(f"builder_fuzz: {n_cases} scenes, {tris_total} triangles, scenes on which the two builders differ: {len(bad)} {bad}; rejected: {rejected}")
sys.exit(1 if bad else 0)
