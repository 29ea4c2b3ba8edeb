#!/usr/bin/env python3
"""Randomised check of the query entry points against the CPU restatement (GPU box): random triangle soups and octree
parameters; random rays of which a share has zero, denormal, huge, infinite or NaN components, origins on vertices and
directions along edges; ragged batch sizes.  rt_ads_trace_closest (t and triangle bit for bit), rt_ads_trace_any (flags),
rt_ads_search (candidate lists element for element), culled and two-sided, traversal forms forced in rotation.
usage: query_fuzz.py [N=200] [first seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mats = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_sphere.npz")).materials
special = np.array([0.0, -0.0, 1e-40, -1e-40, 1e30, -1e30, np.inf, -np.inf, np.nan, 1.0], np.float32)
bad, skipped, rays_total = [], 0, 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(np.exp(rs.uniform(np.log(1), np.log(20000))))
    mb, md = int(rs.choice([1, 2, 4, 8, 16, 64])), int(rs.randint(0, 11))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.03), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 4 == 0:
        pos[::3, :, int(rs.randint(0, 3))] = np.round(pos[::3, :, 0])
    if case % 7 == 0 and n_tri > 10:
        pos[5::9] = pos[4::9][: len(pos[5::9])]
    matid = np.zeros(n_tri, np.int32)
    env = [{}, {"RT_CLIP": "0"}, {"RT_TRAVERSAL": "lane", "RT_DISABLE_SMALL_TREE": "1"}, {"RT_TRAVERSAL": "pool", "RT_DISABLE_SMALL_TREE": "1"},
           {"RT_TRAVERSAL": "wave", "RT_DISABLE_SMALL_TREE": "1", "RT_PRUNE_MIN_SLOTS": "0"}, {"RT_BUILDER": "device"}][case % 6]
    for k in ("RT_CLIP", "RT_TRAVERSAL", "RT_DISABLE_SMALL_TREE", "RT_BUILDER", "RT_PRUNE_MIN_SLOTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
    try:
        g.Build(pos)
    except RuntimeError:
        skipped += 1
        g.close()
        continue
    n = int(rs.choice([1, 63, 65, 257, 3000]))
    lo, hi = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
    o = (lo + (hi - lo) * rs.rand(n, 3) * 1.4 - 0.2 * (hi - lo)).astype(np.float32)
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for i in range(0, n, 5):
        d[i, rs.randint(3)] = special[rs.randint(len(special))]
    for i in range(2, n, 11):
        o[i, rs.randint(3)] = special[rs.randint(len(special))]
    for i in range(3, n, 13):  # origin on a vertex, direction along an edge or towards another triangle's vertex
        j = rs.randint(n_tri)
        o[i] = pos[j, 0]
        d[i] = (pos[j, 1] - pos[j, 0]) if rs.randint(2) else (pos[rs.randint(n_tri), 2] - pos[j, 0])
    osc = orc.Scene.from_arrays(pos, None, matid, mats)
    ot = orc.Octree(osc, mb, md)
    why = []
    with np.errstate(all="ignore"):
        for cull in (True, False):
            t_ref, tri_ref = ot.trace_closest(o, d, cull)
            t, tri = g.trace_closest(o, d, cull)
            if not (np.array_equal(tri, tri_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))):
                why.append(f"closest cull={cull}")
            L = (o + d * np.float32(rs.uniform(0.2, 6.0))).astype(np.float32)
            if not np.array_equal(ot.trace_any(o, L, cull), g.trace_any(o, L, cull)):
                why.append(f"any cull={cull}")
        m = min(n, 40)
        lists = g.search_batch(np.concatenate([o[:m], d[:m]], axis=1))
        for i in range(m):
            if not np.array_equal(lists[i], ot.search(o[i], d[i])):
                why.append(f"search {i}")
                break
    rays_total += 4 * n + m
    g.close()
    if why:
        bad.append(seed0 + case)
        print(f"seed {seed0 + case} {env} n_tri {n_tri} mb {mb} md {md} n {n}: {why}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {len(bad)} differ, {skipped} skipped", flush=True)
print(f"query_fuzz: {n_cases} scenes, {rays_total} queries, scenes that differ from the CPU restatement: {len(bad)} {bad}; scenes the octree build rejects: {skipped}")
sys.exit(1 if bad else 0)
