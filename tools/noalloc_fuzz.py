#!/usr/bin/env python3
"""Randomised check of rt_ads_search_noalloc against rt_ads_search (GPU box): random soups and octree parameters, random
batch sizes inside the entry point's limits, output buffers that are exact, too large or too small, calls back to back on
the one scratch array; offsets, totals and candidate lists must be equal and nothing may be written past the lists.
usage: noalloc_fuzz.py [N=300] [first seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _capi  # noqa: E402
from test_zz_no_alloc import _search_noalloc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad, refused, calls = [], 0, 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(np.exp(rs.uniform(np.log(1), np.log(8000))))
    mb, md = int(rs.choice([1, 2, 4, 8, 16, 64])), int(rs.randint(0, 9))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.03), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
    try:
        g.Build(pos)
    except RuntimeError:
        g.close()
        continue
    nodes = g.tree_stats()["nodes"]
    limit = min(_capi.RT_SEARCH_NOALLOC_MAX_RAYS, _capi.RT_SEARCH_NOALLOC_SCRATCH_BYTES // (max(nodes, 1) * 8))
    why = []
    for _ in range(3):
        n = int(rs.randint(0, max(limit, 1) + 1)) if rs.randint(4) else int(min(limit, rs.choice([1, 63, 64, 65])))
        o = (rs.rand(n, 3) * 14.0 - 7.0).astype(np.float32)
        d = rs.randn(n, 3).astype(np.float32)
        rays = np.concatenate([o, d], axis=1).astype(np.float32)
        ref = g.search_batch(rays)
        ref_off = np.concatenate([[0], np.cumsum([len(r) for r in ref])]).astype(np.uint64)
        total_ref = int(ref_off[-1])
        cap = int([total_ref, total_ref + 7, total_ref // 2][rs.randint(3)])
        try:
            off, out, total = _search_noalloc(g, rays, cap)
        except RuntimeError as e:
            refused += 1
            why.append(f"refused inside its limits: {e}")
            break
        calls += 1
        flat = np.concatenate(ref) if total_ref else np.zeros(0, np.uint32)
        k = min(cap, total_ref)
        if total != total_ref or not np.array_equal(off, ref_off) or not np.array_equal(out[:k], flat[:k]) or not (out[k:] == 0xffffffff).all():
            why.append(f"n {n} cap {cap} total {total} / {total_ref}")
    g.close()
    if why:
        bad.append(seed0 + case)
        print(f"seed {seed0 + case} n_tri {n_tri} mb {mb} md {md} nodes {nodes}: {why}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {len(bad)} differ", flush=True)
print(f"noalloc_fuzz: {n_cases} scenes, {calls} calls, scenes on which rt_ads_search_noalloc differs from rt_ads_search: {len(bad)} {bad}")
sys.exit(1 if bad else 0)
