#!/usr/bin/env python3
"""Randomised check of the per-(light, hit triangle) lists of the small-tree form (GPU box): N random triangle soups small
enough for that form, every third with a third of its triangles in common planes (hit triangles exactly in the plane of
others) and every fourth with slivers, random emissive triangles, octree parameters, cameras near and far, light-sample
counts and recursion depths; every frame is rendered with the lists and with RT_HIT_CULL=0 and must be the same bit for
bit with the same ray counts.
usage: hit_cull_fuzz.py [N=300] [first seed=1000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mats = Scene.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "cornell_sphere.npz")).materials
emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
plain = [m for m in range(mats.shape[0]) if m not in emissive]
pairs_total = left_total = rays_total = with_lists = 0
bad = []
skipped = []
for case in range(n_cases):
    print(f"case {case}", file=sys.stderr, flush=True)
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(rs.randint(8, 700))
    mb, md = [(64, 1), (int(rs.choice([16, 32, 64])), int(rs.randint(1, 3)))][case % 2]
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.05), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 3 == 0:
        pos[: n_tri // 3, :, 1] = np.round(pos[: n_tri // 3, :, 1])
    if case % 4 == 1:  # slivers: a vertex moved (almost) onto the opposite edge
        k = rs.choice(n_tri, max(n_tri // 10, 1), replace=False)
        w = rs.rand(len(k), 1).astype(np.float32)
        pos[k, 2] = pos[k, 0] * w + pos[k, 1] * (1 - w) + (rs.randn(len(k), 3) * 1e-4).astype(np.float32)
    matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
    n_lights = int(rs.randint(1, 6))
    matid[rs.choice(n_tri, min(n_lights, n_tri), replace=False)] = emissive[0]
    far = case % 7 == 3
    cam_pos = (rs.rand(3) * 4.0 - 2.0 + np.array([0.0, 0.0, -300.0 if far else -13.0])).astype(np.float32)
    kw = dict(fov=float(rs.uniform(0.05, 0.1) if far else rs.uniform(0.5, 1.0)), width=64, height=48, spt=int(rs.choice([1, 3, 8, 16, 33])),
              render_type=str(rs.choice(["phong", "blinn_phong"])), shading="flat", max_depth=int(rs.randint(0, 4)), cam_pos=[float(x) for x in cam_pos],
              cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=500.0, aspect_mode=1, seed=int(rs.randint(1, 1 << 30)), cull=bool(case % 9 != 4))
    cfg = rt.make_render_config(**kw)
    out = {}
    for flag in ("1", "0"):
        os.environ["RT_HIT_CULL"] = flag
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
        try:
            g.Build(pos)
        except RuntimeError:  # (a soup on which Octree::Build's own invariant fails, octree.cc:124-152: the reference exits there)
            skipped.append(seed0 + case)
            g.close()
            break
        g.set_shading(None, matid, mats)
        if flag == "1":
            pairs, left = g.hit_lists()
            pairs_total += pairs
            left_total += left
            with_lists += 1 if pairs else 0
        img, st = g.render(cfg)
        out[flag] = (img, (st.rays, st.secondary_rays, st.shadow_rays))
        g.close()
    if len(out) < 2:
        continue
    rays_total += out["1"][1][0]
    if not (np.array_equal(out["1"][0].view(np.uint32), out["0"][0].view(np.uint32)) and out["1"][1] == out["0"][1]):
        bad.append(seed0 + case)
        print(f"seed {seed0 + case}: frames differ", flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, {len(bad)} differ", flush=True)
print(f"hit_cull_fuzz: {n_cases} scenes ({with_lists} with per-hit-triangle lists), {rays_total} rays, {left_total} of {pairs_total} (hit triangle, listed triangle) "
      f"pairs left out of the lists; frames that differ from those of the lights' lists: {len(bad)} {bad}; scenes the octree build rejects: {len(skipped)}")
sys.exit(1 if bad else 0)
