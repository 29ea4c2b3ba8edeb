#!/usr/bin/env python3
"""Randomised check of the per-light lists (GPU box): N random triangle soups with random emissive triangles, random
octree parameters, cameras and light-sample counts; every frame is rendered with the per-light lists and with the full
lists (RT_LIGHT_CULL=0) and must be the same bit for bit with the same ray counts.
usage: light_cull_fuzz.py [N=200] [first seed=1000] [first case=0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mats = Scene.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "cornell_sphere.npz")).materials
emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
plain = [m for m in range(mats.shape[0]) if m not in emissive]
forms = {"small": 0, "clustered": 0, "per-lane": 0, "none": 0}
pairs_total = left_total = rays_total = 0
bad = []
skipped = []
for case in range(first_case, n_cases):
    print(f"case {case}", file=sys.stderr, flush=True)
    rs = np.random.RandomState(seed0 + case)
    kind = case % 3
    n_tri, mb, md = [(int(rs.randint(20, 400)), 64, 1), (int(rs.randint(300, 2500)), int(rs.choice([4, 8, 16])), int(rs.randint(3, 6))),
                     (int(rs.randint(20000, 60000)), 16, 10)][kind]
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    lo, hi = (0.05, 3.0) if kind < 2 else (0.02, 0.4)
    size = np.exp(rs.uniform(np.log(lo), np.log(hi), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 5 == 0:  # coplanar layers and slivers: lights in or near the planes of other triangles
        pos[: n_tri // 3, :, 1] = np.round(pos[: n_tri // 3, :, 1])
    matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
    n_lights = int(rs.randint(1, 7))
    matid[rs.choice(n_tri, n_lights, replace=False)] = emissive[0]
    cam_pos = (rs.rand(3) * 4.0 - 2.0 + np.array([0.0, 0.0, -13.0])).astype(np.float32)
    kw = dict(fov=float(rs.uniform(0.5, 1.0)), width=64, height=48, spt=int(rs.choice([1, 3, 8, 33])), render_type=str(rs.choice(["phong", "blinn_phong"])),
              shading="flat", max_depth=int(rs.randint(0, 4)), cam_pos=[float(x) for x in cam_pos], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0],
              max_distance=40.0, aspect_mode=1, seed=int(rs.randint(1, 1 << 30)))
    cfg = rt.make_render_config(**kw)
    out = {}
    # how lights that do not share forms are served on big trees, in rotation: by the library's own choice, always one light
    # per pass, never merged
    for k in ("RT_LANE_SIDE_MAX_RAYS", "RT_LIGHT_MERGE"):
        os.environ.pop(k, None)
    os.environ.update([{}, {"RT_LANE_SIDE_MAX_RAYS": "0"}, {"RT_LIGHT_MERGE": "0"}, {"RT_LIGHT_MERGE": "0", "RT_LANE_SIDE_MAX_RAYS": "0"}][(case // 3) % 4])
    for flag in ("1", "0"):
        os.environ["RT_LIGHT_CULL"] = flag
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
        try:
            g.Build(pos)
        except RuntimeError as e:  # (a soup on which Octree::Build's own invariant fails, octree.cc:124-152: the reference exits there)
            skipped.append(seed0 + case)
            g.close()
            break
        g.set_shading(None, matid, mats)
        if flag == "1":
            lights, pairs, left = g.light_forms()
            small, clusters, _ = g.tree_forms()
            forms["none" if lights == 0 else ("small" if small else ("clustered" if clusters else "per-lane"))] += 1
            pairs_total += pairs
            left_total += left
        img, st = g.render(cfg)
        out[flag] = (img, (st.rays, st.secondary_rays, st.shadow_rays))
        g.close()
    if len(out) < 2:
        continue
    rays_total += out["1"][1][0]
    if not (np.array_equal(out["1"][0].view(np.uint32), out["0"][0].view(np.uint32)) and out["1"][1] == out["0"][1]):
        bad.append(seed0 + case)
        print(f"seed {seed0 + case}: frames differ", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {len(bad)} differ", flush=True)
print(f"light_cull_fuzz: {n_cases} scenes ({forms}), {rays_total} rays, {left_total} of {pairs_total} (light, triangle) pairs left out of the lists; "
      f"frames that differ from the full lists': {len(bad)} {bad}; scenes the octree build rejects: {len(skipped)}")
sys.exit(1 if bad else 0)
