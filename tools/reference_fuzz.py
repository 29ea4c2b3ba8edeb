#!/usr/bin/env python3
"""Randomised check against the UNMODIFIED REFERENCE BINARY (oracle/_ref/raytracer_ref, built from the reference's own
sources; GPU box): random triangle soups, random reflective / refractive materials, random octree parameters, cameras,
image sizes and culling, in the render types that draw no random numbers (distance, diffusion).  The reference renders the
frame on the host; the device path must give the same PPM bytes (md5), the same Search call count and the same tree shape.
usage: reference_fuzz.py [N=100] [first seed=1]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_runner  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

assert ref_runner.available(), "oracle/_ref/raytracer_ref is missing"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mats = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_sphere.npz")).materials
bad, rejected, rays_total = [], 0, 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(np.exp(rs.uniform(np.log(2), np.log(4000))))
    mb, md = int(rs.choice([1, 2, 4, 8, 16, 64])), int(rs.randint(0, 9))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.03), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 4 == 0:
        pos[::3, :, int(rs.randint(0, 3))] = np.round(pos[::3, :, 0])
    if case % 7 == 0 and n_tri > 10:
        pos[5::9] = pos[4::9][: len(pos[5::9])]
    matid = rs.randint(0, mats.shape[0], n_tri).astype(np.int32)
    cpos = [float(x) for x in (rs.rand(3) * 4.0 - 2.0 + np.array([0.0, 0.0, -13.0]))]
    cdir, fov = [0.0, 0.0, 1.0], float(rs.uniform(0.4, 1.1))
    w, h = int(rs.choice([24, 48, 80])), int(rs.choice([24, 60, 100]))
    rtype, cull, depth = str(rs.choice(["distance", "diffusion"])), bool(rs.randint(0, 2)), int(rs.randint(0, 6))
    builder = ["host", "device"][case % 2]
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md, "builder": builder})
    try:
        g.Build(pos)
    except RuntimeError:  # Octree::Build's own invariant: the reference prints and exits there
        rejected += 1
        g.close()
        continue
    g.set_shading(None, matid, mats)
    img, st = g.render(rt.make_render_config(cam_pos=cpos, cam_up=[0.0, 1.0, 0.0], cam_dir=cdir, fov=fov, width=w, height=h, render_type=rtype,
                                             shading="flat", cull=cull, max_distance=40.0, max_depth=depth))
    ts = g.tree_stats()
    g.close()
    ref = ref_runner.run(pos, None, matid, mats, cpos, [0.0, 1.0, 0.0], cdir, fov, w, h, render_type=rtype, shading="flat", cull=cull,
                         max_distance=40.0, tri_per_bb=mb, ads_depth=md, max_depth=depth)
    q = (255.999 * img.astype(np.float64)).astype(np.int64).reshape(-1, 3)
    text = f"P3\n{w} {h}\n255\n" + "".join(f"{r} {gg} {b}\n" for r, gg, b in q.tolist())
    why = []
    if hashlib.md5(text.encode()).hexdigest() != ref["ppm_md5"]: why.append("image")
    if st.rays != ref["search_calls"]: why.append(f"rays {st.rays} != {ref['search_calls']}")
    if (ts["nodes"], ts["leaves"]) != (ref["nodes"], ref["leaves"]): why.append("tree shape")
    rays_total += st.rays
    if why:
        bad.append(seed0 + case)
        print(f"seed {seed0 + case} n_tri {n_tri} mb {mb} md {md} {rtype} cull {cull} depth {depth} {builder}: {why}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {len(bad)} differ, {rejected} rejected", flush=True)
print(f"reference_fuzz: {n_cases} scenes, {rays_total} rays, scenes that differ from the reference binary: {len(bad)} {bad}; scenes Octree::Build rejects: {rejected}")
sys.exit(1 if bad else 0)
