#!/usr/bin/env python3
"""Randomised parity check against the CPU restatement (GPU box): random triangle soups, random materials (reflection,
refraction, emissive triangles), random octree parameters, cameras, render types, culling, light-sample and pixel-sample
counts.  Per scene: primary hits (t and triangle bit for bit), ray counts, image within 1/255 (bit-exact for the render
types without area lights), on every other scene the Search / ray-triangle counters of the instrumented pass, in the default traversal forms and with RT_CLIP=0 / RT_TRAVERSAL forced on a share of the scenes.
usage: oracle_fuzz.py [N=200] [first seed=1]"""
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
emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
plain = [m for m in range(mats.shape[0]) if m not in emissive]
bad, skipped, rays_total = [], 0, 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_tri = int(np.exp(rs.uniform(np.log(4), np.log(6000))))
    mb, md = int(rs.choice([1, 2, 4, 8, 16, 64])), int(rs.randint(0, 9))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.03), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    if case % 4 == 0:
        pos[::3, :, int(rs.randint(0, 3))] = np.round(pos[::3, :, 0])  # axis-aligned, coplanar layers: exact ties
    if case % 7 == 0 and n_tri > 10:
        pos[5::9] = pos[4::9][: len(pos[5::9])]  # exact duplicates
    matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
    n_lights = int(rs.randint(0, 4))
    if n_lights:
        matid[rs.choice(n_tri, min(n_lights, n_tri), replace=False)] = emissive[0]
    rtype = str(rs.choice(["distance", "diffusion", "phong", "blinn_phong"]))
    kw = dict(fov=float(rs.uniform(0.4, 1.1)), width=int(rs.choice([24, 40, 64])), height=int(rs.choice([16, 40, 50])), spp=int(rs.choice([1, 1, 2])),
              spt=int(rs.choice([1, 2, 5, 16, 33])), render_type=rtype, shading="flat", cull=bool(rs.randint(0, 2)), max_depth=int(rs.randint(0, 5)),
              cam_pos=[float(x) for x in (rs.rand(3) * 4.0 - 2.0 + np.array([0.0, 0.0, -13.0]))], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0],
              max_distance=40.0, seed=int(rs.randint(1, 1 << 30)))
    aspect = int(rs.randint(0, 2))
    env = [{}, {"RT_CLIP": "0"}, {"RT_TRAVERSAL": "lane"}, {"RT_TRAVERSAL": "pool"}, {"RT_DISABLE_SMALL_TREE": "1"}, {"RT_BUILDER": "device"}][case % 6]
    for k in ("RT_CLIP", "RT_TRAVERSAL", "RT_DISABLE_SMALL_TREE", "RT_BUILDER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
    try:
        g.Build(pos)
    except RuntimeError as e:  # Octree::Build's own invariant, or a tree beyond the stated limits
        skipped += 1
        g.close()
        continue
    g.set_shading(None, matid, mats)
    cfg = rt.make_render_config(aspect_mode=aspect, **kw)
    with_stats = case % 2 == 1  # every other scene also runs the instrumented pass (the reference's own counters)
    g.collect_search_stats(with_stats)
    img, st = g.render(cfg)
    q = g.query_stats()
    g.collect_search_stats(False)
    t, tri = g.primary_hits(cfg)
    g.close()
    osc = orc.Scene.from_arrays(pos, None, matid, mats)
    ref, ref_tri, ref_t, ost = orc.render(osc, orc.Octree(osc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, aspect_mode=aspect, **kw))
    rays_total += st.rays
    why = []
    if not np.array_equal(tri, ref_tri): why.append("triangle")
    if not np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)): why.append("t")
    if st.rays != ost.search_calls: why.append(f"rays {st.rays} != {ost.search_calls}")
    if with_stats and (q["search_node_count"], q["search_leaves_visited"], q["search_return_count"], q["slab_tests"], q["ray_triangle_tests"], q["counted_rays"]) != \
            (ost.search_nodes, ost.search_leaves, ost.search_returned, ost.slab_tests, ost.tri_tests, ost.search_calls):
        why.append("counters")
    diff = float(np.nanmax(np.abs(img - ref))) if img.size else 0.0
    exact = rtype in ("distance", "diffusion")
    if (exact and not np.array_equal(img.view(np.uint32), ref.view(np.uint32))) or not (diff <= 1.0 / 255.0): why.append(f"image {diff:g}")
    if why:
        bad.append((seed0 + case, env, why))
        print(f"seed {seed0 + case} {env} n_tri {n_tri} mb {mb} md {md} {rtype}: {why}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {len(bad)} differ, {skipped} skipped", flush=True)
print(f"oracle_fuzz: {n_cases} scenes, {rays_total} rays, scenes that differ from the CPU restatement: {len(bad)} {[b[0] for b in bad]}; "
      f"scenes the octree build rejects: {skipped}")
sys.exit(1 if bad else 0)
