#!/usr/bin/env python3
"""Randomised check that how a frame is scheduled does not show in it (GPU box): random scenes and configurations; the
blocking one-piece frame is compared, bit for bit and in its ray counts, with the same frame rendered as a random K-way
band split, with a random number of concurrent sub-frames, with and without the fused tail, and as frames in flight that
are submitted and retired in random order (every band set of the split a frame in flight of its own).
usage: schedule_fuzz.py [N=100] [first seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (first: sets the hardware queue count)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402
from raytracer_amd.synthetic import cornell_with_icosphere  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
golden = os.path.join(ROOT, "tests", "golden", "scenes")
sphere = Scene.load_npz(os.path.join(golden, "cornell_sphere.npz"))
original = Scene.load_npz(os.path.join(golden, "cornell_original.npz"))
ico = {k: cornell_with_icosphere(original, k) for k in (5, 6)}
CAM_O = bench.CAM_ORIGINAL
CAM_S = bench.CAM_SPHERE
bad = []
frames_total = 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    which = case % 4
    if which == 0:
        sc, cam, smooth = original, CAM_O, False
    elif which == 1:
        sc, cam, smooth = sphere, CAM_S, True
    else:
        sc, cam, smooth = ico[5 + case % 2], CAM_O, False
    kw = dict(fov=float(rs.uniform(0.4, 0.8)), width=int(rs.choice([64, 100, 160])), height=int(rs.choice([40, 72, 130])), spp=int(rs.choice([1, 1, 3])),
              spt=int(rs.choice([1, 2, 7, 16])), render_type=str(rs.choice(["diffusion", "phong", "blinn_phong"])), shading="smooth" if smooth and rs.randint(0, 2) else "flat",
              cull=bool(rs.randint(0, 4) != 0), max_depth=int(rs.randint(0, 7)), aspect_mode=int(rs.randint(0, 2)), seed=int(rs.randint(1, 1 << 30)), **cam)
    for k in ("RT_TAIL_MAX_RAYS",):
        os.environ.pop(k, None)
    if rs.randint(0, 3) == 0:
        os.environ["RT_TAIL_MAX_RAYS"] = str(int(rs.choice([0, 500, 1 << 30])))
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(sc.pos)
    g.set_shading(sc.norm if sc.has_normals else None, sc.matid, sc.materials)
    H, W = kw["height"], kw["width"]
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    _, st0 = g.render(rt.make_render_config(**kw), out_ptr=ref.data_ptr())
    counts0 = (st0.rays, st0.primary_rays, st0.secondary_rays, st0.shadow_rays)
    why = []
    # (a) band split with sub-frames, blocking
    world = int(rs.randint(1, 9))
    g.set_render_streams(int(rs.randint(1, 9)))
    out = torch.zeros_like(ref)
    tot = np.zeros(4, np.int64)
    for r in rs.permutation(world):
        _, st = g.render(rt.make_render_config(band_rank=int(r), band_world=world, **kw), out_ptr=out.data_ptr())
        tot += (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays)
    if not torch.equal(out, ref) or tuple(int(x) for x in tot) != counts0:
        why.append(f"bands {world}")
    g.set_render_streams(1)
    # (b) the band sets as frames in flight, retired in random order; whole frames beside them
    world = int(rs.randint(1, 7))
    out = torch.zeros_like(ref)
    whole = torch.zeros_like(ref)
    tickets = [g.render_submit(rt.make_render_config(band_rank=int(r), band_world=world, **kw), out.data_ptr()) for r in rs.permutation(world)]
    tickets.append(g.render_submit(rt.make_render_config(**kw), whole.data_ptr()))
    tot = np.zeros(4, np.int64)
    order = rs.permutation(len(tickets))
    for i in order:
        st = g.render_wait(tickets[i])
        if i < world:
            tot += (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays)
        elif (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays) != counts0:
            why.append("whole frame in flight: counts")
    if not torch.equal(out, ref) or tuple(int(x) for x in tot) != counts0:
        why.append(f"in flight {world}")
    if not torch.equal(whole, ref):
        why.append("whole frame in flight")
    frames_total += 3
    g.close()
    if why:
        bad.append(seed0 + case)
        print(f"seed {seed0 + case} {kw['render_type']} {W}x{H}: {why}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {len(bad)} differ", flush=True)
print(f"schedule_fuzz: {n_cases} scenes x (band split with sub-frames, band sets in flight, whole frame in flight): frames that differ from the blocking "
      f"one-piece frame: {len(bad)} {bad}")
sys.exit(1 if bad else 0)
