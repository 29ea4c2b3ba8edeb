#!/usr/bin/env python3
"""Per-rank frame time of the default workload when the image is split over K ranks (one rank's share rendered on
this GPU; no gather): shows the fixed per-frame cost that bounds strong scaling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD]
scene = bench.load_scene(wl, Scene)
ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
ads.Build(scene.pos)
ads.set_shading(scene.norm, scene.matid, scene.materials)
W, H = wl["width"], wl["height"]
kw = bench.render_kwargs(wl)
frame = torch.zeros((H + 64, W, 3), dtype=torch.float32, device="cuda")
worlds = [int(x) for x in os.environ.get('BAND_WORLDS', '1,2,4,8').split(',')]
for world in worlds:
    for rank in (0, world - 1):
        cfg = rt.make_render_config(band_rank=rank, band_world=world, **kw)
        for _ in range(3):
            ads.render(cfg, out_ptr=frame.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get("BAND_FRAMES", "20"))
        for _ in range(n):
            _, st = ads.render(cfg, out_ptr=frame.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD} world {world} rank {rank}: {ms:.3f} ms/frame, rays {st.rays}, kernels {st.ms_closest + st.ms_shadow + st.ms_shade:.3f} ms, total(ev) {st.ms_total:.3f}")
