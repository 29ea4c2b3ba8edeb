#!/usr/bin/env python3
"""Throughput of one rank's share of a K-way band split with D frames in flight (rt_render_submit / rt_render_wait),
rendered on this GPU, no gather: ms per frame-share = wall time / frames.  D = 0 is the blocking rt_render.
usage: inflight_time.py [workload]   (INFLIGHT_WORLDS=1,8  INFLIGHT_DEPTHS=0,1,2,3,4,6,8  INFLIGHT_FRAMES=40)
Also checks that a frame rendered in flight equals the blocking call's bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (first: it sets the number of hardware queues, which is read when HIP starts)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
wl = bench.WORKLOADS[name]
scene = bench.load_scene(wl, Scene)
ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
ads.Build(scene.pos)
ads.set_shading(scene.norm, scene.matid, scene.materials)
W, H = wl["width"], wl["height"]
kw = bench.render_kwargs(wl)
worlds = [int(x) for x in os.environ.get("INFLIGHT_WORLDS", "1,8").split(",")]
depths = [int(x) for x in os.environ.get("INFLIGHT_DEPTHS", "0,1,2,3,4,6,8").split(",")]
n = int(os.environ.get("INFLIGHT_FRAMES", "40"))
frames = [torch.zeros((H + 64, W, 3), dtype=torch.float32, device="cuda") for _ in range(max(depths + [1]))]
stream = torch.cuda.current_stream().cuda_stream
for world in worlds:
    cfg = rt.make_render_config(band_rank=world - 1, band_world=world, **kw)
    ref = torch.zeros_like(frames[0])
    ads.render(cfg, out_ptr=ref.data_ptr())
    torch.cuda.synchronize()
    for depth in depths:
        def run(count):
            if depth == 0:
                for _ in range(count):
                    ads.render(cfg, out_ptr=frames[0].data_ptr())
                return
            tickets = []
            for i in range(count):
                if len(tickets) == depth:
                    ads.render_wait(tickets.pop(0))
                tickets.append(ads.render_submit(cfg, frames[i % depth].data_ptr(), stream))
            for t in tickets:
                ads.render_wait(t)
        for f in frames:
            f.zero_()
        run(max(3, 2 * depth))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        same = all(torch.equal(f, ref) for f in frames[:max(depth, 1)])
        print(f"{name} world {world} depth {depth}: {ms:.3f} ms per frame-share, images equal the blocking call's: {same}", flush=True)
