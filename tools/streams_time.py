#!/usr/bin/env python3
"""Frame time of a workload when the frame is rendered as K concurrent sub-frames (bands dealt round-robin to K
handles, each on its own stream and host thread): does overlapping the latency-bound small levels pay?
usage: tools/streams_time.py [workload] (env STREAMS=1,2,4  BAND_WORLD=1  FRAMES=10)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_ico8_phong_4k"
wl = bench.WORKLOADS[name]
scene = bench.load_scene(wl, Scene)
W, H = wl["width"], wl["height"]
kw = bench.render_kwargs(wl)
frame = torch.zeros((H + 64, W, 3), dtype=torch.float32, device="cuda")
world = int(os.environ.get("BAND_WORLD", "1"))
frames = int(os.environ.get("FRAMES", "10"))
for K in [int(x) for x in os.environ.get("STREAMS", "1,2,4").split(",")]:
    handles = []
    for k in range(K):
        a = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
        a.Build(scene.pos)
        a.set_shading(scene.norm, scene.matid, scene.materials)
        handles.append(a)
    streams = [torch.cuda.Stream() for _ in range(K)]
    cfgs = [rt.make_render_config(band_rank=k, band_world=world * K, **kw) for k in range(K)]
    rays = [0] * K

    def work(k, n):
        for _ in range(n):
            _, st = handles[k].render(cfgs[k], out_ptr=frame.data_ptr(), stream=streams[k].cuda_stream)
            rays[k] = st.rays

    def run(n):
        ts = [threading.Thread(target=work, args=(k, n)) for k in range(K)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(frames)
    ms = (time.perf_counter() - t0) / frames * 1e3
    print(f"{name} world {world} streams {K}: {ms:.3f} ms/frame, rays {sum(rays)}")
    for a in handles:
        a.close()
