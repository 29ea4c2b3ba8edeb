#!/usr/bin/env python3
"""Where a wave of the per-lane traversal spends its cycles (descent / leaf work), from the counters of a library built
with -DRT_LANE_PROFILE (tools/lane_profile.sh builds it next to the product library and points RT_LIB_PATH at it)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _capi  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "synthetic_ico8_phong_4k"]
if wl["scene"].startswith("synthetic:"):
    from raytracer_amd.synthetic import cornell_with_icosphere
    box = Scene.load_npz(os.path.join(bench.ROOT, "tests", "golden", "scenes", "cornell_original.npz"))
    scene = cornell_with_icosphere(box, int(wl["scene"].split(":")[1]))
else:
    scene = Scene.load_npz(os.path.join(bench.ROOT, "tests", "golden", "scenes", wl["scene"] + ".npz"))
ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
ads.Build(scene.pos)
ads.set_shading(scene.norm, scene.matid, scene.materials)
W, H = wl["width"], wl["height"]
kw = dict(fov=0.6, width=W, height=H, spp=wl["spp"], spt=wl["spt"], render_type=wl["render_type"], shading=wl["shading"],
          cull=wl["cull"], max_depth=wl["max_depth"], seed=42, aspect_mode=1 if wl.get("aspect") == "true" else 0, **wl["cam"])
frame = torch.zeros((H + 64, W, 3), dtype=torch.float32, device="cuda")
cfg = rt.make_render_config(**kw)
lib = _capi.lib()
out = (ctypes.c_ulonglong * 64)()
ads.render(cfg, out_ptr=frame.data_ptr())
torch.cuda.synchronize()
assert lib.rt_debug_lane_profile(out) == 0
ads.render(cfg, out_ptr=frame.data_ptr())
torch.cuda.synchronize()
assert lib.rt_debug_lane_profile(out) == 0
for k, name in ((0, "closest"), (1, "any-hit")):
    a, b, steps, lanes, chunks, pairs, waves, rounds, npop, nint, nleaf, nhold, ndone = [int(out[k * 16 + i]) for i in range(13)]
    if waves == 0:
        continue
    print(f"{name}: per descent step, lanes that pop {npop / max(steps, 1):.1f}, enter a node {nint / max(steps, 1):.1f}, take a leaf {nleaf / max(steps, 1):.1f}, "
          f"hold and wait {nhold / max(steps, 1):.1f}, are finished {ndone / max(steps, 1):.1f}")
    print(f"{name}: waves {waves}  cycles/wave descent {a / waves:.0f} leaf {b / waves:.0f} | per wave: rounds {rounds / waves:.1f} "
          f"descent steps {steps / waves:.1f} (lanes stepping {lanes / max(steps, 1):.1f}) leaf chunks {chunks / waves:.1f} pairs {pairs / waves:.0f} "
          f"| cycles per descent step {a / max(steps, 1):.0f}, per leaf chunk {b / max(chunks, 1):.0f}, pairs per chunk {pairs / max(chunks, 1):.1f}")
for k, name in ((0, "closest"), (1, "any-hit")):
    a, b, steps, tasks, chunks, pairs, waves, phases = [int(out[48 + k * 8 + i]) for i in range(8)]
    if waves == 0:
        continue
    print(f"pool {name}: waves {waves}  cycles/wave node steps {a / waves:.0f} leaf {b / waves:.0f} | per wave: node steps {steps / waves:.1f} "
          f"(tasks per step {tasks / max(steps, 1):.1f}) leaf phases {phases / waves:.1f} chunks {chunks / waves:.1f} pairs {pairs / waves:.0f} "
          f"| cycles per node step {a / max(steps, 1):.0f}, per leaf chunk {b / max(chunks, 1):.0f}, pairs per chunk {pairs / max(chunks, 1):.1f}")
w = [int(out[32 + i]) for i in range(16)]
if w[0]:
    print(f"wave any-hit, plain walk: waves {w[0]}  per wave: node visits {w[1] / w[0]:.2f} leaf visits {w[2] / w[0]:.2f} triangle tests started {w[3] / w[0]:.2f}")
if w[8]:
    print(f"wave any-hit, clustered walk: waves {w[8]}  per wave: node visits {w[9] / w[8]:.2f} cluster visits {w[10] / w[8]:.2f} leaves reached {w[11] / w[8]:.2f} "
          f"list records scanned {w[12] / w[8]:.2f} (listed by no reached leaf {w[13] / w[8]:.2f}, needed by no live lane {w[14] / w[8]:.2f}) triangle tests started {w[15] / w[8]:.2f}")
