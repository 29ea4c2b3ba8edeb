#!/usr/bin/env python3
"""Octree::Build time of the two builders on the 1,310,754-triangle synthetic scene (and the two shipped scenes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402
from raytracer_amd.synthetic import cornell_with_icosphere  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
box = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_original.npz"))
sphere = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_sphere.npz"))
out = {}
for name, pos in (("cornell_original", box.pos), ("cornell_sphere", sphere.pos), ("synthetic_ico6", cornell_with_icosphere(box, 6).pos),
                  ("synthetic_ico8", cornell_with_icosphere(box, 8).pos)):
    row = {"triangles": int(pos.shape[0])}
    for builder in ("host", "device"):
        best_wall, best_core = 1e9, 1e9
        for rep in range(3):
            g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10, "builder": builder})
            t0 = time.perf_counter()
            g.Build(pos)
            best_wall = min(best_wall, time.perf_counter() - t0)
            best_core = min(best_core, g.tree_stats()["build_seconds"])
            g.close()
        row[builder] = {"rt_ads_build_wall_s": round(best_wall, 5), "build_seconds": round(best_core, 5)}
    out[name] = row
print(json.dumps(out, indent=1))
