"""Child process of test_gpu_parity.py::test_task_pool_under_capacity_pressure: loads the library variant named by
RT_LIB_PATH (the task-pool form built with tiny LDS capacities), traces the rays of the input file in the pool form
and writes the results.  usage: pool_stress_child.py <in.npz> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402

inp = np.load(sys.argv[1])
g = rt.GpuOctree({"max_triangles_per_BB": int(inp["mb"]), "max_depth": int(inp["md"])})
g.Build(inp["pos"])
t, tri = g.trace_closest(inp["o"], inp["d"], True)
occ = g.trace_any(inp["P"], inp["L"], True)
np.savez(sys.argv[2], t=t, tri=tri, occ=occ)
