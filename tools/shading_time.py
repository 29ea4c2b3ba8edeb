import sys, time
sys.path.insert(0, '/root/repo')
import bench
import raytracer_amd as rt
from raytracer_amd.scene import Scene
wl = bench.WORKLOADS["synthetic_ico8_phong_4k"]
scene = bench.load_scene(wl, Scene)
ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
t0 = time.perf_counter(); ads.Build(scene.pos); t1 = time.perf_counter()
ads.set_shading(scene.norm, scene.matid, scene.materials); t2 = time.perf_counter()
print("Build call %.3f s (tree %.4f s), set_shading %.3f s, light forms %s" % (t1 - t0, ads.tree_stats()["build_seconds"], t2 - t1, ads.light_forms()))
