"""Host-side mirror of the reference's Renderer for the GPU backend.

`Renderer(config)` takes the "renderer" object of the reference's config.json (same keys, same error messages,
src/Renderer/renderer.cc:337-379 of the reference); `RenderScene(scene, camera)` runs the frame on the device through
the C ABI and writes the P3 PPM named by config["output"]["filename"] (renderer.cc:25-67, 381-390).
`load_scene` is the counterpart of LoadScene (src/ObjLoader/obj_loader.cc:130-212).
"""
import ctypes as C
import sys
import time

import numpy as np

from . import _capi
from .ads import GpuOctree, ads_kind_from_name, make_render_config
from .scene import Scene

GPU_ADS_NAME = "octree_gpu"


def load_scene(loader_config, scene_config):
    """obj_loader.cc:130-212: reads basepath + source_file; keys "basepath", "triangulate", "source_file"."""
    basepath = str(loader_config["basepath"])
    filename = basepath + str(scene_config["source_file"])
    h = C.c_void_p()
    _capi.check(_capi.lib().rt_scene_load_obj(filename.encode(), basepath.encode(), int(bool(loader_config["triangulate"])), C.byref(h)))
    try:
        n, m, hn = C.c_uint64(), C.c_uint32(), C.c_int()
        _capi.check(_capi.lib().rt_scene_data_sizes(h, C.byref(n), C.byref(m), C.byref(hn)))
        pos = np.zeros((n.value, 3, 3), np.float32)
        nrm = np.zeros((n.value, 3, 3), np.float32)
        mid = np.zeros(n.value, np.int32)
        mats = np.zeros((m.value, 18), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _capi.check(_capi.lib().rt_scene_data_export(h, p(pos), p(nrm), p(mid), p(mats)))
    finally:
        _capi.lib().rt_scene_data_free(h)
    if not hn.value:
        print("No normals found in the model, only flat shading available", file=sys.stderr)  # obj_loader.cc:161-163
    return Scene(pos, nrm, mid, mats, has_normals=bool(hn.value))


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    _capi.check(_capi.lib().rt_write_ppm(str(path).encode(), w, h, rgb.ctypes.data_as(C.c_void_p)))


class Camera:
    """src/camera.h:24-34."""

    def __init__(self, config):
        self.pos = [float(x) for x in config["pos"]]
        self.up = [float(x) for x in config["up"]]
        self.dir = [float(x) for x in config["dir"]]
        self.fov = float(config["fov"])
        self.width = int(config["width"])
        self.height = int(config["height"])
        self.samples_per_pixel = int(config["samples_per_pixel"])


class Renderer:
    def __init__(self, config, seed=42, device=0, aspect_mode=_capi.RT_ASPECT_COMPAT, collect_stats=False):
        self.config_ = config
        self.collect_stats = bool(collect_stats)  # the reference-defined Search / ray-triangle counters (slower)
        self.seed, self.device, self.aspect_mode = int(seed) & 0xffffffff, device, aspect_mode
        # renderer.cc:339-343 (a missing key raises KeyError like json::at throws)
        self.background_color_ = [float(config["background_color"][i]) for i in range(3)]
        self.max_depth_ = int(config["max_depth"])
        self.samples_per_triangle_ = int(config["samples_per_triangle"])
        self.cull_backfaces_ = bool(config["cull_backfaces"])
        self.max_distance_ = float(config["max_distance"])
        if config["render_type"] not in _capi.RENDER_TYPES:
            raise RuntimeError("Invalid render type")  # renderer.cc:355
        self.render_type_ = config["render_type"]
        if config["shading_type"] not in _capi.SHADING_TYPES:
            raise RuntimeError("Invalid shading type")  # renderer.cc:364
        self.shading_type_ = config["shading_type"]
        ads_cfg = config["acceleratied_data_structure"]  # the reference's spelling
        self.acceleration_data_structure_ = ads_kind_from_name(ads_cfg["name"])  # "Invalid data structure" otherwise
        self.stats = None

    def setup_ads(self):
        """renderer.cc:69-80.  This package only carries the GPU backend."""
        if self.acceleration_data_structure_ != _capi.RT_ADS_OCTREE_GPU:
            raise RuntimeError(f'only "{GPU_ADS_NAME}" is implemented by this backend; the other data structures are the reference\'s own')
        return GpuOctree(self.config_["acceleratied_data_structure"], device=self.device)

    def RenderScene(self, scene, camera, log=sys.stderr):
        t0 = time.time()
        ads = self.setup_ads()
        ads.Build(scene.pos)
        ads.set_shading(scene.norm, scene.matid, scene.materials)
        ads.collect_search_stats(self.collect_stats)
        cfg = make_render_config(camera.pos, camera.up, camera.dir, camera.fov, camera.width, camera.height,
                                 spp=camera.samples_per_pixel, max_depth=self.max_depth_, spt=self.samples_per_triangle_,
                                 render_type=self.render_type_, shading=self.shading_type_, cull=self.cull_backfaces_,
                                 max_distance=self.max_distance_, background=self.background_color_,
                                 aspect_mode=self.aspect_mode, seed=self.seed)
        img, st = ads.render(cfg)
        self.stats = st
        if log is not None:
            ads.PrintStats(log)
            log.write(f"Rendering time: {time.time() - t0:g} s\n")
            q = ads.query_stats()
            if q["counted_rays"]:
                log.write(f"Ray-triangle collision count: {q['ray_triangle_tests']}\n")  # renderer.cc:60
        write_ppm(self.config_["output"]["filename"], img)  # renderer.cc:66, 381-390
        return img
