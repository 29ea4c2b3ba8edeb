"""ctypes binding of raytracer_amd/lib/libraytracer_amd.so (the C ABI in include/raytracer_amd.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing any entry
point raises.  PyTorch is not needed here; callers that hold device tensors pass raw pointers.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH selects another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(HERE, "lib", "libraytracer_amd.so")

RT_MEM_HOST, RT_MEM_DEVICE = 0, 1
RT_SEARCH_NOALLOC_MAX_RAYS, RT_SEARCH_NOALLOC_SCRATCH_BYTES = 4096, 16 << 20
RT_ADS_NONE, RT_ADS_OCTREE, RT_ADS_OCTREE_PARAMETRIC, RT_ADS_OCTREE_GPU = 0, 1, 2, 3
RENDER_TYPES = {"distance": 0, "diffusion": 1, "phong": 2, "blinn_phong": 3}
SHADING_TYPES = {"flat": 0, "smooth": 1}
RT_ASPECT_COMPAT, RT_ASPECT_TRUE = 0, 1


class AdsConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_triangles_per_bb", C.c_uint32),
                ("max_depth", C.c_uint32), ("builder", C.c_uint32)]


BUILDERS = {"auto": 0, "host": 1, "device": 2}


class TreeStats(C.Structure):
    _fields_ = [("max_depth", C.c_uint64), ("nodes", C.c_uint64), ("leaves", C.c_uint64),
                ("max_triangles_in_leaf", C.c_uint64), ("avg_leaf_depth", C.c_float),
                ("avg_triangles_in_leaf", C.c_float), ("leaf_refs", C.c_uint64), ("internal_nodes", C.c_uint64),
                ("build_seconds", C.c_double), ("device_bytes", C.c_uint64)]


class QueryStats(C.Structure):
    _fields_ = [("search_calls", C.c_uint64), ("closest_rays", C.c_uint64), ("any_rays", C.c_uint64),
                ("search_node_count", C.c_uint64), ("search_return_count", C.c_uint64), ("search_leaves_visited", C.c_uint64),
                ("ray_triangle_tests", C.c_uint64), ("slab_tests", C.c_uint64), ("counted_rays", C.c_uint64),
                ("search_seconds", C.c_double), ("slab_tests_any", C.c_uint64), ("ray_triangle_tests_any", C.c_uint64),
                ("counted_rays_any", C.c_uint64)]


class RenderConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("cam_pos", C.c_float * 3), ("cam_up", C.c_float * 3), ("cam_dir", C.c_float * 3),
        ("fov", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32), ("samples_per_triangle", C.c_int32),
        ("render_type", C.c_int32), ("shading_type", C.c_int32), ("cull_backfaces", C.c_int32),
        ("max_distance", C.c_float), ("background", C.c_float * 3),
        ("aspect_mode", C.c_int32), ("seed", C.c_uint32),
        ("band_rank", C.c_int32), ("band_world", C.c_int32),
    ]


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("rays", C.c_uint64), ("levels", C.c_uint32),
                ("ms_closest", C.c_float), ("ms_shadow", C.c_float), ("ms_shade", C.c_float), ("ms_total", C.c_float),
                ("launches_closest", C.c_uint32), ("launches_shadow", C.c_uint32), ("launches_shade", C.c_uint32),
                ("timed_closest_rays", C.c_uint64), ("timed_shadow_rays", C.c_uint64),
                ("ms_tail", C.c_float), ("launches_tail", C.c_uint32), ("shadow_rays_not_traced", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/raytracer_amd.h declares
EXPORTS = [
    "rt_ads_kind_from_name", "rt_ads_create", "rt_ads_destroy", "rt_ads_build", "rt_ads_export_tree", "rt_ads_tree_forms", "rt_ads_light_forms", "rt_ads_hit_lists", "rt_ads_tree_stats",
    "rt_ads_query_stats", "rt_ads_reset_query_stats", "rt_ads_collect_search_stats", "rt_ads_search", "rt_ads_search_noalloc", "rt_ads_trace_closest", "rt_ads_trace_any",
    "rt_scene_set_shading", "rt_render", "rt_render_submit", "rt_render_wait", "rt_ads_set_render_streams", "rt_render_primary_hits", "rt_rng_u31", "rt_last_error", "rt_version",
    "rt_host_tree_build", "rt_host_tree_destroy", "rt_host_tree_sizes", "rt_host_tree_export",
    "rt_host_tree_export_flat", "rt_host_tree_export_clustered", "rt_host_tree_stats",
    "rt_debug_triangle_tests", "rt_scene_load_obj", "rt_scene_data_free", "rt_scene_data_sizes", "rt_scene_data_export", "rt_write_ppm",
]

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"raytracer_amd error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                          "(there is no CPU fallback for the query path)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    L.rt_ads_kind_from_name.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.rt_ads_create.argtypes = [C.POINTER(AdsConfig), C.POINTER(vp)]
    L.rt_ads_destroy.argtypes = [vp]
    L.rt_ads_destroy.restype = None
    L.rt_ads_build.argtypes = [vp, vp, C.c_uint64]
    L.rt_ads_export_tree.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.rt_ads_tree_forms.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.rt_ads_light_forms.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rt_ads_hit_lists.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rt_ads_tree_stats.argtypes = [vp, C.POINTER(TreeStats)]
    L.rt_ads_query_stats.argtypes = [vp, C.POINTER(QueryStats)]
    L.rt_ads_reset_query_stats.argtypes = [vp]
    L.rt_ads_collect_search_stats.argtypes = [vp, C.c_int]
    L.rt_ads_search.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rt_ads_search_noalloc.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]
    L.rt_ads_trace_closest.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp]
    L.rt_ads_trace_any.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp]
    L.rt_scene_set_shading.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.rt_render.argtypes = [vp, C.POINTER(RenderConfig), vp, C.c_int, C.POINTER(RenderStats), vp]
    L.rt_render_submit.argtypes = [vp, C.POINTER(RenderConfig), vp, vp, C.POINTER(C.c_int)]
    L.rt_render_wait.argtypes = [vp, C.c_int, C.POINTER(RenderStats)]
    L.rt_render_primary_hits.argtypes = [vp, C.POINTER(RenderConfig), vp, vp, C.c_int, vp]
    L.rt_ads_set_render_streams.argtypes = [vp, C.c_int]
    L.rt_debug_triangle_tests.argtypes = [C.c_int, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_int, vp]
    L.rt_host_tree_build.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.rt_host_tree_destroy.argtypes = [vp]
    L.rt_host_tree_destroy.restype = None
    L.rt_host_tree_sizes.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4 + [C.POINTER(C.c_uint32)]
    L.rt_host_tree_export.argtypes = [vp] * 9
    L.rt_host_tree_export_flat.argtypes = [vp, vp, vp, vp]
    L.rt_host_tree_export_clustered.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rt_host_tree_stats.argtypes = [vp, C.POINTER(TreeStats)]
    L.rt_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rt_scene_data_free.argtypes = [vp]
    L.rt_scene_data_free.restype = None
    L.rt_scene_data_sizes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.rt_scene_data_export.argtypes = [vp, vp, vp, vp, vp]
    L.rt_write_ppm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp]
    L.rt_rng_u31.restype = C.c_uint32
    L.rt_rng_u31.argtypes = [C.c_uint32] * 7
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode())
