"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference's
ray-scene query path.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product
package.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

RENDER_TYPES = {"distance": 0, "diffusion": 1, "phong": 2, "blinn_phong": 3}
SHADING_TYPES = {"flat": 0, "smooth": 1}
ADS_TYPES = {"none": 0, "octree": 1}
RNG_LEGACY, RNG_COUNTER = 0, 1
ASPECT_COMPAT, ASPECT_TRUE = 0, 1


class RenderCfg(C.Structure):
    _fields_ = [
        ("cam_pos", C.c_float * 3), ("cam_up", C.c_float * 3), ("cam_dir", C.c_float * 3),
        ("fov", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32), ("samples_per_triangle", C.c_int32),
        ("render_type", C.c_int32), ("shading_type", C.c_int32), ("cull_backfaces", C.c_int32),
        ("max_distance", C.c_float), ("background", C.c_float * 3),
        ("ads", C.c_int32), ("rng_mode", C.c_int32), ("aspect_mode", C.c_int32), ("seed", C.c_uint32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "search_calls", "search_nodes", "search_leaves", "search_returned", "tri_tests", "slab_tests",
        "primary_rays", "secondary_rays", "shadow_rays", "slab_tests_any", "tri_tests_any")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class TreeStats(C.Structure):
    _fields_ = [("max_depth", C.c_uint64), ("nodes", C.c_uint64), ("leaves", C.c_uint64),
                ("max_tris_in_leaf", C.c_uint64), ("avg_leaf_depth", C.c_float), ("avg_tris_in_leaf", C.c_float),
                ("leaf_refs", C.c_uint64)]


def build_lib():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build_lib()
    L = C.CDLL(LIB_PATH)
    vp, f32p, i32p, u32p, u8p, u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), \
        C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.orc_last_error.restype = C.c_char_p
    L.orc_scene_load_obj.restype = vp
    L.orc_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.orc_scene_from_arrays.restype = vp
    L.orc_scene_from_arrays.argtypes = [C.c_uint64, vp, vp, vp, C.c_uint32, vp]
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_num_triangles.restype = C.c_uint64
    L.orc_scene_num_triangles.argtypes = [vp]
    L.orc_scene_num_materials.restype = C.c_uint32
    L.orc_scene_num_materials.argtypes = [vp]
    L.orc_scene_num_lights.restype = C.c_uint32
    L.orc_scene_num_lights.argtypes = [vp]
    L.orc_scene_has_normals.argtypes = [vp]
    L.orc_scene_export.argtypes = [vp] + [vp] * 6
    L.orc_octree_build.restype = vp
    L.orc_octree_build.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.orc_octree_free.argtypes = [vp]
    L.orc_octree_stats.argtypes = [vp, C.POINTER(TreeStats)]
    L.orc_octree_num_nodes.restype = C.c_uint64
    L.orc_octree_num_nodes.argtypes = [vp]
    L.orc_octree_pool_size.restype = C.c_uint64
    L.orc_octree_pool_size.argtypes = [vp]
    L.orc_octree_export.argtypes = [vp] + [vp] * 7
    L.orc_octree_export_pool.argtypes = [vp, vp]
    L.orc_search.restype = C.c_uint64
    L.orc_search.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(Stats)]
    L.orc_trace_closest.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_int, vp, vp, C.POINTER(Stats)]
    L.orc_trace_any.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_int, vp, C.POINTER(Stats)]
    L.orc_ray_triangle.restype = C.c_float
    L.orc_ray_triangle.argtypes = [vp, vp, vp, C.c_int]
    L.orc_ray_aabb.argtypes = [vp, vp, vp, vp]
    L.orc_tri_aabb.argtypes = [vp, vp, vp, vp]
    L.orc_generate_rays.argtypes = [C.POINTER(RenderCfg), vp]
    L.orc_camera_frame.argtypes = [C.POINTER(RenderCfg), vp, vp, vp]
    L.orc_rng_u31.restype = C.c_uint32
    L.orc_rng_u31.argtypes = [C.c_uint32] * 7
    L.orc_render.argtypes = [vp, vp, C.POINTER(RenderCfg), vp, vp, vp, C.POINTER(Stats)]
    L.orc_ppm_text.restype = C.c_uint64
    L.orc_ppm_text.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_uint64]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """Triangle soup + materials as the reference's loader would hand them to Ads::Build."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle scene: " + lib().orc_last_error().decode())
        self.h = handle
        L = lib()
        self.n_tri = int(L.orc_scene_num_triangles(handle))
        self.n_mat = int(L.orc_scene_num_materials(handle))
        self.n_light = int(L.orc_scene_num_lights(handle))
        self.has_normals = bool(L.orc_scene_has_normals(handle))
        self.pos = np.zeros((self.n_tri, 3, 3), np.float32)
        self.norm = np.zeros((self.n_tri, 3, 3), np.float32)
        self.matid = np.zeros(self.n_tri, np.int32)
        self.materials = np.zeros((self.n_mat, 18), np.float32)
        self.face_normal = np.zeros((self.n_tri, 3), np.float32)
        self.light_tri = np.zeros(self.n_light, np.int32)
        L.orc_scene_export(handle, _p(self.pos), _p(self.norm), _p(self.matid), _p(self.materials), _p(self.face_normal),
                           _p(self.light_tri))

    @classmethod
    def from_obj(cls, obj_path, basepath, triangulate=True):
        return cls(lib().orc_scene_load_obj(obj_path.encode(), basepath.encode(), int(triangulate)))

    @classmethod
    def from_arrays(cls, pos, norm, matid, materials):
        pos = _f32(pos).reshape(-1, 9)
        norm_c = None if norm is None else _f32(norm).reshape(-1, 9)
        matid = np.ascontiguousarray(matid, dtype=np.int32)
        materials = _f32(materials).reshape(-1, 18)
        return cls(lib().orc_scene_from_arrays(pos.shape[0], _p(pos), _p(norm_c), _p(matid), materials.shape[0], _p(materials)))

    @classmethod
    def from_npz(cls, path):
        with np.load(path, allow_pickle=False) as z:
            norm = z["norm"] if bool(z["has_normals"]) else None
            return cls.from_arrays(z["pos"], norm, z["matid"], z["materials"])

    def save_npz(self, path):
        np.savez_compressed(path, pos=self.pos, norm=self.norm, matid=self.matid, materials=self.materials,
                            has_normals=np.array(self.has_normals))

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass


class Octree:
    def __init__(self, scene, max_triangles_per_bb=16, max_depth=10):
        self.scene = scene
        self.h = lib().orc_octree_build(scene.h, max_triangles_per_bb, max_depth)
        if not self.h:
            raise RuntimeError("oracle octree: " + lib().orc_last_error().decode())

    def stats(self):
        st = TreeStats()
        lib().orc_octree_stats(self.h, C.byref(st))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def export(self):
        L = lib()
        n = int(L.orc_octree_num_nodes(self.h))
        out = dict(is_leaf=np.zeros(n, np.uint8), depth=np.zeros(n, np.uint32), bmin=np.zeros((n, 3), np.float32),
                   bmax=np.zeros((n, 3), np.float32), child=np.zeros((n, 8), np.int32), tri_begin=np.zeros(n, np.uint64),
                   tri_count=np.zeros(n, np.uint32))
        L.orc_octree_export(self.h, _p(out["is_leaf"]), _p(out["depth"]), _p(out["bmin"]), _p(out["bmax"]), _p(out["child"]),
                            _p(out["tri_begin"]), _p(out["tri_count"]))
        pool = np.zeros(int(L.orc_octree_pool_size(self.h)), np.uint32)
        L.orc_octree_export_pool(self.h, _p(pool))
        out["pool"] = pool
        return out

    def search(self, orig, direction, cap=1 << 20, stats=None):
        o, d = _f32(orig), _f32(direction)
        buf = np.zeros(cap, np.uint32)
        n = int(lib().orc_search(self.h, _p(o), _p(d), _p(buf), cap, C.byref(stats) if stats is not None else None))
        if n > cap:  # (a ray with NaN components passes every box: its list is every leaf's list) -- once more with room for all
            buf = np.zeros(n, np.uint32)
            n = int(lib().orc_search(self.h, _p(o), _p(d), _p(buf), n, None))
        return buf[:n].copy()

    def trace_closest(self, orig, direction, cull=True, stats=None):
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        lib().orc_trace_closest(self.scene.h, self.h, n, _p(o), _p(d), int(cull), _p(t), _p(tri),
                                C.byref(stats) if stats is not None else None)
        return t, tri

    def trace_any(self, point, light, cull=True, stats=None):
        p, l = _f32(point).reshape(-1, 3), _f32(light).reshape(-1, 3)
        n = p.shape[0]
        m = np.zeros(n, np.uint8)
        lib().orc_trace_any(self.scene.h, self.h, n, _p(p), _p(l), int(cull), _p(m), C.byref(stats) if stats is not None else None)
        return m

    def __del__(self):
        try:
            lib().orc_octree_free(self.h)
        except Exception:
            pass


def make_cfg(cam_pos, cam_up, cam_dir, fov, width, height, spp=1, max_depth=10, spt=1, render_type="distance",
             shading="flat", cull=True, max_distance=2000.0, background=(0.01, 0.01, 0.01), ads="octree",
             rng_mode=RNG_COUNTER, aspect_mode=ASPECT_COMPAT, seed=42, rows=None):
    c = RenderCfg()
    c.cam_pos[:] = cam_pos
    c.cam_up[:] = cam_up
    c.cam_dir[:] = cam_dir
    c.fov = fov
    c.width, c.height, c.samples_per_pixel = width, height, spp
    c.max_depth, c.samples_per_triangle = max_depth, spt
    c.render_type = RENDER_TYPES[render_type]
    c.shading_type = SHADING_TYPES[shading]
    c.cull_backfaces = int(cull)
    c.max_distance = max_distance
    c.background[:] = background
    c.ads = ADS_TYPES[ads]
    c.rng_mode, c.aspect_mode, c.seed = rng_mode, aspect_mode, seed
    if rows is not None:
        c.row_begin, c.row_end = rows
    return c


def render(scene, octree, cfg, want_hits=True):
    n = cfg.width * cfg.height
    rgb = np.zeros((cfg.height, cfg.width, 3), np.float32)
    tri = np.full(n, -2, np.int32) if want_hits else None
    t = np.zeros(n, np.float32) if want_hits else None
    st = Stats()
    rc = lib().orc_render(scene.h, octree.h, C.byref(cfg), _p(rgb), _p(tri), _p(t), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle render: " + lib().orc_last_error().decode())
    return rgb, tri, t, st


def ppm_bytes(rgb):
    h, w, _ = rgb.shape
    rgb = _f32(rgb)
    n = int(lib().orc_ppm_text(w, h, _p(rgb), None, 0))
    buf = C.create_string_buffer(n)
    lib().orc_ppm_text(w, h, _p(rgb), C.cast(buf, C.c_void_p), n)
    return buf.raw[:n]


def ppm_md5(rgb):
    return hashlib.md5(ppm_bytes(rgb)).hexdigest()


def camera_frame(cfg):
    p00, qw, qh = (np.zeros(3, np.float32) for _ in range(3))
    lib().orc_camera_frame(C.byref(cfg), _p(p00), _p(qw), _p(qh))
    return p00, qw, qh


def generate_rays(cfg):
    out = np.zeros((cfg.height, cfg.width, 3), np.float32)
    lib().orc_generate_rays(C.byref(cfg), _p(out))
    return out
