"""Host-side mirror of the reference's acceleration-data-structure interface.

`Ads` / `GpuOctree` follow the reference's `Ads` contract (src/ADS/ads.h:16-28 of the reference):
construct from the "acceleratied_data_structure" config object, `Build(triangles)`,
`Search(ray)`, `PrintStats(os)`.  `GpuOctree` is the new subclass; it is selected by the config
name "octree_gpu" (ADS_NAMES below extends renderer.cc:368-376 of the reference).  Besides the
single-ray `Search`, it offers the batched calls the renderer's hot loops map to.
"""
import ctypes as C

import numpy as np

from . import _capi


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ads_kind_from_name(name):
    """renderer.cc:368-376: unknown names raise RuntimeError("Invalid data structure")."""
    out = C.c_int()
    rc = _capi.lib().rt_ads_kind_from_name(str(name).encode(), C.byref(out))
    if rc != 0:
        raise RuntimeError(_capi.lib().rt_last_error().decode())
    return out.value


class Ads:
    """src/ADS/ads.h:16-28."""

    def __init__(self, config):
        self.config_ = config
        self.triangles_ = None

    def Build(self, triangles):
        self.triangles_ = triangles

    def Search(self, ray, first_hit=False):
        raise NotImplementedError

    def PrintStats(self, os):
        raise NotImplementedError


class GpuOctree(Ads):
    """The MI355X backend behind the reference's Ads interface.

    config: the "acceleratied_data_structure" object; reads "max_triangles_per_BB" and "max_depth"
    (octree.cc:268-275; a missing key raises KeyError like json::at) and the optional "builder"
    ("auto" | "host" | "device": where Octree::Build runs).  triangles for Build: array of shape (n, 3, 3)
    (vertex positions in the scene's triangle order)."""

    def __init__(self, config, device=0):
        super().__init__(config)
        self.max_triangles_per_BB_ = int(config["max_triangles_per_BB"])
        self.max_depth_ = int(config["max_depth"])
        builder = str(config.get("builder", "auto"))
        if builder not in _capi.BUILDERS:
            raise RuntimeError("Invalid builder")
        cfg = _capi.AdsConfig(C.sizeof(_capi.AdsConfig), int(device), self.max_triangles_per_BB_, self.max_depth_,
                              _capi.BUILDERS[builder])
        h = C.c_void_p()
        _capi.check(_capi.lib().rt_ads_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().rt_ads_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Ads interface -------------------------------------------------------------------------------------------
    def Build(self, triangles):
        pos = _f32(triangles).reshape(-1, 9)
        super().Build(pos.reshape(-1, 3, 3))
        _capi.check(_capi.lib().rt_ads_build(self._h, _p(pos), pos.shape[0]))

    def Search(self, ray, first_hit=False):
        """ray: (origin xyz, direction xyz).  Returns candidate triangle indices in the reference's order."""
        return self.search_batch(np.asarray(ray, np.float32).reshape(1, 6))[0]

    def PrintStats(self, os):
        """octree.cc:209-224, line for line.  The four Search counters after the call count are exact for Search()
        itself; for the batched trace / render calls they are collected while collect_search_stats(True) is on."""
        t, q = self.tree_stats(), self.query_stats()
        os.write("Octree stats: \n")
        os.write(f"  Max depth: {t['max_depth']}\n")
        os.write(f"  Nodes count: {t['nodes']}\n")
        os.write(f"  Leaf nodes count: {t['leaves']}\n")
        os.write(f"  Average depth of leaf nodes: {t['avg_leaf_depth']:g}\n")
        os.write(f"  Max triangles in leaf nodes: {t['max_triangles_in_leaf']}\n")
        os.write(f"  Average triangles in leaf nodes: {t['avg_triangles_in_leaf']:g}\n")
        os.write(f"  Search method call count: {q['search_calls']}\n")
        os.write(f"  Search node count: {q['search_node_count']}\n")
        os.write(f"  Search time: {q['search_seconds']:g} s\n")
        os.write(f"  Search return count: {q['search_return_count']}\n")
        avg = q['search_return_count'] / q['search_calls'] if q['search_calls'] else float('nan')
        os.write(f"  Average search return count: {avg:g}\n")
        os.write(f"  Search leaves visited: {q['search_leaves_visited']}\n")

    # ---- batched queries -----------------------------------------------------------------------------------------
    def search_batch(self, rays, cap=None):
        rays = _f32(rays).reshape(-1, 6)
        n = rays.shape[0]
        offsets = np.zeros(n + 1, np.uint64)
        total = C.c_uint64()
        cap = int(cap) if cap is not None else max(1024, 64 * n)
        while True:
            out = np.zeros(cap, np.uint32)
            rc = _capi.lib().rt_ads_search(self._h, _p(rays), n, _p(offsets), _p(out), cap, C.byref(total))
            if rc == -5 and total.value > cap:  # RT_ERR_CAPACITY: retry with the size the library asked for
                cap = int(total.value)
                continue
            _capi.check(rc)
            break
        return [out[int(offsets[i]):int(offsets[i + 1])].copy() for i in range(n)]

    def search_noalloc_device(self, rays_ptr, n, offsets_ptr, out_ptr, cap, total_ptr, stream=None):
        """rt_ads_search_noalloc: every buffer is device memory; asynchronous on `stream`; allocates nothing."""
        _capi.check(_capi.lib().rt_ads_search_noalloc(self._h, rays_ptr, n, offsets_ptr, out_ptr, cap, total_ptr, stream))

    def trace_closest(self, origins, dirs, cull_backfaces=True):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        _capi.check(_capi.lib().rt_ads_trace_closest(self._h, _p(o), _p(d), n, int(cull_backfaces), _capi.RT_MEM_HOST,
                                                     _p(t), _p(tri), None))
        return t, tri

    def trace_any(self, points, lights, cull_backfaces=True):
        p, l = _f32(points).reshape(-1, 3), _f32(lights).reshape(-1, 3)
        n = p.shape[0]
        m = np.zeros(n, np.uint8)
        _capi.check(_capi.lib().rt_ads_trace_any(self._h, _p(p), _p(l), n, int(cull_backfaces), _capi.RT_MEM_HOST, _p(m), None))
        return m

    def trace_closest_device(self, origins_ptr, dirs_ptr, n, out_t_ptr, out_tri_ptr, cull_backfaces=True, stream=None):
        _capi.check(_capi.lib().rt_ads_trace_closest(self._h, origins_ptr, dirs_ptr, n, int(cull_backfaces), _capi.RT_MEM_DEVICE,
                                                     out_t_ptr, out_tri_ptr, stream))

    def trace_any_device(self, points_ptr, lights_ptr, n, out_ptr, cull_backfaces=True, stream=None):
        _capi.check(_capi.lib().rt_ads_trace_any(self._h, points_ptr, lights_ptr, n, int(cull_backfaces), _capi.RT_MEM_DEVICE,
                                                 out_ptr, stream))

    # ---- shading inputs + whole-frame rendering ----------------------------------------------------------------------
    def set_shading(self, normals, material_ids, materials):
        nrm = None if normals is None else _f32(normals).reshape(-1, 9)
        mid = np.ascontiguousarray(material_ids, dtype=np.int32)
        mats = _f32(materials).reshape(-1, 18)
        _capi.check(_capi.lib().rt_scene_set_shading(self._h, _p(nrm), _p(mid), _p(mats), mats.shape[0]))

    def render(self, cfg, out=None, out_ptr=None, stream=None):
        """cfg: _capi.RenderConfig.  Renders into `out` (host ndarray H x W x 3, created if None) or into the
        device buffer at out_ptr.  Returns (out, stats)."""
        st = _capi.RenderStats()
        if out_ptr is not None:
            _capi.check(_capi.lib().rt_render(self._h, C.byref(cfg), out_ptr, _capi.RT_MEM_DEVICE, C.byref(st), stream))
            return None, st
        if out is None:
            out = np.zeros((cfg.height, cfg.width, 3), np.float32)
        _capi.check(_capi.lib().rt_render(self._h, C.byref(cfg), _p(out), _capi.RT_MEM_HOST, C.byref(st), stream))
        return out, st

    def render_submit(self, cfg, out_ptr, stream=None):
        """Queues a frame into the device buffer at out_ptr and returns a ticket at once (up to 8 frames in flight)."""
        ticket = C.c_int()
        _capi.check(_capi.lib().rt_render_submit(self._h, C.byref(cfg), out_ptr, stream, C.byref(ticket)))
        return ticket.value

    def render_wait(self, ticket):
        """Blocks until the frame of `ticket` is complete; returns its statistics."""
        st = _capi.RenderStats()
        _capi.check(_capi.lib().rt_render_wait(self._h, int(ticket), C.byref(st)))
        return st

    def set_render_streams(self, n):
        """render() calls render their bands as n concurrent sub-frames on streams of the library's own (same image)."""
        _capi.check(_capi.lib().rt_ads_set_render_streams(self._h, int(n)))

    def primary_hits(self, cfg):
        n = cfg.width * cfg.height
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        _capi.check(_capi.lib().rt_render_primary_hits(self._h, C.byref(cfg), _p(t), _p(tri), _capi.RT_MEM_HOST, None))
        return t, tri

    def export_tree(self):
        """The resident tree copied back: (node records, leaf slots, root_info, max_stack, built_on_device)."""
        nr, ns, ms, dev = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_int()
        root = (C.c_uint32 * 2)()
        _capi.check(_capi.lib().rt_ads_export_tree(self._h, None, None, root, C.byref(nr), C.byref(ns), C.byref(ms), C.byref(dev)))
        nodes = np.zeros(nr.value, HostTree.NODE_DT)
        slots = np.zeros(ns.value, HostTree.SLOT_DT)
        _capi.check(_capi.lib().rt_ads_export_tree(self._h, _p(nodes), _p(slots), root, None, None, None, None))
        return nodes, slots, (int(root[0]), int(root[1])), int(ms.value), bool(dev.value)

    def tree_forms(self):
        """(triangles of the small-tree form, clusters of the clustered form, hand-scheduled any-hit test in use);
        0 = that form is not in use on this tree."""
        a, b, c = C.c_uint64(), C.c_uint32(), C.c_int()
        _capi.check(_capi.lib().rt_ads_tree_forms(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def light_forms(self):
        """(lights whose any-hit passes walk lists of their own, (light, triangle) pairs of the scene, pairs left out of
        those lists because the triangle can never occlude that light with back-face culling on); zeros = not in use."""
        a, b, c = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _capi.check(_capi.lib().rt_ads_light_forms(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def hit_lists(self):
        """((hit triangle, listed triangle) pairs of the per-hit-triangle lists of the small-tree form, pairs left out
        because the listed triangle has all of the hit triangle behind it); zeros = not in use."""
        a, b = C.c_uint64(), C.c_uint64()
        _capi.check(_capi.lib().rt_ads_hit_lists(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def tree_stats(self):
        st = _capi.TreeStats()
        _capi.check(_capi.lib().rt_ads_tree_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def collect_search_stats(self, on=True):
        """Also run the instrumented pass that counts nodes / leaves / candidates / ray-triangle tests in the
        reference's definitions (octree.cc:185-187,204; renderer.cc:137,305,311) for trace and render calls."""
        _capi.check(_capi.lib().rt_ads_collect_search_stats(self._h, int(bool(on))))

    def reset_query_stats(self):
        _capi.check(_capi.lib().rt_ads_reset_query_stats(self._h))

    def query_stats(self):
        st = _capi.QueryStats()
        _capi.check(_capi.lib().rt_ads_query_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}


def make_render_config(cam_pos, cam_up, cam_dir, fov, width, height, spp=1, max_depth=10, spt=1, render_type="distance",
                       shading="flat", cull=True, max_distance=2000.0, background=(0.01, 0.01, 0.01),
                       aspect_mode=_capi.RT_ASPECT_COMPAT, seed=42, band_rank=0, band_world=1):
    c = _capi.RenderConfig()
    c.struct_size = C.sizeof(_capi.RenderConfig)
    c.cam_pos[:] = cam_pos
    c.cam_up[:] = cam_up
    c.cam_dir[:] = cam_dir
    c.fov = fov
    c.width, c.height, c.samples_per_pixel = width, height, spp
    c.max_depth, c.samples_per_triangle = max_depth, spt
    if render_type not in _capi.RENDER_TYPES:
        raise RuntimeError("Invalid render type")  # renderer.cc:355
    if shading not in _capi.SHADING_TYPES:
        raise RuntimeError("Invalid shading type")  # renderer.cc:364
    c.render_type = _capi.RENDER_TYPES[render_type]
    c.shading_type = _capi.SHADING_TYPES[shading]
    c.cull_backfaces = int(cull)
    c.max_distance = max_distance
    c.background[:] = background
    c.aspect_mode, c.seed = aspect_mode, seed
    c.band_rank, c.band_world = band_rank, band_world
    return c


class HostTree:
    """The octree exactly as Build constructs it on the host, without touching a GPU (inspection / tests)."""

    NODE_DT = np.dtype([("lo", np.float32, (3, 2)), ("hi", np.float32, (3, 2)), ("nonempty", np.uint32), ("pad", np.uint32, 3),
                        ("info", np.uint32, (8, 2))])
    SLOT_DT = np.dtype([("a", np.float32, 3), ("e1", np.float32, 3), ("e2", np.float32, 3), ("tri", np.uint32),
                        ("pad", np.uint32, 2)])

    def __init__(self, triangles, max_triangles_per_bb=16, max_depth=10):
        pos = _f32(triangles).reshape(-1, 9)
        h = C.c_void_p()
        _capi.check(_capi.lib().rt_host_tree_build(_p(pos), pos.shape[0], max_triangles_per_bb, max_depth, C.byref(h)))
        self._h = h
        a, b, c, d = (C.c_uint64() for _ in range(4))
        ms = C.c_uint32()
        _capi.check(_capi.lib().rt_host_tree_sizes(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(ms)))
        self.n_nodes, self.n_pool, self.n_records, self.n_slots, self.max_stack = a.value, b.value, c.value, d.value, ms.value

    def export(self):
        n = self.n_nodes
        out = dict(is_leaf=np.zeros(n, np.uint8), depth=np.zeros(n, np.uint32), bmin=np.zeros((n, 3), np.float32),
                   bmax=np.zeros((n, 3), np.float32), child=np.zeros((n, 8), np.int32), tri_begin=np.zeros(n, np.uint64),
                   tri_count=np.zeros(n, np.uint32), pool=np.zeros(self.n_pool, np.uint32))
        _capi.check(_capi.lib().rt_host_tree_export(self._h, _p(out["is_leaf"]), _p(out["depth"]), _p(out["bmin"]), _p(out["bmax"]),
                                                    _p(out["child"]), _p(out["tri_begin"]), _p(out["tri_count"]), _p(out["pool"])))
        return out

    def export_flat(self):
        recs = np.zeros(self.n_records, self.NODE_DT)
        slots = np.zeros(self.n_slots, self.SLOT_DT)
        root = np.zeros(2, np.uint32)
        _capi.check(_capi.lib().rt_host_tree_export_flat(self._h, _p(recs), _p(slots), _p(root)))
        return recs, slots, root

    def export_clustered(self):
        """The clustered form: (node records, cluster triangle lists, root_info[1], number of subtree clusters), or
        None if the form does not apply to this tree."""
        r1, nc, nt = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _capi.check(_capi.lib().rt_host_tree_export_clustered(self._h, None, None, C.byref(nt), C.byref(r1), C.byref(nc)))
        if not nc.value:
            return None
        recs = np.zeros(self.n_records, self.NODE_DT)
        tris = np.zeros(nt.value, self.SLOT_DT)
        _capi.check(_capi.lib().rt_host_tree_export_clustered(self._h, _p(recs), _p(tris), None, None, None))
        return recs, tris, int(r1.value), int(nc.value)

    def stats(self):
        st = _capi.TreeStats()
        _capi.check(_capi.lib().rt_host_tree_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def __del__(self):
        try:
            if self._h:
                _capi.lib().rt_host_tree_destroy(self._h)
                self._h = None
        except Exception:
            pass
