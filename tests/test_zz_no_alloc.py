"""GPU test of the entry point that allocates nothing (rt_ads_search_noalloc): its output must equal rt_ads_search's
-- the same offsets and the same candidate lists, element for element -- on two calls in a row, the second with a
smaller input, and sizes beyond the fixed scratch must be refused with a clear error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Hip:
    """Device buffers for the test through the HIP runtime the library itself is linked to (the caller side may
    allocate; the entry point under test may not)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so.7")  # already loaded as a dependency of the library: the same runtime
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]

    def check(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def alloc(self, nbytes, fill=0xff):
        p = self.C.c_void_p()
        self.check(self.lib.hipMalloc(self.C.byref(p), max(nbytes, 16)))
        self.check(self.lib.hipMemset(p, fill, max(nbytes, 16)))
        return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        if arr.nbytes:
            self.check(self.lib.hipMemcpy(p, arr.ctypes.data_as(self.C.c_void_p), arr.nbytes, 1))
        return p

    def download(self, p, dtype, count):
        out = np.zeros(count, dtype)
        if count:
            self.check(self.lib.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, out.nbytes, 2))
        return out

    def free(self, *ptrs):
        for p in ptrs:
            self.check(self.lib.hipFree(p))


def _search_noalloc(g, rays, cap):
    hip = _Hip()
    n = rays.shape[0]
    d_rays = hip.upload(rays.astype(np.float32))
    d_off = hip.alloc((n + 1) * 8)       # every byte 0xff before the call
    d_out = hip.alloc(max(cap, 1) * 4)
    d_total = hip.alloc(8)
    hip.check(hip.lib.hipDeviceSynchronize())
    g.search_noalloc_device(d_rays, n, d_off, d_out, cap, d_total, stream=None)
    hip.check(hip.lib.hipDeviceSynchronize())
    off = hip.download(d_off, np.uint64, n + 1)
    out = hip.download(d_out, np.uint32, max(cap, 1))
    total = int(hip.download(d_total, np.uint64, 1)[0])
    hip.free(d_rays, d_off, d_out, d_total)
    return off, out, total


def test_search_noalloc_equals_search(scenes):
    import raytracer_amd as rt
    from raytracer_amd import _capi
    from test_gpu_parity import random_rays
    sc = scenes["sphere"]
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(sc.pos)
    o, d = random_rays(sc, 1200, 77)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    # two calls in a row on the same GPU (they share the scratch array), the second with a smaller input
    for n in (1200, 37, 1, 0):
        ref = g.search_batch(rays[:n])
        ref_off = np.concatenate([[0], np.cumsum([len(r) for r in ref])]).astype(np.uint64)
        total_ref = int(ref_off[-1])
        off, out, total = _search_noalloc(g, rays[:n], total_ref + 5)
        assert total == total_ref
        assert np.array_equal(off, ref_off)
        if total_ref:
            assert np.array_equal(out[:total_ref], np.concatenate(ref))
        assert (out[total_ref:] == 0xffffffff).all()  # nothing written past the lists
    assert total_ref == 0
    # an output buffer that is too small: the total is still reported, entries beyond cap are not written
    ref = g.search_batch(rays[:50])
    flat = np.concatenate(ref)
    off, out, total = _search_noalloc(g, rays[:50], len(flat) // 2)
    assert total == len(flat) and np.array_equal(out[: len(flat) // 2], flat[: len(flat) // 2])
    # limits: more rays than the maximum, and more queue than the scratch holds (1,584 nodes x 8 bytes per ray)
    with pytest.raises(RuntimeError, match="at most 4096 rays"):
        g.search_noalloc_device(1, _capi.RT_SEARCH_NOALLOC_MAX_RAYS + 1, 1, 1, 1, 1)
    too_many = _capi.RT_SEARCH_NOALLOC_SCRATCH_BYTES // (1584 * 8) + 1
    assert too_many <= _capi.RT_SEARCH_NOALLOC_MAX_RAYS
    with pytest.raises(RuntimeError, match="fixed scratch"):
        g.search_noalloc_device(1, too_many, 1, 1, 1, 1)
    # another tree on the same GPU afterwards (a root-leaf tree: every triangle is a candidate of every ray)
    g2 = rt.GpuOctree({"max_triangles_per_BB": 5000, "max_depth": 10})
    g2.Build(sc.pos)
    off, out, total = _search_noalloc(g2, rays[:3], 3 * sc.n_tri)
    assert total == 3 * sc.n_tri and np.array_equal(out[:total], np.tile(np.arange(sc.n_tri, dtype=np.uint32), 3))
