"""GPU parity tests: every query entry point of the C ABI against the oracle on the same inputs.

Bar: hit distances, triangle indices, occlusion flags and candidate lists bit-exact; images of the
RNG-free render types bit-exact; Phong / Blinn-Phong images within 1/255 per channel (device powf is
not libm's); ray counts equal to the oracle's Search-call counts."""
import numpy as np
import pytest

from conftest import CAMERAS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


def build(rt, sc, mb=16, md=10):
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
    g.Build(sc.pos)
    g.set_shading(sc.norm if sc.has_normals else None, sc.matid, sc.materials)
    return g


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def primary_rays(orc, scene, w, h):
    cfg = orc.make_cfg(fov=0.6, width=w, height=h, **CAMERAS[scene])
    d = orc.generate_rays(cfg).reshape(-1, 3)
    o = np.tile(np.asarray(CAMERAS[scene]["cam_pos"], np.float32), (d.shape[0], 1))
    return o, d


def random_rays(sc, n, seed):
    rs = np.random.RandomState(seed)
    lo, hi = sc.pos.reshape(-1, 3).min(0), sc.pos.reshape(-1, 3).max(0)
    o = (lo + (hi - lo) * rs.rand(n, 3) * 1.4 - 0.2 * (hi - lo)).astype(np.float32)
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # axis-parallel directions and exact zeros exercise the inf / NaN paths of the slab test
    d[::7, 0] = 0.0
    d[::11, 1] = 0.0
    d[::13, 2] = -0.0
    d[5::97] = [0.0, 0.0, 1.0]
    o[3::50] = sc.pos.reshape(-1, 3)[rs.randint(0, sc.pos.size // 3, len(o[3::50]))]  # origins on vertices
    return o, d.astype(np.float32)


@pytest.mark.parametrize("scene,mb,md", [("original", 16, 10), ("sphere", 16, 10), ("sphere", 4, 3), ("original", 1000, 10),
                                         ("original", 14, 3), ("original", 8, 2)])
@pytest.mark.parametrize("cull", [True, False])
def test_trace_closest(rt, orc, scenes, scene, mb, md, cull):
    sc = scenes[scene]
    g = build(rt, sc, mb, md)
    ot = orc.Octree(sc, mb, md)
    o1, d1 = primary_rays(orc, scene, 96, 64)
    o2, d2 = random_rays(sc, 20000, 7)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    t_ref, tri_ref = ot.trace_closest(o, d, cull)
    t, tri = g.trace_closest(o, d, cull)
    assert np.array_equal(tri, tri_ref)
    assert np.array_equal(bits(t), bits(t_ref))
    assert (tri >= 0).sum() > 1000


@pytest.mark.parametrize("scene", ["original", "sphere"])
def test_trace_any(rt, orc, scenes, scene):
    sc = scenes[scene]
    g = build(rt, sc)
    ot = orc.Octree(sc, 16, 10)
    o, d = primary_rays(orc, scene, 128, 96)
    t, tri = ot.trace_closest(o, d, True)
    hit = tri >= 0
    P = (o + d * t[:, None])[hit].astype(np.float32)
    rs = np.random.RandomState(3)
    lt = sc.pos[sc.light_tri[rs.randint(0, sc.n_light, P.shape[0])]]
    u, v = rs.rand(P.shape[0], 1).astype(np.float32), rs.rand(P.shape[0], 1).astype(np.float32)
    L = (lt[:, 0] + (lt[:, 1] - lt[:, 0]) * u * 0.5 + (lt[:, 2] - lt[:, 0]) * v * 0.5).astype(np.float32)
    for cull in (True, False):
        ref = ot.trace_any(P, L, cull)
        got = g.trace_any(P, L, cull)
        assert np.array_equal(ref, got)
        assert 0 < ref.sum() < ref.size


def test_division_fast_paths_are_exact():
    """rt_math.h replaces IEEE divisions by the division's own refinement chain where the operands are in range.
    tools/rcp_check (built by `make rcp_check`) compares, on the GPU, the reciprocal with 1.0f / d for every float of
    the range and the quotient with n / d for 2^35 operand pairs; any difference makes it exit non-zero."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rcp_check")
    assert os.path.exists(exe), "build it with `make -C raytracer_amd/csrc rcp_check` (__graft_entry__.build() does)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ from 1.0f / d" in r.stdout and "0 differ from n / d" in r.stdout, r.stdout


def test_task_pool_under_capacity_pressure(rt, scenes, monkeypatch, tmp_path):
    """The task-pool traversal bounds its LDS pool: a step takes fewer tasks when the pool is nearly full and a single
    one beyond the soft limit.  The product's capacities rarely get there, so a variant of the library built with tiny
    capacities (72 tasks soft limit, 128 leaf items; `make stress`) walks a deep tree (2 triangles per leaf, depth 12)
    with bundles of 64 rays and must give the same hits and occlusion flags, bit for bit, as the per-lane form of the
    product library."""
    import os
    import subprocess
    import sys
    from raytracer_amd import _capi
    from raytracer_amd.synthetic import cornell_with_icosphere
    stress = os.path.join(os.path.dirname(_capi.LIB_PATH), "libraytracer_amd_poolstress.so")
    assert os.path.exists(stress), "build it with `make -C raytracer_amd/csrc stress` (__graft_entry__.build() does)"
    sc = cornell_with_icosphere(scenes["original"], 5)
    o, d = random_rays(sc, 60000, 23)
    # coherent bundles: 64 neighbouring rays from one point towards the sphere
    rs = np.random.RandomState(5)
    centre = np.array([278.0, 200.0, 280.0], np.float32)
    ob = np.repeat((centre + rs.randn(400, 3) * 300).astype(np.float32), 64, axis=0)
    db = centre - ob + rs.randn(len(ob), 3).astype(np.float32) * 40
    db = (db / np.linalg.norm(db, axis=1, keepdims=True)).astype(np.float32)
    o, d = np.concatenate([o, ob]), np.concatenate([d, db])
    P = o[:40000].copy()
    L = (P + d[:40000] * 500.0).astype(np.float32)
    monkeypatch.setenv("RT_TRAVERSAL", "lane")
    g = rt.GpuOctree({"max_triangles_per_BB": 2, "max_depth": 12})
    g.Build(sc.pos)
    t_ref, tri_ref = g.trace_closest(o, d, True)
    occ_ref = g.trace_any(P, L, True)
    assert (tri_ref >= 0).sum() > 10000 and 0 < occ_ref.sum() < occ_ref.size
    fin, fout = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(fin, pos=sc.pos, o=o, d=d, P=P, L=L, mb=2, md=12)
    env = dict(os.environ, RT_LIB_PATH=stress, RT_TRAVERSAL="pool")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "pool_stress_child.py"), fin, fout], env=env, check=True,
                   timeout=200)
    got = np.load(fout)
    assert np.array_equal(got["tri"], tri_ref) and np.array_equal(bits(got["t"]), bits(t_ref))
    assert np.array_equal(got["occ"], occ_ref)


def test_small_tree_form_equals_general_form(rt, orc, scenes, monkeypatch):
    """Trees with at most 32 leaves are traversed in a special form (one pass over the triangle array with
    per-lane leaf-reach masks).  It must give the same hits as the general leaf-by-leaf traversal, ties included:
    rays aimed exactly at shared vertices and edges produce equal-t candidates."""
    sc = scenes["original"]
    o1, d1 = random_rays(sc, 20000, 21)
    verts = sc.pos.reshape(-1, 3)
    rs = np.random.RandomState(4)
    o2 = np.tile(np.asarray(CAMERAS["original"]["cam_pos"], np.float32), (4000, 1))
    tri = sc.pos[rs.randint(0, sc.n_tri, 4000)]
    w = rs.rand(4000, 1).astype(np.float32)
    target = np.where(rs.rand(4000, 1) < 0.5, tri[:, 0], tri[:, 0] * w + tri[:, 1] * (1 - w))  # a vertex, or a point on an edge
    d2 = (target - o2).astype(np.float32)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    for mb, md in ((16, 10), (14, 3)):
        ot = orc.Octree(sc, mb, md)
        for cull in (True, False):
            t_ref, tri_ref = ot.trace_closest(o, d, cull)
            monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "0")
            g_small = build(rt, sc, mb, md)
            monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
            g_general = build(rt, sc, mb, md)
            for g in (g_small, g_general):
                t, tri_ = g.trace_closest(o, d, cull)
                assert np.array_equal(tri_, tri_ref) and np.array_equal(bits(t), bits(t_ref))
    assert len(verts) > 0


@pytest.mark.parametrize("scene,mb,md", [("sphere", 16, 10), ("sphere", 4, 3), ("sphere", 2, 12), ("original", 16, 10), ("original", 1, 6),
                                         ("synthetic5", 16, 10), ("soup", 8, 6)])
def test_clustered_any_hit_form(rt, orc, scenes, monkeypatch, scene, mb, md):
    """Any-hit rays that the wave walks together use the clustered form of the tree (a triangle listed by several
    reached leaves of a cluster is tested once).  Its flags must equal the plain walk's and the oracle's: coherent and
    random segments, both culling modes, NaN / inf / zero components, segments ending on the first surface, and a
    rendered frame with area lights."""
    if scene == "synthetic5":
        sc = synthetic_scene(5)  # 20,514 triangles; RT_TRAVERSAL=wave keeps its any-hit rays in the wave-together form
        sc = orc.Scene.from_arrays(sc.pos, None, sc.matid, sc.materials)
    elif scene == "soup":
        rs = np.random.RandomState(21)
        c = rs.rand(3000, 1, 3).astype(np.float32) * 100
        pos = (c + rs.randn(3000, 3, 3).astype(np.float32) * rs.choice([0.5, 4.0, 30.0], (3000, 1, 1)).astype(np.float32)).astype(np.float32)
        pos[100:110] = pos[90:100]  # exact duplicates
        base = scenes["original"]
        sc = orc.Scene.from_arrays(pos, None, np.zeros(3000, np.int32), base.materials)
    else:
        sc = scenes[scene]
    monkeypatch.setenv("RT_TRAVERSAL", "wave")
    monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    monkeypatch.setenv("RT_DISABLE_CLUSTERS", "1")
    monkeypatch.setenv("RT_DISABLE_FAST_TRI", "1")
    g_plain = build(rt, sc, mb, md)
    assert g_plain.tree_forms() == (0, 0, 0)
    monkeypatch.delenv("RT_DISABLE_CLUSTERS")
    g_compiled = build(rt, sc, mb, md)  # clustered walk, compiled triangle test
    assert g_compiled.tree_forms()[1:] == (g_compiled.tree_forms()[1], 0) and g_compiled.tree_forms()[1] > 0
    monkeypatch.delenv("RT_DISABLE_FAST_TRI")
    g = build(rt, sc, mb, md)  # clustered walk, hand-scheduled triangle test
    assert g.tree_forms()[0] == 0 and g.tree_forms()[1] > 0 and g.tree_forms()[2] == 1
    monkeypatch.setenv("RT_DISABLE_CLUSTERS", "1")
    g_plain_fast = build(rt, sc, mb, md)  # plain walk, hand-scheduled triangle test
    assert g_plain_fast.tree_forms() == (0, 0, 1)
    monkeypatch.delenv("RT_DISABLE_CLUSTERS")
    ot = orc.Octree(sc, mb, md)
    o, d = random_rays(sc, 40000, 23)
    t_ref, tri_ref = ot.trace_closest(o, d, False)
    hit = tri_ref >= 0
    assert hit.sum() > 5000
    rs = np.random.RandomState(6)
    P = (o[hit] + d[hit] * t_ref[hit, None]).astype(np.float32)
    lo, hi = sc.pos.reshape(-1, 3).min(0), sc.pos.reshape(-1, 3).max(0)
    target = (lo + (hi - lo) * np.array([0.5, 0.95, 0.5], np.float32) + rs.randn(len(P), 3) * 0.02 * (hi - lo)).astype(np.float32)
    cases = [(P, target),                                     # from surfaces towards one region: coherent, like shadow rays
             (o[hit], P),                                     # segments that end on the first surface they meet
             (o, (o + d * rs.rand(len(o), 1) * np.linalg.norm(hi - lo)).astype(np.float32))]  # random segments
    with np.errstate(all="ignore"):
        special = np.array([0.0, -0.0, 1e-40, 1e30, -1e30, np.inf, -np.inf, np.nan], np.float32)
        Ps, Ls = P[:4096].copy(), target[:4096].copy()
        for i in range(0, len(Ps), 3):
            (Ps if i % 2 else Ls)[i, rs.randint(3)] = special[rs.randint(len(special))]
        cases.append((Ps, Ls))
        for Pc, Lc in cases:
            for n in (len(Pc), 65, 1):
                for cull in (True, False):
                    ref = ot.trace_any(Pc[:n], Lc[:n], cull)
                    for other in (g, g_plain, g_compiled, g_plain_fast):
                        assert np.array_equal(other.trace_any(Pc[:n], Lc[:n], cull), ref)
        occ = ot.trace_any(cases[0][0], cases[0][1], True)
        assert 0 < occ.sum() < occ.size
    if scene in ("sphere", "original"):
        kw = dict(fov=0.6, width=64, height=48, spt=5, render_type="blinn_phong", shading="smooth" if sc.has_normals else "flat",
                  aspect_mode=1, **CAMERAS[scene])
        cfg = rt.make_render_config(**kw)
        img, st = g.render(cfg)
        img_p, st_p = g_plain.render(cfg)
        assert np.array_equal(bits(img), bits(img_p)) and st.rays == st_p.rays and st.shadow_rays > 0
        ref, _, _, ost = orc.render(sc, ot, orc.make_cfg(rng_mode=orc.RNG_COUNTER, **{**kw, "aspect_mode": orc.ASPECT_TRUE}), want_hits=False)
        assert np.abs(img - ref).max() <= 1.0 / 255.0 and st.rays == ost.search_calls


def test_distances_of_the_three_triangle_test_forms(rt, orc):
    """collision_ray_triangle (collision_detection.h:12-68) exists three times in the kernels: compiled (ray_triangle<>),
    hand-scheduled with the triangle in scalar registers (any_hit_exec) and hand-scheduled with one triangle per lane
    (any_hit_exec_pairs).  The two hand-scheduled forms restate the reciprocal's refinement chain in their own
    instructions and normally only hand back a flag; rt_debug_triangle_tests makes all three return the DISTANCE they
    computed for every (ray, triangle) pair: the same bits everywhere, both culling modes, and -- through a root-leaf
    tree, whose Search returns every triangle -- the closest of them equal to the oracle's closest hit."""
    import ctypes as C
    from raytracer_amd import _capi
    rng = np.random.default_rng(77)
    n_tris = 12
    tris = (rng.uniform(-3, 3, (n_tris, 1, 3)) + rng.normal(0, 1.2, (n_tris, 3, 3))).astype(np.float32)
    tris[3] = tris[2][[0, 2, 1]]       # the same triangle with the opposite winding
    tris[5, 0] = tris[4, 0]            # a shared vertex ...
    tris[5, 1] = tris[4, 1]            # ... and a shared edge
    origins, targets = [], []
    for k in range(6000):
        t = tris[rng.integers(n_tris)]
        mode = k % 6
        if mode == 0:    # a vertex, exactly
            p = t[rng.integers(3)]
        elif mode == 1:  # a point on an edge
            a, b = t[rng.integers(3)], t[rng.integers(3)]
            w = np.float32(rng.uniform())
            p = a + (b - a) * w
        elif mode == 2:  # grazing: a point in the triangle's plane just outside
            u, v = rng.uniform(0.9, 1.1), rng.uniform(-0.1, 0.1)
            p = t[0] + (t[1] - t[0]) * np.float32(u) + (t[2] - t[0]) * np.float32(v)
        else:            # inside
            u, v = rng.uniform(), rng.uniform()
            if u + v > 1:
                u, v = 1 - u, 1 - v
            p = t[0] + (t[1] - t[0]) * np.float32(u) + (t[2] - t[0]) * np.float32(v)
        o = rng.uniform(-8, 8, 3).astype(np.float32)
        if mode == 5:    # an origin on the surface itself (secondary and shadow rays start there)
            o, p = p.astype(np.float32), rng.uniform(-8, 8, 3).astype(np.float32)
        origins.append(o)
        targets.append(np.asarray(p, np.float32))
    o = np.asarray(origins, np.float32)
    d = np.asarray(targets, np.float32) - o
    ln = np.sqrt((d * d).sum(1, dtype=np.float32), dtype=np.float32)
    d = (d / np.maximum(ln, np.float32(1e-30))[:, None]).astype(np.float32)
    d[::97] = 0.0  # zero-length directions (renderer.cc:187 skips such rays, the test must still agree)
    n = o.shape[0]
    sc = orc.Scene.from_arrays(tris, None, np.zeros(n_tris, np.int32), np.zeros((1, 18), np.float32))
    ot = orc.Octree(sc, 16, 10)  # 12 triangles <= 16: the root is a leaf, every triangle is a candidate of every ray
    for cull in (1, 0):
        out = np.zeros((3, n, n_tris), np.float32)
        _capi.check(_capi.lib().rt_debug_triangle_tests(0, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n,
                                                        np.ascontiguousarray(tris).ctypes.data_as(C.c_void_p), n_tris, cull,
                                                        out.ctypes.data_as(C.c_void_p)))
        u = out.view(np.uint32)
        assert np.array_equal(u[0], u[1]), cull  # the scalar-register form computes the compiled form's distances
        assert np.array_equal(u[0], u[2]), cull  # so does the per-lane form
        hits = np.isfinite(out[0])
        assert hits.sum() > 3000 and (~hits).sum() > 3000
        ref_t, ref_tri = ot.trace_closest(o, d, cull=bool(cull))
        best = out[0].min(axis=1)
        assert np.array_equal(best.view(np.uint32), np.asarray(ref_t, np.float32).view(np.uint32)), cull


def test_hand_scheduled_any_hit_test_and_its_range_gate(rt, orc, scenes, monkeypatch):
    """Shadow rays of scenes whose triangles are small enough use a triangle test whose reciprocal never takes the full
    division; it must give the oracle's flags in the small-tree pass too, and a scene beyond the bound (edges of 1e8
    units: |e1| * |e2| > 2^30) must fall back to the compiled test by itself."""
    sc = scenes["original"]
    ot = orc.Octree(sc, 16, 10)
    o, d = random_rays(sc, 30000, 31)
    t_ref, tri_ref = ot.trace_closest(o, d, False)
    hit = tri_ref >= 0
    P = (o[hit] + d[hit] * t_ref[hit, None]).astype(np.float32)
    rs = np.random.RandomState(9)
    lt = sc.pos[sc.light_tri[rs.randint(0, sc.n_light, P.shape[0])]]
    L = (lt[:, 0] * 0.5 + lt[:, 1] * 0.2 + lt[:, 2] * 0.3).astype(np.float32)
    L[::5] = P[::5]                       # zero-length segments
    L[1::7, 1] = np.inf                   # infinite and NaN targets
    L[2::11, 0] = np.nan
    for k in ("RT_DISABLE_SMALL_TREE", "RT_TRAVERSAL", "RT_DISABLE_CLUSTERS", "RT_DISABLE_FAST_TRI"):
        monkeypatch.delenv(k, raising=False)
    g_fast = build(rt, sc)
    assert g_fast.tree_forms()[0] > 0 and g_fast.tree_forms()[2] == 1
    monkeypatch.setenv("RT_DISABLE_FAST_TRI", "1")
    g_compiled = build(rt, sc)
    assert g_compiled.tree_forms()[2] == 0
    monkeypatch.delenv("RT_DISABLE_FAST_TRI")
    with np.errstate(all="ignore"):
        for cull in (True, False):
            ref = ot.trace_any(P, L, cull)
            assert np.array_equal(g_fast.trace_any(P, L, cull), ref) and np.array_equal(g_compiled.trace_any(P, L, cull), ref)
        assert 0 < ref.sum() < ref.size
    # the same scene scaled by 1e6: the bound fails, results still equal the oracle's
    big = orc.Scene.from_arrays((sc.pos * np.float32(1e6)).astype(np.float32), None, sc.matid, sc.materials)
    g_big = build(rt, big)
    assert g_big.tree_forms()[2] == 0
    ot_big = orc.Octree(big, 16, 10)
    Pb, Lb = (P * np.float32(1e6)).astype(np.float32), (L * np.float32(1e6)).astype(np.float32)
    with np.errstate(all="ignore"):
        for cull in (True, False):
            assert np.array_equal(g_big.trace_any(Pb, Lb, cull), ot_big.trace_any(Pb, Lb, cull))


@pytest.mark.parametrize("mode", ["wave", "lane", "pool"])
@pytest.mark.parametrize("scene,mb,md", [("sphere", 16, 10), ("sphere", 4, 3), ("original", 8, 2), ("original", 1000, 10)])
def test_both_traversals(rt, orc, scenes, monkeypatch, mode, scene, mb, md):
    """The traversal kernels (the wave walks together / every lane walks its own ray / the rays of a wave share a task pool) are forced in
    turn and must each reproduce the oracle: closest hits, occlusion flags and a rendered frame with shadow and secondary rays."""
    monkeypatch.setenv("RT_TRAVERSAL", mode)
    monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    monkeypatch.setenv("RT_PRUNE_MIN_SLOTS", "0")  # (wave form: the pruned closest-hit walk, which trees this small do not get by default)
    sc = scenes[scene]
    g = build(rt, sc, mb, md)
    ot = orc.Octree(sc, mb, md)
    o, d = random_rays(sc, 30000, 17)
    for cull in (True, False):
        t_ref, tri_ref = ot.trace_closest(o, d, cull)
        t, tri = g.trace_closest(o, d, cull)
        assert np.array_equal(tri, tri_ref) and np.array_equal(bits(t), bits(t_ref))
    hit = tri_ref >= 0
    P = (o[hit] + d[hit] * t_ref[hit, None]).astype(np.float32)
    rs = np.random.RandomState(5)
    lt = sc.pos[sc.light_tri[rs.randint(0, sc.n_light, P.shape[0])]]
    L = (lt[:, 0] * 0.4 + lt[:, 1] * 0.3 + lt[:, 2] * 0.3).astype(np.float32)
    for cull in (True, False):
        assert np.array_equal(ot.trace_any(P, L, cull), g.trace_any(P, L, cull))
    cam = CAMERAS[scene]
    shading = "smooth" if sc.has_normals else "flat"
    kw = dict(fov=0.6, width=48, height=40, spt=3, render_type="blinn_phong", shading=shading, **cam)
    ref, ref_tri, ref_t, st = orc.render(sc, ot, orc.make_cfg(rng_mode=orc.RNG_COUNTER, **kw))
    img, gst = g.render(rt.make_render_config(**kw))
    assert gst.rays == st.search_calls
    assert np.abs(img - ref).max() <= 1.0 / 255.0


def coplanar_layers_scene(orc, scenes):
    """A plane of 32 x 32 unit cells (2 triangles each) covered by the same plane again in triangles with legs 8, 16 and
    32 -- listed after the cells -- plus a little geometry above and below.  Every coordinate is a small
    integer and every leg a power of two, so a ray with dyadic components meets all layers at bit-equal distances: the
    winner is decided by the reference's list order alone, and the big triangles are listed by leaves far from the hit."""
    def cover(n, z=0.0):
        out = []
        for i in range(0, 32, n):
            for j in range(0, 32, n):
                out += [[[i, j, z], [i + n, j, z], [i, j + n, z]], [[i + n, j + n, z], [i, j + n, z], [i + n, j, z]]]
        return out
    tris = cover(1) + cover(8) + cover(32) + cover(16)
    tris += [[[3, 3, 6], [5, 3, 6], [3, 5, 6]], [[20, 20, -10], [28, 20, -10], [20, 28, -10]], [[0, 0, -10], [4, 0, -10], [0, 4, -10]]]
    pos = np.asarray(tris, np.float32)
    base = scenes["original"]
    return orc.Scene.from_arrays(pos, None, np.zeros(pos.shape[0], base.matid.dtype), base.materials)


def tie_rays(n, rs):
    """Dyadic origins over the square, dyadic directions from steep to 1 : 16 shallow, from both sides of the plane."""
    o = np.stack([rs.randint(0, 128, n) / 4.0, rs.randint(0, 128, n) / 4.0, rs.choice([4.0, 2.0, 1.0, -4.0], n)], axis=1).astype(np.float32)
    steps = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, -3.0, 4.0, 0.25], np.float32)
    d = np.stack([rs.choice(steps, n), rs.choice(steps, n), -np.sign(o[:, 2]) * rs.choice([1.0, 0.5, 0.25], n)], axis=1).astype(np.float32)
    return o, d


@pytest.mark.parametrize("mode", ["auto", "wave", "lane", "pool"])
@pytest.mark.parametrize("clip", ["1", "0"])
def test_pruned_walks_keep_the_tie_rule(rt, orc, scenes, monkeypatch, mode, clip):
    """Closest-hit walks that stop following a ray beyond its best hit, and any-hit walks that cover the segment only,
    must still name the triangle the reference's loop over the WHOLE candidate list names when several triangles are met
    at the same distance (renderer.cc:130: the first one in list order) -- including when the first listing leaf lies
    beyond the hit or behind the origin.  Every traversal form, clipped and unclipped, against the oracle; the scene makes
    nearly every hit such a tie."""
    monkeypatch.setenv("RT_CLIP", clip)
    monkeypatch.setenv("RT_PRUNE_MIN_SLOTS", "0")  # (the wave-together closest-hit walk prunes on trees of any size)
    if mode != "auto":
        monkeypatch.setenv("RT_TRAVERSAL", mode)
        monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    sc = coplanar_layers_scene(orc, scenes)
    for mb, md in ((8, 6), (16, 10)):
        g = build(rt, sc, mb, md)
        ot = orc.Octree(sc, mb, md)
        rs = np.random.RandomState(23)
        n = 6000
        o, d = tie_rays(n, rs)
        for cull in (False, True):
            t_ref, tri_ref = ot.trace_closest(o, d, cull)
            t, tri = g.trace_closest(o, d, cull)
            assert np.array_equal(tri, tri_ref) and np.array_equal(bits(t), bits(t_ref))
        # the ties are real and not trivially ordered: the plane is met inside the square by most rays, and the winner is
        # neither always the lowest nor always the highest triangle index of the layers
        t_ref, tri_ref = ot.trace_closest(o, d, False)
        on_plane = (tri_ref >= 0) & (tri_ref < len(sc.pos) - 3)
        assert on_plane.sum() > 2000
        layer = np.bincount(np.digitize(tri_ref[on_plane], [2048, 2048 + 32, 2048 + 34]), minlength=4)
        assert layer[0] > 300 and layer[1:].sum() > 300
        # segments that end exactly on the plane (not occluded by it: renderer.cc:302 wants t < distance) or just beyond it
        oo, P = o[on_plane], (o[on_plane] + d[on_plane] * t_ref[on_plane, None]).astype(np.float32)
        beyond = (oo + (P - oo) * np.float32(1.25)).astype(np.float32)
        short = (oo + (P - oo) * np.float32(0.75)).astype(np.float32)
        for cull in (False, True):
            for L in (P, beyond, short):
                assert np.array_equal(ot.trace_any(oo, L, cull), g.trace_any(oo, L, cull))
        assert ot.trace_any(oo, beyond, False).sum() > 1000 and ot.trace_any(oo, short, False).sum() < 100


@pytest.mark.parametrize("scene,mb,md", [("original", 16, 10), ("sphere", 16, 10), ("sphere", 4, 3)])
def test_search_candidate_lists(rt, orc, scenes, scene, mb, md):
    sc = scenes[scene]
    g = build(rt, sc, mb, md)
    ot = orc.Octree(sc, mb, md)
    o, d = random_rays(sc, 300, 11)
    got = g.search_batch(np.concatenate([o, d], axis=1))
    for i in range(o.shape[0]):
        assert np.array_equal(got[i], ot.search(o[i], d[i])), i
    one = g.Search(np.concatenate([o[0], d[0]]))
    assert np.array_equal(one, ot.search(o[0], d[0]))


def test_search_of_a_nan_ray_lists_every_leaf_reference(rt, orc, scenes):
    """A ray with NaN components passes every slab test (std::max / std::min keep their first argument,
    collision_detection.h:82-83), so Search returns the list of every leaf: millions of candidates for one ray on a soup
    of large triangles -- more than the oracle wrapper's default buffer and than one region of the device's batched
    search.  Element for element the oracle's list, alone and inside a batch."""
    rs = np.random.RandomState(77)
    n_tri = 9000
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.03), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    sc = orc.Scene.from_arrays(pos, None, np.zeros(n_tri, np.int32), scenes["sphere"].materials)
    g = build(rt, sc, 4, 8)
    ot = orc.Octree(sc, 4, 8)
    o, d = random_rays(sc, 9, 3)
    o[4, 1] = np.nan
    d[4, 2] = np.nan
    ref = ot.search(o[4], d[4])
    assert len(ref) == g.tree_stats()["leaf_refs"] > (1 << 20)
    got = g.search_batch(np.concatenate([o, d], axis=1))
    for i in range(o.shape[0]):
        assert np.array_equal(got[i], ref if i == 4 else ot.search(o[i], d[i])), i
    assert np.array_equal(g.Search(np.concatenate([o[4], d[4]])), ref)


def test_empty_and_tiny_scenes(rt, orc, scenes):
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(np.zeros((0, 3, 3), np.float32))
    t, tri = g.trace_closest([[0, 0, 0]], [[0, 0, 1]])
    assert tri[0] == -1 and np.isinf(t[0])
    assert len(g.Search([0, 0, 0, 0, 0, 1])) == 0
    sc = orc.Scene.from_arrays(scenes["original"].pos[:5], None, scenes["original"].matid[:5], scenes["original"].materials)
    g = build(rt, sc)
    assert g.tree_stats()["nodes"] == 1
    ot = orc.Octree(sc, 16, 10)
    o, d = random_rays(sc, 2000, 5)
    t_ref, tri_ref = ot.trace_closest(o, d, False)
    t, tri = g.trace_closest(o, d, False)
    assert np.array_equal(tri, tri_ref) and np.array_equal(bits(t), bits(t_ref))
    assert np.array_equal(g.Search(np.concatenate([o[0], d[0]])), np.arange(5))


@pytest.mark.parametrize("mode", ["auto", "wave", "lane", "pool"])
@pytest.mark.parametrize("scene", ["original", "sphere"])
def test_degenerate_rays_and_ragged_batches(rt, orc, scenes, monkeypatch, scene, mode):
    """Zero, denormal, huge, infinite and NaN ray components take the inf / NaN branches of the slab and triangle tests;
    batch sizes 0, 1, 63, 65 and 257 leave waves and blocks partly empty.  Results must still equal the oracle's bits,
    in every traversal form."""
    if mode != "auto":
        monkeypatch.setenv("RT_TRAVERSAL", mode)
        monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
        monkeypatch.setenv("RT_PRUNE_MIN_SLOTS", "0")
    sc = scenes[scene]
    g = build(rt, sc)
    ot = orc.Octree(sc, 16, 10)
    o, d = random_rays(sc, 4096, 99)
    special = np.array([0.0, -0.0, 1e-40, -1e-40, 1e30, -1e30, np.inf, -np.inf, np.nan, 1.0], np.float32)
    rs = np.random.RandomState(8)
    for i in range(0, 2048, 2):
        d[i, rs.randint(3)] = special[rs.randint(len(special))]
    for i in range(1, 1024, 4):
        o[i, rs.randint(3)] = special[rs.randint(len(special))]
    d[7] = [0.0, 0.0, 0.0]
    d[9] = [np.nan, np.nan, np.nan]
    o[11] = sc.pos[0, 0]           # origin on a vertex ...
    d[11] = sc.pos[0, 1] - sc.pos[0, 0]  # ... direction along an edge
    with np.errstate(all="ignore"):
        for cull in (True, False):
            t_ref, tri_ref = ot.trace_closest(o, d, cull)
            t, tri = g.trace_closest(o, d, cull)
            assert np.array_equal(tri, tri_ref) and np.array_equal(bits(t), bits(t_ref))
            L = (o + d * 0.75).astype(np.float32)
            assert np.array_equal(ot.trace_any(o, L, cull), g.trace_any(o, L, cull))
        for n in (0, 1, 63, 65, 257):
            t_ref, tri_ref = ot.trace_closest(o[:n], d[:n], True)
            t, tri = g.trace_closest(o[:n], d[:n], True)
            assert t.shape == (n,) and np.array_equal(tri, tri_ref) and np.array_equal(bits(t), bits(t_ref))
            assert np.array_equal(ot.trace_any(o[:n], d[:n], True), g.trace_any(o[:n], d[:n], True))
    assert len(g.search_batch(np.zeros((0, 6), np.float32))) == 0


def test_tree_stats_match_reference(rt, scenes):
    g = build(rt, scenes["sphere"])
    st = g.tree_stats()
    assert (st["nodes"], st["leaves"], st["max_triangles_in_leaf"], st["max_depth"]) == (1584, 1300, 34, 10)


RENDER_CASES = [
    # name, scene, render_type, shading, w, h, spt, spp, cull, exact
    ("orig_distance", "original", "distance", "flat", 128, 128, 1, 1, True, True),
    ("orig_distance_wide", "original", "distance", "flat", 200, 120, 1, 1, True, True),
    ("sphere_diffusion", "sphere", "diffusion", "smooth", 128, 128, 1, 1, True, True),
    ("sphere_distance_nocull", "sphere", "distance", "flat", 128, 128, 1, 1, False, True),
    ("orig_blinn", "original", "blinn_phong", "flat", 96, 96, 4, 1, True, False),
    ("orig_phong_tall", "original", "phong", "flat", 64, 96, 2, 1, True, False),
    ("sphere_phong", "sphere", "phong", "smooth", 64, 64, 2, 1, True, False),
    ("sphere_blinn_spp", "sphere", "blinn_phong", "smooth", 64, 64, 2, 3, True, False),
    ("sphere_diffusion_spp", "sphere", "diffusion", "smooth", 61, 45, 1, 2, True, True),
]


@pytest.mark.parametrize("case", RENDER_CASES, ids=[c[0] for c in RENDER_CASES])
def test_render_matches_oracle(rt, orc, scenes, case):
    name, scene, rtype, shading, w, h, spt, spp, cull, exact = case
    sc = scenes[scene]
    g = build(rt, sc)
    ot = orc.Octree(sc, 16, 10)
    cam = CAMERAS[scene]
    ocfg = orc.make_cfg(fov=0.6, width=w, height=h, spp=spp, spt=spt, render_type=rtype, shading=shading, cull=cull,
                        rng_mode=orc.RNG_COUNTER, **cam)
    ref, ref_tri, ref_t, st = orc.render(sc, ot, ocfg)
    gcfg = rt.make_render_config(fov=0.6, width=w, height=h, spp=spp, spt=spt, render_type=rtype, shading=shading, cull=cull, **cam)
    img, gst = g.render(gcfg)
    t, tri = g.primary_hits(gcfg)
    assert np.array_equal(tri, ref_tri)
    assert np.array_equal(bits(t), bits(ref_t))
    # one ray = one Search call of the reference
    assert gst.rays == st.search_calls
    assert gst.primary_rays == st.primary_rays and gst.secondary_rays == st.secondary_rays and gst.shadow_rays == st.shadow_rays
    if exact:
        assert np.array_equal(bits(img), bits(ref))
        assert orc.ppm_md5(img) == orc.ppm_md5(ref)
    else:
        assert np.abs(img - ref).max() <= 1.0 / 255.0
        q = lambda a: (255.999 * a.astype(np.float64)).astype(np.int64)
        assert np.abs(q(img) - q(ref)).max() <= 1


@pytest.mark.parametrize("levels,w,h,spt", [(5, 96, 96, 2), (8, 64, 64, 1)])
def test_synthetic_icosphere_scene(rt, orc, levels, w, h, spt):
    """The deep-tree workload (levels = 8: 1,310,754 triangles, 533,107 nodes, depth 10): primary hits, shadow rays
    and the Phong image against the oracle at a size the oracle finishes in seconds, in true-aspect mode."""
    import os
    from conftest import GOLDEN
    from raytracer_amd.scene import Scene
    from raytracer_amd.synthetic import cornell_with_icosphere
    s = cornell_with_icosphere(Scene.load_npz(os.path.join(GOLDEN, "scenes", "cornell_original.npz")), levels)
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(s.pos)
    g.set_shading(None, s.matid, s.materials)
    osc = orc.Scene.from_arrays(s.pos, None, s.matid, s.materials)
    ot = orc.Octree(osc, 16, 10)
    cam = CAMERAS["original"]
    kw = dict(fov=0.6, width=w, height=h, spt=spt, render_type="phong", shading="flat", **cam)
    ref, ref_tri, ref_t, st = orc.render(osc, ot, orc.make_cfg(rng_mode=orc.RNG_COUNTER, aspect_mode=orc.ASPECT_TRUE, **kw))
    gcfg = rt.make_render_config(aspect_mode=1, **kw)
    img, gst = g.render(gcfg)
    t, tri = g.primary_hits(gcfg)
    assert np.array_equal(tri, ref_tri) and np.array_equal(bits(t), bits(ref_t))
    assert (tri >= 34).sum() > 100  # the sphere is in view
    assert gst.rays == st.search_calls and gst.shadow_rays == st.shadow_rays
    assert np.abs(img - ref).max() <= 1.0 / 255.0
    # random rays through the deep part of the tree, both culling modes
    o, d = random_rays(osc, 20000, 3)
    for cull in (True, False):
        t_ref, tri_ref = ot.trace_closest(o, d, cull)
        t2, tri2 = g.trace_closest(o, d, cull)
        assert np.array_equal(tri2, tri_ref) and np.array_equal(bits(t2), bits(t_ref))


def synthetic_scene(levels):
    import os
    from conftest import GOLDEN
    from raytracer_amd.scene import Scene
    from raytracer_amd.synthetic import cornell_with_icosphere
    return cornell_with_icosphere(Scene.load_npz(os.path.join(GOLDEN, "scenes", "cornell_original.npz")), levels)


@pytest.mark.parametrize("scene,mb,md", [("original", 16, 10), ("original", 8, 2), ("original", 1000, 10), ("original", 4, 0),
                                         ("sphere", 16, 10), ("sphere", 4, 3), ("sphere", 2, 12), ("ico5", 16, 10),
                                         ("ico6", 8, 7), ("ico8", 16, 10)])
def test_device_builder_equals_host_builder(rt, scenes, scene, mb, md):
    """Octree::Build run on the device gives the resident tree of the host builder bit for bit: internal-node
    records (octant planes, links, counts), leaf slots in the reference's candidate order, root link, stack bound and
    the PrintStats figures."""
    from raytracer_amd.ads import HostTree
    pos = synthetic_scene(int(scene[3:])).pos if scene.startswith("ico") else scenes[scene].pos
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md, "builder": "device"})
    g.Build(pos)
    nodes, slots, root, max_stack, on_device = g.export_tree()
    assert on_device
    h = HostTree(pos, mb, md)
    h_nodes, h_slots, h_root = h.export_flat()
    assert root == tuple(int(x) for x in h_root) and max_stack == h.max_stack
    assert nodes.shape == h_nodes.shape and slots.shape == h_slots.shape
    assert nodes.tobytes() == h_nodes.tobytes()
    assert slots.tobytes() == h_slots.tobytes()
    st, hst = g.tree_stats(), h.stats()
    for key in ("max_depth", "nodes", "leaves", "max_triangles_in_leaf", "leaf_refs", "internal_nodes"):
        assert st[key] == hst[key], key
    for key in ("avg_leaf_depth", "avg_triangles_in_leaf"):
        assert st[key] == hst[key] or (np.isnan(st[key]) and np.isnan(hst[key])), key
    # and the host-built resident tree is what rt_host_tree_export_flat shows
    g2 = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md, "builder": "host"})
    g2.Build(pos)
    n2, s2, r2, m2, d2 = g2.export_tree()
    assert not d2 and n2.tobytes() == h_nodes.tobytes() and s2.tobytes() == h_slots.tobytes() and r2 == root and m2 == max_stack


def _rng_free_kats():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "ref_kats.json")) as f:
        kats = json.load(f)
    return {k: v for k, v in kats.items() if v["spec"]["render_type"] in ("distance", "diffusion") and v["spec"]["ads"] == "octree"}


@pytest.mark.parametrize("name", sorted(_rng_free_kats()))
def test_reference_binary_known_answers(rt, scenes, name):
    """The render types that draw no random numbers, straight against what the unmodified reference binary produced
    (tests/golden/ref_kats.json): the md5 of the PPM file it wrote and its "Search method call count" -- camera rays plus
    every reflection / refraction ray of the recursion -- and the tree shape it printed."""
    import hashlib
    k = _rng_free_kats()[name]
    sp = k["spec"]
    if sp["scene"] == "synthetic":
        s = synthetic_scene(sp["levels"])
        pos, nrm, matid, mats, cam = s.pos, None, s.matid, s.materials, CAMERAS["original"]
    else:
        sc = scenes[sp["scene"]]
        pos, nrm, matid, mats, cam = sc.pos, (sc.norm if sc.has_normals else None), sc.matid, sc.materials, CAMERAS[sp["scene"]]
    g = rt.GpuOctree({"max_triangles_per_BB": sp.get("tri_per_bb", 16), "max_depth": sp.get("ads_depth", 10)})
    g.Build(pos)
    g.set_shading(nrm, matid, mats)
    img, st = g.render(rt.make_render_config(fov=0.6, width=sp["w"], height=sp["h"], spp=sp.get("spp", 1), spt=sp.get("spt", 1),
                                             render_type=sp["render_type"], shading=sp["shading"], cull=sp.get("cull", True), **cam))
    # the P3 text of Renderer::save_image (renderer.cc:381-390, color.h:12-15)
    q = (255.999 * img.astype(np.float64)).astype(np.int64).reshape(-1, 3)
    text = f"P3\n{sp['w']} {sp['h']}\n255\n" + "".join(f"{r} {gg} {b}\n" for r, gg, b in q.tolist())
    assert hashlib.md5(text.encode()).hexdigest() == k["ppm_md5"]
    assert st.rays == k["search_calls"]
    ts = g.tree_stats()
    assert (ts["nodes"], ts["leaves"], ts["max_depth"], ts["max_triangles_in_leaf"]) == (k["nodes"], k["leaves"], k["max_depth"], k["max_tris_in_leaf"])


LIVE_CASES = [
    # scene, render_type, shading, cull, w, h, camera position, direction, fov, octree (per box, depth)
    ("original", "distance", "flat", True, 90, 120, [100.0, 400.0, -700.0], [0.2, -0.15, 1.0], 0.8, (16, 10)),
    ("original", "diffusion", "flat", False, 75, 75, [278.0, 273.0, -400.0], [0.0, 0.0, 1.0], 1.1, (6, 4)),
    ("sphere", "diffusion", "smooth", True, 64, 100, [0.3, 1.2, 3.9], [-0.05, -0.1, -1.0], 0.7, (16, 10)),
    ("sphere", "distance", "flat", False, 101, 67, [-0.4, 0.8, 4.2], [0.1, 0.05, -1.0], 0.5, (3, 12)),
]


@pytest.mark.parametrize("case", LIVE_CASES, ids=[f"{c[0]}_{c[1]}_{i}" for i, c in enumerate(LIVE_CASES)])
def test_against_the_reference_binary_run_live(rt, scenes, case):
    """Configurations that are in no fixture: the reference binary (oracle/_ref/raytracer_ref, travelling with the
    repository) renders them on this host, the device path renders them too, and the PPM files must be identical byte
    for byte, like the Search call counts (camera + reflection + refraction rays) and the tree shape."""
    import hashlib
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_runner
    # the binary is built where the reference checkout exists and travels with the working tree; a GPU run without it
    # would silently lose the checks against the reference itself
    assert ref_runner.available(), "oracle/_ref/raytracer_ref is missing: run `make -C oracle ref` where /root/reference exists"
    scene, rtype, shading, cull, w, h, cpos, cdir, fov, (mb, md) = case
    sc = scenes[scene]
    nrm = sc.norm if sc.has_normals else None
    maxd = CAMERAS[scene]["max_distance"]
    ref = ref_runner.run(sc.pos, nrm, sc.matid, sc.materials, cpos, [0.0, 1.0, 0.0], cdir, fov, w, h, render_type=rtype,
                         shading=shading, cull=cull, max_distance=maxd, tri_per_bb=mb, ads_depth=md)
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
    g.Build(sc.pos)
    g.set_shading(nrm, sc.matid, sc.materials)
    img, st = g.render(rt.make_render_config(cam_pos=cpos, cam_up=[0.0, 1.0, 0.0], cam_dir=cdir, fov=fov, width=w, height=h,
                                             render_type=rtype, shading=shading, cull=cull, max_distance=maxd))
    q = (255.999 * img.astype(np.float64)).astype(np.int64).reshape(-1, 3)
    text = f"P3\n{w} {h}\n255\n" + "".join(f"{r} {gg} {b}\n" for r, gg, b in q.tolist())
    assert hashlib.md5(text.encode()).hexdigest() == ref["ppm_md5"]
    assert st.rays == ref["search_calls"]
    ts = g.tree_stats()
    assert (ts["nodes"], ts["leaves"]) == (ref["nodes"], ref["leaves"])


@pytest.mark.parametrize("seed,n_tri,mb,md", [(1, 300, 16, 10), (2, 2000, 8, 6), (3, 5000, 16, 10), (4, 64, 2, 8)])
@pytest.mark.parametrize("builder", ["host", "device"])
def test_random_triangle_soups_against_the_reference_binary(rt, scenes, seed, n_tri, mb, md, builder):
    """Geometry nobody tuned for: random triangles of very different sizes, slivers, exact duplicates and triangles
    that share vertices, with random materials (so reflection and refraction rays occur).  The reference binary builds
    its octree and renders the diffusion image on this host; both builders of the device path must give the same tree
    shape, the same Search call count and the same PPM bytes."""
    import hashlib
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_runner
    # the binary is built where the reference checkout exists and travels with the working tree; a GPU run without it
    # would silently lose the checks against the reference itself
    assert ref_runner.available(), "oracle/_ref/raytracer_ref is missing: run `make -C oracle ref` where /root/reference exists"
    rs = np.random.RandomState(seed)
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.02), np.log(4.0), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    pos[::17, 2] = pos[::17, 1] + (pos[::17, 1] - pos[::17, 0]) * 1e-3 + 1e-4   # slivers
    pos[5::40] = pos[4::40][: len(pos[5::40])]                                   # exact duplicates
    pos[7::23, 0] = pos[6::23, 0][: len(pos[7::23])]                             # shared vertices
    mats = scenes["sphere"].materials
    matid = rs.randint(0, mats.shape[0], n_tri).astype(np.int32)
    cpos, cdir, fov, w, h = [0.5, 0.3, -14.0], [0.0, 0.0, 1.0], 0.9, 96, 128
    ref = ref_runner.run(pos, None, matid, mats, cpos, [0.0, 1.0, 0.0], cdir, fov, w, h, render_type="diffusion", shading="flat",
                         cull=False, max_distance=40.0, tri_per_bb=mb, ads_depth=md, max_depth=4)
    g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md, "builder": builder})
    g.Build(pos)
    g.set_shading(None, matid, mats)
    img, st = g.render(rt.make_render_config(cam_pos=cpos, cam_up=[0.0, 1.0, 0.0], cam_dir=cdir, fov=fov, width=w, height=h,
                                             render_type="diffusion", shading="flat", cull=False, max_distance=40.0, max_depth=4))
    ts = g.tree_stats()
    assert (ts["nodes"], ts["leaves"]) == (ref["nodes"], ref["leaves"])
    assert st.rays == ref["search_calls"]
    q = (255.999 * img.astype(np.float64)).astype(np.int64).reshape(-1, 3)
    text = f"P3\n{w} {h}\n255\n" + "".join(f"{r} {gg} {b}\n" for r, gg, b in q.tolist())
    assert hashlib.md5(text.encode()).hexdigest() == ref["ppm_md5"]


@pytest.mark.parametrize("name", ["K5", "K9"])
def test_area_light_images_match_the_reference_statistically(rt, orc, scenes, name):
    """With area lights the reference consumes one global rand() stream in pixel order, which no parallel renderer can
    replay; the counter-based generator must give the same picture up to sampling noise.  The reference's own image is
    reproduced by the oracle in its sequential rand() mode (pinned by md5 in test_oracle_pin.py)."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "ref_kats.json")) as f:
        k = json.load(f)[name]
    sp = k["spec"]
    sc = scenes[sp["scene"]]
    kw = dict(fov=0.6, width=sp["w"], height=sp["h"], spp=sp.get("spp", 1), spt=sp.get("spt", 1), render_type=sp["render_type"],
              shading=sp["shading"], cull=sp.get("cull", True), **CAMERAS[sp["scene"]])
    ref, _, _, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_LEGACY, **kw), want_hits=False)
    assert orc.ppm_md5(ref) == k["ppm_md5"]
    img, st = build(rt, sc).render(rt.make_render_config(**kw))
    diff = img.astype(np.float64) - ref.astype(np.float64)
    assert abs(diff.mean()) < 2e-3  # no bias
    # no more than sampling noise: the CPU restatement with the counter generator and another seed differs as much
    other, _, _, _ = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, seed=1234, **kw), want_hits=False)
    noise = np.sqrt(((other.astype(np.float64) - ref.astype(np.float64)) ** 2).mean())
    assert np.sqrt((diff ** 2).mean()) < 1.25 * noise + 1e-3
    assert abs(st.rays - ost.search_calls) <= 0.01 * st.rays  # ray counts differ only through noise-dependent early exits


def test_render_is_partition_independent(rt, scenes):
    """Rendering the image in interleaved bands (as 2 or 3 ranks would) gives the same pixels as one call."""
    sc = scenes["original"]
    g = build(rt, sc)
    cam = CAMERAS["original"]
    kw = dict(fov=0.6, width=72, height=100, spt=3, render_type="blinn_phong", shading="flat", **cam)
    full, _ = g.render(rt.make_render_config(**kw))
    for world in (2, 3):
        out = np.zeros_like(full)
        rays = 0
        for r in range(world):
            _, st = g.render(rt.make_render_config(band_rank=r, band_world=world, **kw), out=out)
            rays += st.rays
        assert np.array_equal(bits(out), bits(full))


def test_full_size_frame_properties(rt, orc, scenes, monkeypatch):
    """The headline configuration at its full size (Cornell-Original, Blinn-Phong, 16 light samples, 1920x1080):
    the reference's ray count, the oracle's pixels on a band of rows, and size-independent properties -- every
    traversal form renders the same bits, a 4-way band split reassembles to the same frame, two runs are identical."""
    sc = scenes["original"]
    cam = CAMERAS["original"]
    kw = dict(fov=0.6, width=1920, height=1080, spt=16, render_type="blinn_phong", shading="flat", **cam)
    g = build(rt, sc)
    cfg = rt.make_render_config(**kw)
    full, st = g.render(cfg)
    assert st.rays == 77841000  # "Search method call count" of the reference binary for this frame (golden K-table)
    again, st2 = g.render(cfg)
    assert np.array_equal(bits(again), bits(full)) and st2.rays == st.rays
    # rows 536..543 against the oracle (compat aspect: every row is the same scanline, the RNG differs per pixel)
    ref, _, _, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(536, 544), **kw), want_hits=False)
    assert np.abs(full[536:544] - ref[536:544]).max() <= 1.0 / 255.0
    # the other traversal forms
    for env in ({"RT_DISABLE_SMALL_TREE": "1"}, {"RT_DISABLE_SMALL_TREE": "1", "RT_TRAVERSAL": "lane"},
                {"RT_DISABLE_SMALL_TREE": "1", "RT_TRAVERSAL": "pool"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g2 = build(rt, sc)
        img2, s2 = g2.render(cfg)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(bits(img2), bits(full)) and s2.rays == st.rays, env
    # 4 ranks' bands
    out = np.zeros_like(full)
    rays = 0
    for r in range(4):
        _, sr = g.render(rt.make_render_config(band_rank=r, band_world=4, **kw), out=out)
        rays += sr.rays
    assert np.array_equal(bits(out), bits(full)) and rays == st.rays


@pytest.mark.parametrize("spt", [3, 5, 33, 70])
@pytest.mark.parametrize("mode", ["auto", "lane", "pool"])
def test_light_sample_layouts(rt, orc, scenes, monkeypatch, spt, mode):
    """Sample counts that are not powers of two, exceed one visibility word (32) or one wave (64), with three lights
    (one more triangle made emissive): the any-hit pass spreads (light, sample) pairs over lane groups differently
    in each case, the image and the ray counts must not notice."""
    if mode != "auto":
        monkeypatch.setenv("RT_TRAVERSAL", mode)
        monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    base = scenes["original"]
    matid = base.matid.copy()
    light_mat = int(matid[base.light_tri[0]])
    matid[5] = light_mat  # a wall triangle becomes a third light
    sc = orc.Scene.from_arrays(base.pos, None, matid, base.materials)
    assert sc.n_light == 3
    g = build(rt, sc)
    ot = orc.Octree(sc, 16, 10)
    kw = dict(fov=0.6, width=40, height=48, spt=spt, render_type="phong", shading="flat", **CAMERAS["original"])
    ref, _, _, ost = orc.render(sc, ot, orc.make_cfg(rng_mode=orc.RNG_COUNTER, **kw), want_hits=False)
    img, st = g.render(rt.make_render_config(**kw))
    assert st.shadow_rays == ost.shadow_rays and st.rays == ost.search_calls
    assert np.abs(img - ref).max() <= 1.0 / 255.0


def test_large_tree_frame_properties(rt, monkeypatch):
    """The 1,310,754-triangle scene at a size no CPU check finishes quickly (1920x1080, true aspect): the frame must not
    depend on where the tree was built or on which traversal form walks it."""
    s = synthetic_scene(8)
    cam = CAMERAS["original"]
    cfg = rt.make_render_config(fov=0.6, width=1920, height=1080, spt=1, render_type="phong", shading="flat", aspect_mode=1, **cam)
    frames = []
    for builder, traversal in (("device", "auto"), ("host", "auto"), ("device", "wave"), ("device", "lane"), ("device", "pool")):
        monkeypatch.setenv("RT_TRAVERSAL", traversal)
        g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10, "builder": builder})
        g.Build(s.pos)
        g.set_shading(None, s.matid, s.materials)
        img, st = g.render(cfg)
        frames.append((img, st.rays, st.shadow_rays, st.secondary_rays))
        g.close()
    monkeypatch.delenv("RT_TRAVERSAL")
    for img, rays, shadow, secondary in frames[1:]:
        assert np.array_equal(bits(img), bits(frames[0][0]))
        assert (rays, shadow, secondary) == frames[0][1:]
    assert frames[0][1] > 2 * 1920 * 1080


def test_renderer_mirror_writes_the_reference_ppm(rt, orc, scenes, tmp_path):
    """Renderer(config).RenderScene(...) with the reference's config keys: the PPM file it writes is byte-identical to
    the oracle's for an RNG-free render type (reflections included)."""
    import hashlib
    from raytracer_amd import renderer as rr
    from raytracer_amd.scene import Scene
    for name, rtype, shading in (("original", "distance", "flat"), ("sphere", "diffusion", "smooth")):
        sc = scenes[name]
        out = tmp_path / f"{name}.ppm"
        cfg = {"max_depth": 10, "samples_per_triangle": 1, "render_type": rtype, "max_distance": CAMERAS[name]["max_distance"],
               "shading_type": shading, "cull_backfaces": True, "background_color": [0.01, 0.01, 0.01],
               "acceleratied_data_structure": {"name": "octree_gpu", "max_triangles_per_BB": 16, "max_depth": 10},
               "output": {"filename": str(out)}}
        cam = rr.Camera({"pos": CAMERAS[name]["cam_pos"], "up": CAMERAS[name]["cam_up"], "dir": CAMERAS[name]["cam_dir"], "fov": 0.6,
                         "width": 100, "height": 100, "samples_per_pixel": 1})
        scene = Scene(sc.pos, sc.norm if sc.has_normals else None, sc.matid, sc.materials, has_normals=sc.has_normals)
        rr.Renderer(cfg).RenderScene(scene, cam, log=None)
        ref, _, _, _ = orc.render(sc, orc.Octree(sc, 16, 10),
                                  orc.make_cfg(fov=0.6, width=100, height=100, render_type=rtype, shading=shading, **CAMERAS[name]))
        assert hashlib.md5(out.read_bytes()).hexdigest() == orc.ppm_md5(ref)


def test_errors(rt, scenes):
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    with pytest.raises(RuntimeError, match="rt_ads_build"):
        g.trace_closest([[0, 0, 0]], [[0, 0, 1]])
    with pytest.raises(KeyError):
        rt.GpuOctree({"max_depth": 10})
    sc = scenes["original"]
    g = build(rt, sc)
    with pytest.raises(RuntimeError, match="smooth shading needs vertex normals"):
        g.render(rt.make_render_config(fov=0.6, width=8, height=8, shading="smooth", **CAMERAS["original"]))
    with pytest.raises(RuntimeError, match="samples_per_pixel"):
        g.render(rt.make_render_config(fov=0.6, width=8, height=8, spp=0, **CAMERAS["original"]))
    # C ABI argument checks
    import ctypes as C
    from raytracer_amd import _capi
    h = C.c_void_p()
    bad = _capi.AdsConfig(C.sizeof(_capi.AdsConfig), 0, 16, 10, 7)
    assert _capi.lib().rt_ads_create(C.byref(bad), C.byref(h)) == -1 and b"builder" in _capi.lib().rt_last_error()
    bad = _capi.AdsConfig(C.sizeof(_capi.AdsConfig) - 4, 0, 16, 10, 0)
    assert _capi.lib().rt_ads_create(C.byref(bad), C.byref(h)) == -1 and b"struct_size" in _capi.lib().rt_last_error()
    # limits are reported, never silently exceeded: a tree deeper than the traversal stacks allow
    # (two triangles that share one vertex: the octant around the vertex is subdivided down to max_depth)
    v = [0.3, 0.2, 0.1]
    two = np.array([[v, [1, 0, 0], [0, 1, 0]], [v, [-1, 0, 0.5], [0, -1, 0.7]]], np.float32)
    for builder in ("host", "device"):
        deep = rt.GpuOctree({"max_triangles_per_BB": 1, "max_depth": 24, "builder": builder})
        with pytest.raises(RuntimeError, match="deep"):
            deep.Build(two)
        ok = rt.GpuOctree({"max_triangles_per_BB": 1, "max_depth": 16, "builder": builder})
        ok.Build(two)
        assert ok.tree_stats()["max_depth"] == 16


@pytest.mark.parametrize("builder", ["host", "device"])
def test_build_invariant_failure_is_reported_by_both_builders(rt, builder):
    """A soup on which Octree::Build's own check fails (a triangle of a subdivided node reaches none of its octants; the
    reference prints and exits, octree.cc:124-152): both builders return the error, and the handle builds the next scene."""
    from conftest import soup_that_breaks_the_build_invariant
    pos = soup_that_breaks_the_build_invariant()
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10, "builder": builder})
    with pytest.raises(RuntimeError, match="assigned to no octant"):
        g.Build(pos)
    with pytest.raises(RuntimeError, match="rt_ads_build"):
        g.trace_closest([[0, 0, 0]], [[0, 0, 1]])
    g.Build(pos[:2000])
    t, tri = g.trace_closest(pos[:8, 0] + [0.0, 0.0, -30.0], np.tile([0.0, 0.0, 1.0], (8, 1)))
    assert t.shape == (8,)
    g.close()


def test_dropin_behind_reference_ads_interface(scenes, tmp_path):
    """oracle/_ref/ads_dropin_test links the REFERENCE's own Octree sources and this build's GpuOctree subclass
    (raytracer_amd/host/gpu_octree_ads.h) into one program and compares them through the reference's Ads
    interface.  The binary is built where the reference checkout exists and travels with the tree."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ads_dropin_test")
    assert os.path.exists(exe), "oracle/_ref/ads_dropin_test is missing: run `make -C oracle ref` where /root/reference exists"
    for name in ("original", "sphere"):
        raw = tmp_path / f"{name}.raw"
        pos = np.ascontiguousarray(scenes[name].pos, np.float32)
        with open(raw, "wb") as f:
            f.write(np.uint64(pos.shape[0]).tobytes())
            f.write(pos.tobytes())
        p = subprocess.run([exe, str(raw), "1500"], capture_output=True, text=True, timeout=100)
        assert p.returncode == 0, p.stderr[-500:]
        assert "DROPIN PASS" in p.stdout and "segments=" in p.stdout  # Search lists, PrintStats text, closest hits, any-hit flags
