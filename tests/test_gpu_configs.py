"""GPU tests of the benchmark configurations at their true parameters.

Each configuration is checked against the oracle at a size the oracle finishes in seconds, and at its full size
through properties that do not need the oracle for the whole frame: every traversal form renders the same bits, a band
split over ranks reassembles to the same frame, a band of rows equals the oracle's pixels, the ray count is stable."""
import numpy as np
import pytest

from conftest import CAMERAS

pytestmark = pytest.mark.gpu

FORMS = {
    "auto": {},
    "general": {"RT_DISABLE_SMALL_TREE": "1"},                              # any-hit rays: the clustered form
    "plain": {"RT_DISABLE_SMALL_TREE": "1", "RT_DISABLE_CLUSTERS": "1"},
    "lane": {"RT_DISABLE_SMALL_TREE": "1", "RT_TRAVERSAL": "lane"},
    "pool": {"RT_DISABLE_SMALL_TREE": "1", "RT_TRAVERSAL": "pool"},
    "full_lists": {"RT_LIGHT_CULL": "0"},                                   # no per-light lists: every any-hit pass walks everything
    "every_ray": {"RT_SKIP_UNLIT": "0"},                                    # shadow rays that cannot change the image are traced too
    "no_hit_lists": {"RT_HIT_CULL": "0"},                                   # small-tree form: the light's list whatever triangle the hits lie on
}


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def build(rt, sc, monkeypatch=None, form="auto", builder="auto"):
    if monkeypatch is not None:
        for k in ("RT_DISABLE_SMALL_TREE", "RT_TRAVERSAL", "RT_DISABLE_CLUSTERS", "RT_LIGHT_CULL", "RT_SKIP_UNLIT", "RT_HIT_CULL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in FORMS[form].items():
            monkeypatch.setenv(k, v)
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10, "builder": builder})
    g.Build(sc.pos)
    g.set_shading(sc.norm if getattr(sc, "has_normals", False) else None, sc.matid, sc.materials)
    return g


# ---- Cornell-Original, fuzzysampling 64 samples per pixel, 32 samples per light (renderer.cc:43-49, 202-247) ----------

@pytest.mark.parametrize("w,h,aspect", [(48, 27, 0), (32, 32, 0), (40, 24, 1)])
def test_original_spp64_spt32_against_the_oracle(rt, orc, scenes, monkeypatch, w, h, aspect):
    sc = scenes["original"]
    kw = dict(fov=0.6, width=w, height=h, spp=64, spt=32, render_type="blinn_phong", shading="flat", **CAMERAS["original"])
    ref, ref_tri, ref_t, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, aspect_mode=aspect, **kw))
    first = None
    for form in FORMS:
        g = build(rt, sc, monkeypatch, form)
        cfg = rt.make_render_config(aspect_mode=aspect, **kw)
        img, st = g.render(cfg)
        t, tri = g.primary_hits(cfg)
        assert np.array_equal(tri, ref_tri) and np.array_equal(bits(t), bits(ref_t)), form
        assert (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays) == \
               (ost.search_calls, ost.primary_rays, ost.secondary_rays, ost.shadow_rays), form
        assert np.abs(img - ref).max() <= 1.0 / 255.0, form
        if first is None:
            first = img
        assert np.array_equal(bits(img), bits(first)), form  # the forms agree bit for bit
    assert ost.primary_rays == 64 * w * h and ost.shadow_rays > 30 * ost.primary_rays


def test_original_spp64_spt32_full_size_properties(rt, orc, scenes, monkeypatch):
    """3840x2160 is 39 G rays per frame (1.2 s on one GPU): the full-size check renders one eighth of the bands -- the
    share of one of 8 ranks -- in two traversal forms and compares a band of it with the oracle."""
    sc = scenes["original"]
    kw = dict(fov=0.6, width=3840, height=2160, spp=64, spt=32, render_type="blinn_phong", shading="flat", **CAMERAS["original"])
    out = {}
    for form in ("auto", "general", "full_lists", "every_ray", "no_hit_lists"):
        g = build(rt, sc, monkeypatch, form)
        img = np.zeros((2160, 3840, 3), np.float32)
        _, st = g.render(rt.make_render_config(band_rank=3, band_world=8, **kw), out=img)
        out[form] = (img, st.rays, st.shadow_rays)
    for form in ("general", "full_lists", "every_ray", "no_hit_lists"):
        assert np.array_equal(bits(out["auto"][0]), bits(out[form][0])) and out["auto"][1:] == out[form][1:], form
    img = out["auto"][0]
    # band 3 = rows 24..31 belong to rank 3 of 8; rows 0..23 do not and must be untouched
    assert not img[:24].any() and img[24:32].any()
    ref, _, _, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(24, 25), **kw), want_hits=False)
    assert np.abs(img[24] - ref[24]).max() <= 1.0 / 255.0
    # every pixel sample issues its camera ray; the frame share is 1/8 of the 270 bands (34 bands for rank 3)
    assert out["auto"][1] > 64 * 3840 * 8 * 33 * 30


# ---- Cornell-Sphere, smooth Blinn-Phong, 4 samples per pixel, 32 samples per light, reflection + refraction ------------

@pytest.mark.parametrize("w,h,aspect", [(64, 36, 0), (40, 40, 0), (48, 27, 1)])
def test_sphere_spp4_spt32_against_the_oracle(rt, orc, scenes, monkeypatch, w, h, aspect):
    sc = scenes["sphere"]
    kw = dict(fov=0.6, width=w, height=h, spp=4, spt=32, render_type="blinn_phong", shading="smooth", max_depth=10, **CAMERAS["sphere"])
    ref, ref_tri, ref_t, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, aspect_mode=aspect, **kw))
    first = None
    for form in ("auto", "plain", "lane", "pool"):
        g = build(rt, sc, monkeypatch, form)
        cfg = rt.make_render_config(aspect_mode=aspect, **kw)
        img, st = g.render(cfg)
        t, tri = g.primary_hits(cfg)
        assert np.array_equal(tri, ref_tri) and np.array_equal(bits(t), bits(ref_t)), form
        assert (st.rays, st.secondary_rays, st.shadow_rays) == (ost.search_calls, ost.secondary_rays, ost.shadow_rays), form
        assert np.abs(img - ref).max() <= 1.0 / 255.0, form
        if first is None:
            first = img
        assert np.array_equal(bits(img), bits(first)), form
    if h >= w or aspect:
        assert ost.secondary_rays > 0  # reflection / refraction levels occur (in compat aspect with h < w the one scanline may miss the spheres)


def test_sphere_spp4_spt32_full_size_properties(rt, orc, scenes, monkeypatch):
    sc = scenes["sphere"]
    kw = dict(fov=0.6, width=1920, height=1080, spp=4, spt=32, render_type="blinn_phong", shading="smooth", max_depth=10,
              **CAMERAS["sphere"])
    cfg = rt.make_render_config(**kw)
    g = build(rt, sc, monkeypatch, "auto")
    full, st = g.render(cfg)
    again, st2 = g.render(cfg)
    assert np.array_equal(bits(again), bits(full)) and st2.rays == st.rays
    assert st.primary_rays == 4 * 1920 * 1080 and st.shadow_rays > 50 * st.primary_rays
    # rows 536..543 against the oracle
    ref, _, _, ost = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(536, 544), **kw), want_hits=False)
    assert np.abs(full[536:544] - ref[536:544]).max() <= 1.0 / 255.0
    # the per-lane form and the plain wave-together walk on the same frame
    for other in ("lane", "plain", "full_lists"):
        g2 = build(rt, sc, monkeypatch, other)
        img2, s2 = g2.render(cfg)
        assert np.array_equal(bits(img2), bits(full)) and s2.rays == st.rays, other
    # 4 ranks' bands
    out = np.zeros_like(full)
    rays = 0
    for r in range(4):
        _, sr = g.render(rt.make_render_config(band_rank=r, band_world=4, **kw), out=out)
        rays += sr.rays
    assert np.array_equal(bits(out), bits(full)) and rays == st.rays
    # true aspect: the whole picture (both spheres in view), forms agree, a band equals the oracle
    cfg_t = rt.make_render_config(aspect_mode=1, **kw)
    full_t, st_t = g.render(cfg_t)
    img_t, s_t = g2.render(cfg_t)
    assert np.array_equal(bits(img_t), bits(full_t)) and s_t.rays == st_t.rays and st_t.secondary_rays > 1000000
    ref, _, _, _ = orc.render(sc, orc.Octree(sc, 16, 10), orc.make_cfg(rng_mode=orc.RNG_COUNTER, aspect_mode=1, rows=(700, 702), **kw),
                              want_hits=False)
    assert np.abs(full_t[700:702] - ref[700:702]).max() <= 1.0 / 255.0


# ---- 1,310,754 triangles, Phong, hard shadows, 3840x2160 ---------------------------------------------------------------

def test_large_tree_4k_frame_properties(rt, orc, monkeypatch):
    from test_gpu_parity import synthetic_scene
    s = synthetic_scene(8)
    kw = dict(fov=0.6, width=3840, height=2160, spt=1, render_type="phong", shading="flat", aspect_mode=1, **CAMERAS["original"])
    cfg = rt.make_render_config(**kw)
    frames = {}
    for form, builder in (("auto", "device"), ("lane", "device"), ("pool", "host")):
        for k in ("RT_DISABLE_SMALL_TREE", "RT_TRAVERSAL"):
            monkeypatch.delenv(k, raising=False)
        if form != "auto":
            monkeypatch.setenv("RT_TRAVERSAL", form)
        g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10, "builder": builder})
        g.Build(s.pos)
        g.set_shading(None, s.matid, s.materials)
        img, st = g.render(cfg)
        frames[form] = (img, st.rays, st.shadow_rays, st.secondary_rays)
        if form == "auto":
            # Ads::Search on a tree so big (533,107 nodes) that the queues of 64 rays exceed the call's workspace: the
            # batch is walked in chunks, every list equals the single-ray call's and the oracle's candidate order
            rng = np.random.default_rng(5)
            rays = np.concatenate([rng.uniform(50, 500, (64, 3)), rng.normal(size=(64, 3))], axis=1).astype(np.float32)
            lists = g.search_batch(rays)
            for i in (0, 17, 62, 63):
                assert np.array_equal(lists[i], g.search_batch(rays[i:i + 1])[0]), i
            assert sum(len(x) for x in lists) > 64
            # two ranks' bands reassemble to the same frame
            out = np.zeros_like(img)
            rays = 0
            for r in range(2):
                _, sr = g.render(rt.make_render_config(band_rank=r, band_world=2, **kw), out=out)
                rays += sr.rays
            assert np.array_equal(bits(out), bits(img)) and rays == st.rays
        g.close()
    for form in ("lane", "pool"):
        assert np.array_equal(bits(frames[form][0]), bits(frames["auto"][0])), form
        assert frames[form][1:] == frames["auto"][1:], form
    assert frames["auto"][1] > 2 * 3840 * 2160
    # rows 1000..1001 (through the sphere) against the oracle
    osc = orc.Scene.from_arrays(s.pos, None, s.matid, s.materials)
    ref, _, _, ost = orc.render(osc, orc.Octree(osc, 16, 10),
                                orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(1000, 1002), **{**kw, "aspect_mode": orc.ASPECT_TRUE}), want_hits=False)
    assert np.abs(frames["auto"][0][1000:1002] - ref[1000:1002]).max() <= 1.0 / 255.0


# ---- how a frame is scheduled must not show in it -----------------------------------------------------------------------

def test_fused_tail_and_sub_frames_leave_the_frame_unchanged(rt, monkeypatch):
    """A big-tree frame whose paths are chains (no material reflects AND transmits) may run the small deep levels of
    its recursion as one fused launch, and any frame may be split into concurrent sub-frames.  Neither may change a bit
    of the image, a ray count or the depth reached: level by level, fused from the first level that qualifies, fused from
    level 1 on, and 3 / 8 sub-frames with and without the fused launch all give the same frame."""
    from test_gpu_parity import synthetic_scene
    monkeypatch.delenv("RT_CLIP", raising=False)  # (the fused launch belongs to the clipped per-lane walks, the default)
    s = synthetic_scene(6)  # 81,954 triangles: per-lane form
    kw = dict(fov=0.6, width=960, height=540, spt=1, render_type="phong", shading="flat", aspect_mode=1, **CAMERAS["original"])
    cfg = rt.make_render_config(**kw)

    def frame(tail_max_rays, streams):
        if tail_max_rays is None:
            monkeypatch.delenv("RT_TAIL_MAX_RAYS", raising=False)
        else:
            monkeypatch.setenv("RT_TAIL_MAX_RAYS", str(tail_max_rays))
        g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
        g.Build(s.pos)
        g.set_shading(None, s.matid, s.materials)
        g.set_render_streams(streams)
        img, st = g.render(cfg)
        g.close()
        return img, st

    ref, st0 = frame(0, 1)
    assert st0.launches_tail == 0 and st0.levels >= 5 and st0.secondary_rays > 10000
    for tail, streams in ((None, 1), (1 << 30, 1), (20000, 1), (0, 3), (None, 3), (1 << 30, 8)):
        img, st = frame(tail, streams)
        assert np.array_equal(bits(img), bits(ref)), (tail, streams)
        assert (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays, st.levels) == \
               (st0.rays, st0.primary_rays, st0.secondary_rays, st0.shadow_rays, st0.levels), (tail, streams)
        assert (st.launches_tail > 0) == (tail != 0), (tail, streams)
        if tail == 1 << 30 and streams == 1:
            assert st.launches_closest == 1  # level 0 only: everything below ran in the fused launch


def test_frames_in_flight_equal_blocking_frames(rt, scenes, monkeypatch):
    """rt_render_submit / rt_render_wait: frames in flight are the frames rt_render gives, bit for bit, with the same ray
    counts -- several frames of one configuration, different configurations side by side, the bands of one frame as
    separate frames in flight into one buffer, on a small tree (wave-together forms, branching recursion) and on a big one
    (per-lane form, fused tail).  The ninth frame in flight is refused, so is a ticket that is not in flight."""
    from test_gpu_parity import synthetic_scene
    from test_zz_no_alloc import _Hip
    hip = _Hip()  # device frame buffers through the HIP runtime the library is linked to

    def dev_frame(kw):
        return hip.alloc(kw["height"] * kw["width"] * 12, fill=0)

    def host(ptr, kw):
        return hip.download(ptr, np.float32, kw["height"] * kw["width"] * 3).reshape(kw["height"], kw["width"], 3)

    cases = []
    sc = scenes["sphere"]
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(sc.pos)
    g.set_shading(sc.norm, sc.matid, sc.materials)
    cases.append((g, dict(fov=0.6, width=200, height=120, spp=2, spt=5, render_type="blinn_phong", shading="smooth", **CAMERAS["sphere"]),
                  dict(fov=0.5, width=96, height=64, spt=2, render_type="diffusion", shading="flat", **CAMERAS["sphere"])))
    s = synthetic_scene(6)
    g = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10})
    g.Build(s.pos)
    g.set_shading(None, s.matid, s.materials)
    cases.append((g, dict(fov=0.6, width=640, height=360, spt=1, render_type="phong", shading="flat", aspect_mode=1, **CAMERAS["original"]),
                  dict(fov=0.4, width=128, height=200, spt=2, render_type="distance", shading="flat", **CAMERAS["original"])))
    counts = lambda st: (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays, st.levels)
    for g, kw_a, kw_b in cases:
        cfg_a, cfg_b = rt.make_render_config(**kw_a), rt.make_render_config(**kw_b)
        ref_a, st_a = g.render(cfg_a)
        ref_b, st_b = g.render(cfg_b)
        q0 = g.query_stats()["search_calls"]
        # six frames in flight, the two configurations alternating
        kws = [kw_a if i % 2 == 0 else kw_b for i in range(6)]
        bufs = [dev_frame(kw) for kw in kws]
        tickets = [g.render_submit(cfg_a if i % 2 == 0 else cfg_b, bufs[i]) for i in range(6)]
        assert sorted(tickets) == list(range(6))
        for i in (3, 0, 5, 1, 4, 2):  # retired in another order than submitted
            st = g.render_wait(tickets[i])
            ref, st_ref = (ref_a, st_a) if i % 2 == 0 else (ref_b, st_b)
            assert np.array_equal(bits(host(bufs[i], kws[i])), bits(ref)), i
            assert counts(st) == counts(st_ref), i
        assert g.query_stats()["search_calls"] == q0 + 3 * (st_a.rays + st_b.rays)
        # the bands of one frame as eight frames in flight into one buffer; a ninth is refused
        out = dev_frame(kw_a)
        tickets = [g.render_submit(rt.make_render_config(band_rank=r, band_world=8, **kw_a), out) for r in range(8)]
        with pytest.raises(RuntimeError, match="in flight already"):
            g.render_submit(cfg_b, bufs[1])
        rays = sum(g.render_wait(t).rays for t in tickets)
        assert np.array_equal(bits(host(out, kw_a)), bits(ref_a)) and rays == st_a.rays
        with pytest.raises(RuntimeError, match="no such frame in flight"):
            g.render_wait(tickets[0])
        with pytest.raises(RuntimeError, match="no such frame in flight"):
            g.render_wait(11)
        # the blocking call between frames in flight, and the instrumented pass refused for frames in flight
        t = g.render_submit(cfg_a, bufs[0])
        again, _ = g.render(cfg_b)
        with pytest.raises(RuntimeError, match="still in flight"):
            g.Build(np.zeros((1, 3, 3), np.float32))  # the scene may not change under a frame in flight
        g.render_wait(t)
        assert np.array_equal(bits(again), bits(ref_b)) and np.array_equal(bits(host(bufs[0], kw_a)), bits(ref_a))
        g.collect_search_stats(True)
        with pytest.raises(RuntimeError, match="instrumented pass"):
            g.render_submit(cfg_a, bufs[0])
        g.collect_search_stats(False)
        g.close()
        hip.free(out, *bufs)


@pytest.mark.parametrize("case", ["original", "sphere", "soup_small_tree", "soup_clustered", "big_tree_per_lane", "big_tree_own_forms_side_by_side",
                                  "big_tree_own_forms_light_passes", "original_own_forms", "soup_two_sided", "soup_many_lights"])
def test_per_light_lists_change_no_visibility_bit(rt, orc, scenes, monkeypatch, case):
    """With back-face culling on, the any-hit passes towards a light walk lists without the triangles that hold the whole
    light in front of them (they can never be accepted occluders of a ray that ends on it).  The frames must be the ones
    of the full lists bit for bit -- same image, same ray counts -- and the oracle's within the shading tolerance, on the
    two shipped scenes, on random triangle soups with emissive triangles at arbitrary angles (small-tree and clustered
    form) and on a big tree, whose per-lane walks get per-light node records and slot geometry (octants and subtrees
    left without a triangle become empty); and the lists must really leave something out.  The halves of the Cornell
    light quad share one set of forms; the *_own_forms cases give every light its own (RT_LIGHT_MERGE=0), on the big tree
    once with the lights of a hit side by side (every lane in its own light's tree) and once with one light per pass."""
    rows = None
    own_forms = "own_forms" in case
    for k in ("RT_LIGHT_MERGE", "RT_LANE_SIDE_MAX_RAYS"):
        monkeypatch.delenv(k, raising=False)
    if own_forms:
        monkeypatch.setenv("RT_LIGHT_MERGE", "0")
    if case == "big_tree_own_forms_light_passes":
        monkeypatch.setenv("RT_LANE_SIDE_MAX_RAYS", "0")
    case = {"big_tree_own_forms_side_by_side": "big_tree_per_lane", "big_tree_own_forms_light_passes": "big_tree_per_lane", "original_own_forms": "original"}.get(case, case)
    if case in ("original", "sphere"):
        sc = scenes[case]
        pos, nrm, matid, mats = sc.pos, (sc.norm if sc.has_normals else None), sc.matid, sc.materials
        cam = CAMERAS[case]
        kw = dict(fov=0.6, width=160, height=120, spt=7, render_type="blinn_phong", shading="smooth" if nrm is not None else "flat", max_depth=3, **cam)
        mb, md = 16, 10
    elif case == "big_tree_per_lane":
        from test_gpu_parity import synthetic_scene
        sy = synthetic_scene(6)  # 81,954 triangles
        pos, nrm, matid, mats = sy.pos, None, sy.matid, sy.materials
        kw = dict(fov=0.6, width=480, height=270, spt=3, render_type="phong", shading="flat", aspect_mode=1, max_depth=4, **CAMERAS["original"])
        mb, md = 16, 10
        rows = (120, 150)  # (the oracle's share: rows through the sphere)
    else:
        n_tri, mb, md = (1500, 8, 5) if case == "soup_clustered" else (150, 64, 1)
        rs = np.random.RandomState(12 if case == "soup_clustered" else 11)
        centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
        size = np.exp(rs.uniform(np.log(0.05), np.log(3.0), (n_tri, 1, 1)))
        pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
        mats = scenes["sphere"].materials
        emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
        plain = [m for m in range(mats.shape[0]) if m not in emissive]
        matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
        matid[[3, n_tri // 2, n_tri - 5]] = emissive[0]  # three lights wherever the soup put them
        if case == "soup_many_lights":
            matid[::2] = emissive[0]  # 75 lights: more than get lists of their own
        nrm = None
        kw = dict(fov=0.9, width=120, height=96, spt=5, render_type="phong", shading="flat", max_depth=2, cam_pos=[0.5, 0.3, -14.0],
                  cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=40.0)
        if case == "soup_many_lights":
            kw.update(width=48, height=40, spt=2)
        if case == "soup_two_sided":
            kw["cull"] = False  # the lists exist, the two-sided test must not use them
    cfg = rt.make_render_config(**kw)
    frames = {}
    for cull_lists in ("1", "0"):
        monkeypatch.setenv("RT_LIGHT_CULL", cull_lists)
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
        g.Build(pos)
        g.set_shading(nrm, matid, mats)
        lights, pairs, left_out = g.light_forms()
        small, clusters, _ = g.tree_forms()
        if case in ("original", "soup_small_tree", "soup_two_sided", "soup_many_lights"):
            assert small > 0  # the small-tree form serves the shadow rays
        elif case == "big_tree_per_lane":
            assert small == 0 and clusters == 0  # the per-lane form does
        else:
            assert small == 0 and clusters > 0  # the clustered form does
        if case == "soup_many_lights":
            assert (lights, pairs, left_out) == (0, 0, 0)
        elif cull_lists == "1":
            assert lights >= 2 and 0 < left_out < pairs, (lights, pairs, left_out)
        else:
            assert (lights, pairs, left_out) == (0, 0, 0)
        img, st = g.render(cfg)
        frames[cull_lists] = (img, (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays, st.levels))
        g.close()
    monkeypatch.delenv("RT_LIGHT_CULL")
    assert np.array_equal(bits(frames["1"][0]), bits(frames["0"][0]))
    assert frames["1"][1] == frames["0"][1] and frames["1"][1][3] > 10000
    osc = orc.Scene.from_arrays(pos, nrm, matid, mats)
    okw = dict(kw, aspect_mode=orc.ASPECT_TRUE) if kw.get("aspect_mode") else kw
    if rows is None:
        ref, _, _, ost = orc.render(osc, orc.Octree(osc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, **okw), want_hits=False)
        assert np.abs(frames["1"][0] - ref).max() <= 1.0 / 255.0
        assert frames["1"][1][0] == ost.search_calls
    else:
        ref, _, _, _ = orc.render(osc, orc.Octree(osc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=rows, **okw), want_hits=False)
        assert np.abs(frames["1"][0][rows[0]:rows[1]] - ref[rows[0]:rows[1]]).max() <= 1.0 / 255.0


@pytest.mark.parametrize("case", ["original", "original_integer_aspect", "sphere", "soup", "soup_layers", "soup_two_sided", "soup_far_camera", "soup_many_lights"])
def test_hit_triangle_lists_change_no_visibility_bit(rt, orc, scenes, monkeypatch, case):
    """Small-tree form, back-face culling on: a wave whose hits lie on one triangle h walks the light's list without the
    triangles that have all of h behind their plane -- a ray that starts behind a plane never meets its front.  The
    frames must be those of RT_HIT_CULL=0 bit for bit, with the same ray counts, and the oracle's within the shading
    tolerance: on Cornell-Original with the true and the reference's integer aspect ratio, on random triangle soups
    (also with a third of the triangles in common planes: hit triangles exactly IN the plane of others), with the
    two-sided test (which must not use the lists), with a camera farther away than the lists' margin allows (they must
    stay unused: same frame either way) and with more lights than get lists.  Cornell-Sphere has no small-tree form."""
    nrm = None
    mb, md = 16, 10
    if case in ("original", "original_integer_aspect", "sphere"):
        name = case.split("_")[0]
        sc = scenes[name]
        pos, matid, mats = sc.pos, sc.matid, sc.materials
        kw = dict(fov=0.6, width=200, height=150, spt=6, render_type="blinn_phong", shading="flat", max_depth=3,
                  aspect_mode=0 if case.endswith("integer_aspect") else 1, **CAMERAS[name])
    else:
        n_tri, mb, md = 150, 64, 1
        rs = np.random.RandomState(41)
        centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
        size = np.exp(rs.uniform(np.log(0.05), np.log(3.0), (n_tri, 1, 1)))
        pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
        if case == "soup_layers":
            pos[: n_tri // 3, :, 1] = np.round(pos[: n_tri // 3, :, 1])
        mats = scenes["sphere"].materials
        emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
        plain = [m for m in range(mats.shape[0]) if m not in emissive]
        matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
        matid[[3, n_tri // 2, n_tri - 5]] = emissive[0]
        if case == "soup_many_lights":
            matid[::2] = emissive[0]
        z_cam = -200.0 if case == "soup_far_camera" else -14.0  # (the soup's box has a diagonal of about 30)
        kw = dict(fov=0.1 if case == "soup_far_camera" else 0.9, width=120, height=96, spt=5, render_type="phong", shading="flat", max_depth=2,
                  cam_pos=[0.5, 0.3, z_cam], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=400.0, cull=case != "soup_two_sided")
        if case == "soup_many_lights":
            kw.update(width=48, height=40, spt=2)
    cfg = rt.make_render_config(**kw)
    frames = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RT_HIT_CULL", flag)
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
        g.Build(pos)
        g.set_shading(nrm, matid, mats)
        small, _, _ = g.tree_forms()
        pairs, left_out = g.hit_lists()
        if flag == "0" or case in ("sphere", "soup_many_lights"):
            assert (pairs, left_out) == (0, 0)
        else:
            assert small > 0 and 0 < left_out < pairs, (pairs, left_out)
        img, st = g.render(cfg)
        frames[flag] = (img, (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays, st.levels))
        g.close()
    monkeypatch.delenv("RT_HIT_CULL")
    assert np.array_equal(bits(frames["1"][0]), bits(frames["0"][0]))
    assert frames["1"][1] == frames["0"][1] and frames["1"][1][3] > 10000
    if case in ("original", "soup", "soup_layers"):
        osc = orc.Scene.from_arrays(pos, nrm, matid, mats)
        okw = dict(kw, aspect_mode=orc.ASPECT_TRUE) if kw.get("aspect_mode") else kw
        ref, _, _, ost = orc.render(osc, orc.Octree(osc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, **okw), want_hits=False)
        assert np.abs(frames["1"][0] - ref).max() <= 1.0 / 255.0
        assert frames["1"][1][0] == ost.search_calls


@pytest.mark.parametrize("factor", [1.02, 1.5, 4.0, 64.0])
def test_hit_triangle_lists_at_their_margin(rt, scenes, monkeypatch, factor):
    """The worst case for the rule behind the per-hit-triangle lists: a wall whose top edge ends only `factor` times the
    rule's margin BELOW the plane of a 20 m floor that faces up (1.9 mm for factor 1.02), lit by a light wall at the same
    height on the other side -- the shadow rays run under the floor, almost parallel to it, where the triangle test's
    determinant is rounding noise.  The floor is left out of the wall's lists; the frames must still be those of the
    light's own lists, bit for bit, one- and two-sided.  (Run by hand with RT_LIGHT_CULL_MARGIN_SCALE, which shrinks the
    margin in the library and here alike, to see how far below the margin the rule still holds.)"""
    def quad(p0, p1, p2, p3):
        return [[p0, p1, p2], [p0, p2, p3]]

    import os
    diag = float(np.sqrt(20.0 ** 2 + 20.0 ** 2 + 1.0 ** 2))
    unit = 64.0 * 2.0 ** -24 * diag * float(os.environ.get("RT_LIGHT_CULL_MARGIN_SCALE", "1"))
    margin = unit * (2.0 * 1.0 + 1.0) + unit * (10.0 * 1.0 + 5.0)  # m_j of the floor + m_hit of the wall (right triangles: k = 1)
    g = factor * margin
    tris = []
    tris += quad([-10, 0, -10], [-10, 0, 10], [10, 0, 10], [10, 0, -10])      # the floor, normal +y
    tris += quad([8, -1, -2], [8, -g, -2], [8, -g, 2], [8, -1, 2])            # the light wall, facing -x, top edge g below the floor's plane
    tris += quad([-8, -1, -2], [-8, -1, 2], [-8, -g, 2], [-8, -g, -2])        # the wall that is looked at, facing +x
    pos = np.asarray(tris, np.float32)
    assert pos[4:6, :, 1].max() == np.float32(-g)
    mats = scenes["sphere"].materials
    emissive = next(m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0))
    plain = next(m for m in range(mats.shape[0]) if not np.any(mats[m, 12:15] != 0) and not np.any(mats[m, 6:12] != 0))
    matid = np.full(len(pos), plain, np.int32)
    matid[2:4] = emissive
    frames = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RT_HIT_CULL", flag)
        a = rt.GpuOctree({"max_triangles_per_BB": 4, "max_depth": 1})  # (one subdivision: a root leaf has no lists to thin out)
        a.Build(pos)
        a.set_shading(None, matid, mats)
        small, _, _ = a.tree_forms()
        assert small > 0
        if flag == "1":
            pairs, left_out = a.hit_lists()
            assert left_out >= 4  # at least the two floor triangles, for both triangles of the wall
        imgs = []
        # the camera looks at the wall through a keyhole: hit points within a few margins of its top edge, then within a
        # centimetre, then the whole wall
        for fov, cam_y in ((2.0 * np.arctan(4.0 * g / 8.0), -2.0 * g), (2.0 * np.arctan(0.01 / 8.0), -g - 0.005), (0.25, -0.5)):
            for two_sided in (False, True):
                cfg = rt.make_render_config(cam_pos=[0.0, float(cam_y), 0.0], cam_up=[0.0, 1.0, 0.0], cam_dir=[-1.0, 0.0, 0.0], fov=float(fov), width=96, height=96,
                                            spt=16, render_type="blinn_phong", shading="flat", cull=not two_sided, max_distance=40.0, aspect_mode=1, max_depth=0)
                img, st = a.render(cfg)
                imgs.append((img, st.rays, st.shadow_rays))
        frames[flag] = imgs
        a.close()
    monkeypatch.delenv("RT_HIT_CULL")
    lit = 0
    for k, ((x, rx, sx), (y, ry, sy)) in enumerate(zip(frames["1"], frames["0"])):
        assert np.array_equal(bits(x), bits(y)) and (rx, sx) == (ry, sy), k
        lit += int((x > 1e-6).any())
    assert lit >= 3  # the light reaches the wall: a visibility bit that differed would show in the pixels


@pytest.mark.parametrize("walk", ["lane", "pool"])
@pytest.mark.parametrize("case", ["original_blinn", "original_phong", "sphere_flat_blinn", "sphere_flat_phong", "sphere_smooth", "big_tree_phong",
                                  "soup_phong", "soup_blinn_two_sided", "soup_odd_materials", "soup_huge_light"])
def test_shadow_rays_without_effect_change_no_pixel(rt, orc, scenes, monkeypatch, case, walk):
    """A light sample whose contribution is a zero whether it is visible or not -- the light faces away from the hit, or
    the surface faces away from the light and has no specular colour -- gets no shadow ray (flat shading, finite and
    moderate lights and materials, no material emission under Phong).  The frames must be those of RT_SKIP_UNLIT=0, which
    traces every ray, bit for bit, with the same ray counts (the counts are the reference's, skipped rays included):
    on the shipped scenes with both lit render types (the Sphere scene has reflecting and refracting materials, so the
    deeper levels see the rule as well; with smooth shading only the light-faces-away half of the rule applies), on a big tree (per-lane walks and the
    fused tail), on triangle soups with lights at arbitrary angles, two-sided, and with materials and lights the rule
    must keep away from: infinite and NaN colours, a negative shininess, an emission that overflows a product.  The rule
    is applied where every lane walks its own ray (per-lane and task-pool forms, the fused tail), so the small scenes are
    made to use those forms."""
    from test_gpu_parity import synthetic_scene
    monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    monkeypatch.setenv("RT_TRAVERSAL", walk)
    nrm = None
    mb, md = 16, 10
    if case.startswith("original") or case.startswith("sphere"):
        name = case.split("_")[0]
        sc = scenes[name]
        pos, matid, mats = sc.pos, sc.matid, sc.materials
        smooth = case == "sphere_smooth"
        if smooth:
            nrm = sc.norm
        kw = dict(fov=0.6, width=200, height=150, spt=5, render_type="phong" if case.endswith("phong") else "blinn_phong",
                  shading="smooth" if smooth else "flat", max_depth=4, aspect_mode=1, **CAMERAS[name])  # (true aspect ratio: floor and ceiling in view)
    elif case == "big_tree_phong":
        sy = synthetic_scene(6)  # 81,954 triangles
        pos, matid, mats = sy.pos, sy.matid, sy.materials
        kw = dict(fov=0.6, width=480, height=270, spt=3, render_type="phong", shading="flat", aspect_mode=1, max_depth=4, **CAMERAS["original"])
    else:
        n_tri = 400
        rs = np.random.RandomState(31)
        centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
        size = np.exp(rs.uniform(np.log(0.05), np.log(3.0), (n_tri, 1, 1)))
        pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
        mats = scenes["sphere"].materials.copy()
        emissive = [m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0)]
        plain = [m for m in range(mats.shape[0]) if m not in emissive]
        if case == "soup_odd_materials":  # (a material is 18 floats: diffuse 3:6, specular 6:9, emission 12:15, shininess 15)
            mats[plain[0], 15] = -2.0
            mats[plain[1], 3] = np.inf
            mats[plain[2], 7] = np.nan
            mats[plain[3], 15] = 3e7
        if case == "soup_huge_light":
            mats[emissive[0], 12:15] = 3e35
        matid = np.asarray(plain, np.int32)[rs.randint(0, len(plain), n_tri)]
        matid[[3, n_tri // 2, n_tri - 5]] = emissive[0]
        mb, md = 8, 4
        kw = dict(fov=0.9, width=160, height=120, spt=4, render_type="blinn_phong" if "blinn" in case else "phong", shading="flat", max_depth=3,
                  cam_pos=[0.5, 0.3, -14.0], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=40.0, cull="two_sided" not in case)
    cfg = rt.make_render_config(**kw)
    frames = {}
    for skip in ("1", "0"):
        monkeypatch.setenv("RT_SKIP_UNLIT", skip)
        g = rt.GpuOctree({"max_triangles_per_BB": mb, "max_depth": md})
        g.Build(pos)
        g.set_shading(nrm, matid, mats)
        img, st = g.render(cfg)
        frames[skip] = (img, (st.rays, st.primary_rays, st.secondary_rays, st.shadow_rays, st.levels), st.shadow_rays_not_traced)
        g.close()
    monkeypatch.delenv("RT_SKIP_UNLIT")
    assert np.array_equal(bits(frames["1"][0]), bits(frames["0"][0]))
    assert frames["1"][1] == frames["0"][1] and frames["1"][1][3] > 10000
    # the rule is at work where it may be, and only there
    assert frames["0"][2] == 0
    if case == "soup_huge_light":
        assert frames["1"][2] == 0
    else:
        assert 0.02 * frames["1"][1][3] < frames["1"][2] < frames["1"][1][3], (frames["1"][2], frames["1"][1][3])
    if case in ("original_phong", "soup_phong") and walk == "lane":  # ... and the oracle's frame, which knows no such rule
        osc = orc.Scene.from_arrays(pos, nrm, matid, mats)
        okw = dict(kw, aspect_mode=orc.ASPECT_TRUE) if kw.get("aspect_mode") else kw
        ref, _, _, ost = orc.render(osc, orc.Octree(osc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, **okw), want_hits=False)
        assert np.abs(frames["1"][0] - ref).max() <= 1.0 / 255.0
        assert frames["1"][1][0] == ost.search_calls


@pytest.mark.parametrize("factor", [1.02, 1.5, 4.0, 64.0])
def test_per_light_lists_at_their_margin(rt, scenes, monkeypatch, factor):
    """The worst case for the rule behind the per-light lists: a light that hovers only `factor` times the rule's rounding
    margin in front of a large triangle pair (0.3 mm above a 20 m floor for factor 1.02), seen from surface points at
    about the same height -- so the shadow rays run almost parallel to that floor, the regime where the triangle test's
    determinant is rounding noise -- and from points well above and below it.  The floor is left out of the light's lists;
    the frame must still be the one the full lists give, bit for bit.  (The test has teeth: run by hand with
    RT_LIGHT_CULL_MARGIN_SCALE, which shrinks the margin in the library and here alike, it still passes at 1/256 of the
    margin and fails at 1/4096.)"""
    def quad(p0, p1, p2, p3):
        return [[p0, p1, p2], [p0, p2, p3]]

    diag = float(np.sqrt(20.0 ** 2 + 20.0 ** 2 + 2.0 ** 2))
    import os
    margin = 64.0 * 2.0 ** -24 * (2.0 * 1.0 + 1.0) * diag  # right triangles: |e1||e2| / |e1 x e2| = 1
    margin *= float(os.environ.get("RT_LIGHT_CULL_MARGIN_SCALE", "1"))  # (a manual experiment: how far below the rule's margin does it hold?)
    h = factor * margin
    tris = []
    tris += quad([-10, 0, -10], [-10, 0, 10], [10, 0, 10], [10, 0, -10])              # the floor, normal +y
    tris += quad([-2, h, -2], [-2, h, 2], [2, h, 2], [2, h, -2])                        # the light, just above it, facing up
    for sx in (-1.0, 1.0):                                                             # walls whose points see the light at grazing angles
        tris += quad([9 * sx, -1, -10], [9 * sx, 1, -10], [9 * sx, 1, 10], [9 * sx, -1, 10])
        tris += quad([-10, -1, 9 * sx], [10, -1, 9 * sx], [10, 1, 9 * sx], [-10, 1, 9 * sx])
    pos = np.asarray(tris, np.float32)
    mats = scenes["sphere"].materials
    emissive = next(m for m in range(mats.shape[0]) if np.any(mats[m, 12:15] != 0))
    plain = next(m for m in range(mats.shape[0]) if not np.any(mats[m, 12:15] != 0) and not np.any(mats[m, 6:12] != 0))
    matid = np.full(len(pos), plain, np.int32)
    matid[2:4] = emissive
    frames = {}
    for cull_lists in ("1", "0"):
        monkeypatch.setenv("RT_LIGHT_CULL", cull_lists)
        g = rt.GpuOctree({"max_triangles_per_BB": 4, "max_depth": 3})  # (a root leaf has no lists to thin out)
        g.Build(pos)
        g.set_shading(None, matid, mats)
        if cull_lists == "1":
            lights, pairs, left_out = g.light_forms()
            assert lights == 2 and left_out >= 4  # the two floor triangles, for both light triangles
        imgs = []
        # the camera looks at a wall through a keyhole: hit points within a few margins of the light's height, then
        # within a centimetre, then the whole wall
        for fov, cam_y in ((2.0 * np.arctan(8.0 * h / 9.0), h), (2.0 * np.arctan(0.01 / 9.0), 0.5 * h), (0.2, 0.3), (0.2, -0.3)):
            for cam_dir in ([1.0, 0.0, 0.0], [0.0, 0.0, -1.0]):
                for two_sided in (False, True):
                    cfg = rt.make_render_config(cam_pos=[0.0, float(cam_y), 0.0], cam_up=[0.0, 1.0, 0.0], cam_dir=cam_dir, fov=float(fov), width=96, height=96,
                                                spt=16, render_type="blinn_phong", shading="flat", cull=not two_sided, max_distance=40.0,
                                                aspect_mode=1, max_depth=0)
                    img, st = g.render(cfg)
                    imgs.append((img, st.rays, st.shadow_rays))
        frames[cull_lists] = imgs
        g.close()
    monkeypatch.delenv("RT_LIGHT_CULL")
    lit = 0
    for k, ((a, ra, sa), (b, rb, sb)) in enumerate(zip(frames["1"], frames["0"])):
        assert np.array_equal(bits(a), bits(b)) and (ra, sa) == (rb, sb), k
        lit += int((a > 1e-6).any())
    assert lit >= 8  # the light reaches the walls: a visibility bit that differed would show in the pixels
