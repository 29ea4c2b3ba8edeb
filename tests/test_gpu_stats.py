"""GPU tests of the statistics contract: the counters the reference prints (Octree::PrintStats, octree.cc:209-224, and
"Ray-triangle collision count", renderer.cc:60) produced on the device in the reference's definitions.

Pins: tests/golden/ref_kats.json holds what the unmodified reference binary printed; configurations with area lights
(whose random stream the reference cannot share) are compared with the oracle run with the same counter generator."""
import io
import json
import os

import numpy as np
import pytest

from conftest import CAMERAS, GOLDEN

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


def _kats():
    with open(os.path.join(GOLDEN, "ref_kats.json")) as f:
        kats = json.load(f)
    return {k: v for k, v in kats.items() if v["spec"]["render_type"] in ("distance", "diffusion") and v["spec"]["ads"] == "octree"}


@pytest.mark.parametrize("name", sorted(_kats()))
def test_counters_equal_the_reference_binary(rt, scenes, name):
    """Render types without random numbers: every counter the reference binary printed for the frame."""
    from test_gpu_parity import synthetic_scene
    k = _kats()[name]
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
    g.collect_search_stats(True)
    cfg = rt.make_render_config(fov=0.6, width=sp["w"], height=sp["h"], spp=sp.get("spp", 1), spt=sp.get("spt", 1),
                                render_type=sp["render_type"], shading=sp["shading"], cull=sp.get("cull", True), **cam)
    img, st = g.render(cfg)
    q = g.query_stats()
    assert q["search_calls"] == k["search_calls"] == q["counted_rays"]
    assert q["search_node_count"] == k["search_nodes"]
    assert q["search_leaves_visited"] == k["search_leaves"]
    assert q["search_return_count"] == k["search_returned"]
    assert q["ray_triangle_tests"] == k["tri_tests"]
    # the instrumented pass does not change the picture
    g.collect_search_stats(False)
    img2, _ = g.render(cfg)
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))
    q2 = g.query_stats()
    assert q2["search_calls"] == 2 * k["search_calls"] and q2["counted_rays"] == k["search_calls"]


STAT_CASES = [
    # scene, render_type, shading, w, h, spt, spp, cull, octree
    ("original", "blinn_phong", "flat", 96, 96, 4, 1, True, (16, 10)),
    ("original", "phong", "flat", 40, 56, 5, 2, False, (8, 2)),
    ("sphere", "phong", "smooth", 64, 64, 2, 1, True, (16, 10)),
    ("sphere", "blinn_phong", "smooth", 48, 40, 3, 3, True, (4, 3)),
]


@pytest.mark.parametrize("case", STAT_CASES, ids=[f"{c[0]}_{c[1]}_{i}" for i, c in enumerate(STAT_CASES)])
@pytest.mark.parametrize("mode", ["auto", "lane"])
def test_counters_with_shadow_rays_equal_the_oracle(rt, orc, scenes, monkeypatch, case, mode):
    """Area lights: shadow rays stop counting triangle tests at their first occluder in candidate-list order
    (renderer.cc:305).  Same counter generator on both sides, so every counter must be equal."""
    if mode != "auto":
        monkeypatch.setenv("RT_TRAVERSAL", mode)
        monkeypatch.setenv("RT_DISABLE_SMALL_TREE", "1")
    scene, rtype, shading, w, h, spt, spp, cull, (mb, md) = case
    sc = scenes[scene]
    g = build(rt, sc, mb, md)
    g.collect_search_stats(True)
    kw = dict(fov=0.6, width=w, height=h, spt=spt, spp=spp, render_type=rtype, shading=shading, cull=cull, **CAMERAS[scene])
    ref, _, _, ost = orc.render(sc, orc.Octree(sc, mb, md), orc.make_cfg(rng_mode=orc.RNG_COUNTER, **kw), want_hits=False)
    img, st = g.render(rt.make_render_config(**kw))
    q = g.query_stats()
    assert st.rays == ost.search_calls == q["search_calls"] == q["counted_rays"]
    assert q["search_node_count"] == ost.search_nodes
    assert q["search_leaves_visited"] == ost.search_leaves
    assert q["search_return_count"] == ost.search_returned
    assert q["slab_tests"] == ost.slab_tests
    assert q["ray_triangle_tests"] == ost.tri_tests
    assert ost.tri_tests < ost.search_returned  # some shadow ray did stop early
    assert np.abs(img - ref).max() <= 1.0 / 255.0


def test_counters_of_the_batched_calls_and_of_search(rt, orc, scenes):
    from test_gpu_parity import random_rays
    sc = scenes["sphere"]
    g = build(rt, sc)
    ot = orc.Octree(sc, 16, 10)
    o, d = random_rays(sc, 5000, 41)
    for cull in (True, False):
        g.reset_query_stats()
        g.collect_search_stats(True)
        ost = orc.Stats()
        t_ref, tri_ref = ot.trace_closest(o, d, cull, stats=ost)
        t, tri = g.trace_closest(o, d, cull)
        assert np.array_equal(tri, tri_ref)
        hit = tri_ref >= 0
        P = (o[hit] + d[hit] * t_ref[hit, None]).astype(np.float32)
        L = np.tile(sc.pos[sc.light_tri[0]].mean(0).astype(np.float32), (P.shape[0], 1))
        assert np.array_equal(ot.trace_any(P, L, cull, stats=ost), g.trace_any(P, L, cull))
        q = g.query_stats()
        assert q["search_calls"] == ost.search_calls == q["counted_rays"] == len(o) + len(P)
        assert (q["search_node_count"], q["search_leaves_visited"], q["search_return_count"], q["slab_tests"], q["ray_triangle_tests"]) == \
               (ost.search_nodes, ost.search_leaves, ost.search_returned, ost.slab_tests, ost.tri_tests)
    # Search itself always counts (it walks like the reference): single rays and a batch
    g.reset_query_stats()
    g.collect_search_stats(False)
    ost = orc.Stats()
    for i in range(40):
        assert np.array_equal(g.Search(np.concatenate([o[i], d[i]])), ot.search(o[i], d[i], stats=ost))
    got = g.search_batch(np.concatenate([o[40:700], d[40:700]], axis=1))
    for i in range(40, 700):
        assert np.array_equal(got[i - 40], ot.search(o[i], d[i], stats=ost))
    q = g.query_stats()
    assert (q["search_calls"], q["search_node_count"], q["search_leaves_visited"], q["search_return_count"]) == \
           (ost.search_calls, ost.search_nodes, ost.search_leaves, ost.search_returned)
    # PrintStats: the reference's twelve lines (octree.cc:211-223), in its order
    buf = io.StringIO()
    g.PrintStats(buf)
    labels = [ln.split(":")[0].strip() for ln in buf.getvalue().splitlines()]
    assert labels == ["Octree stats", "Max depth", "Nodes count", "Leaf nodes count", "Average depth of leaf nodes",
                      "Max triangles in leaf nodes", "Average triangles in leaf nodes", "Search method call count",
                      "Search node count", "Search time", "Search return count", "Average search return count",
                      "Search leaves visited"]
    assert f"Search node count: {ost.search_nodes}" in buf.getvalue()


def test_search_with_long_candidate_lists(rt, orc, scenes):
    """A root-leaf tree returns every triangle for every ray: lists longer than the per-ray region of the one-launch
    form (1024 entries) take the two-pass form; both must give the reference's lists and counters."""
    sc = scenes["sphere"]
    g = build(rt, sc, 5000, 10)  # 2,188 triangles <= 5,000: the root is a leaf
    ot = orc.Octree(sc, 5000, 10)
    o, d = np.zeros((3, 3), np.float32), np.eye(3, dtype=np.float32)
    ost = orc.Stats()
    for i in range(3):
        assert np.array_equal(g.Search(np.concatenate([o[i], d[i]])), ot.search(o[i], d[i], stats=ost))
    got = g.search_batch(np.concatenate([o, d], axis=1))
    for i in range(3):
        assert np.array_equal(got[i], np.arange(sc.n_tri))
        ot.search(o[i], d[i], stats=ost)
    q = g.query_stats()
    assert (q["search_calls"], q["search_node_count"], q["search_leaves_visited"], q["search_return_count"]) == \
           (ost.search_calls, ost.search_nodes, ost.search_leaves, ost.search_returned) == (6, 6, 6, 6 * sc.n_tri)
