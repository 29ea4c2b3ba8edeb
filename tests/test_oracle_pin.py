"""Pins the oracle (oracle/rt_oracle.c, a CPU restatement) to the unmodified reference.

tests/golden/ref_kats.json holds, per known-answer configuration, the md5 of the PPM written by the
reference binary and the counters it printed (produced by oracle/gen_ref_kats.py in a container that
has the reference checkout).  The oracle, run in its sequential legacy-rand() mode on the committed
scene dumps, must reproduce every one of them exactly."""
import json
import os

import pytest

from conftest import CAMERAS, GOLDEN

with open(os.path.join(GOLDEN, "ref_kats.json")) as f:
    KATS = json.load(f)

FAST = [k for k in sorted(KATS) if k not in ("K8", "S8")]


def synthetic_scene(orc, scenes, levels):
    """The icosphere-in-Cornell-box scene of raytracer_amd/synthetic.py as an oracle scene."""
    from raytracer_amd.scene import Scene
    from raytracer_amd.synthetic import cornell_with_icosphere
    box = Scene.load_npz(os.path.join(GOLDEN, "scenes", "cornell_original.npz"))
    s = cornell_with_icosphere(box, levels)
    return orc.Scene.from_arrays(s.pos, None, s.matid, s.materials)


def run_kat(orc, scenes, name):
    k = KATS[name]
    sp = k["spec"]
    if sp["scene"] == "synthetic":
        sc = synthetic_scene(orc, scenes, sp["levels"])
        sp = dict(sp, scene="original")  # same camera as the Cornell box
    else:
        sc = scenes[sp["scene"]]
    tree = orc.Octree(sc, sp.get("tri_per_bb", 16), sp.get("ads_depth", 10))
    cfg = orc.make_cfg(fov=0.6, width=sp["w"], height=sp["h"], spp=sp.get("spp", 1), spt=sp.get("spt", 1),
                       render_type=sp["render_type"], shading=sp["shading"], cull=sp.get("cull", True), ads=sp["ads"],
                       rng_mode=orc.RNG_LEGACY, **CAMERAS[sp["scene"]])
    rgb, tri, t, st = orc.render(sc, tree, cfg)
    return k, sp, tree, rgb, st


@pytest.mark.parametrize("name", FAST + [pytest.param("K8", marks=pytest.mark.slow), pytest.param("S8", marks=pytest.mark.slow)])
def test_reference_image_and_counters(orc, scenes, name):
    k, sp, tree, rgb, st = run_kat(orc, scenes, name)
    assert orc.ppm_md5(rgb) == k["ppm_md5"]
    assert st.search_calls == k["search_calls"]
    assert st.search_returned == k["search_returned"]
    assert st.tri_tests == k["tri_tests"]
    if sp["ads"] == "octree":
        assert st.search_nodes == k["search_nodes"]
        assert st.search_leaves == k["search_leaves"]
        ts = tree.stats()
        for key in ("nodes", "leaves", "max_depth", "max_tris_in_leaf"):
            assert ts[key] == k[key], key
        assert abs(ts["avg_leaf_depth"] - k["avg_leaf_depth"]) < 1e-4
        assert abs(ts["avg_tris_in_leaf"] - k["avg_tris_in_leaf"]) < 1e-3


@pytest.mark.parametrize("name", ["K2", "K3", "K4", "K5", "K11"])
def test_reference_binary_on_scene_arrays(scenes, name):
    """oracle/ref_runner.py feeds the reference binary a scene written from the committed arrays (not the reference's
    own resource files, which do not travel).  It must print and write what it did for its own files: same PPM md5,
    same counters -- so timing it on another host measures the reference on the real scene."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
    import ref_runner
    if not ref_runner.available():
        pytest.skip("oracle/_ref/raytracer_ref has not been built")
    k = KATS[name]
    sp = k["spec"]
    sc = scenes[sp["scene"]]
    cam = CAMERAS[sp["scene"]]
    out = ref_runner.run(sc.pos, sc.norm if sc.has_normals else None, sc.matid, sc.materials, cam["cam_pos"], cam["cam_up"],
                         cam["cam_dir"], 0.6, sp["w"], sp["h"], spp=sp.get("spp", 1), spt=sp.get("spt", 1),
                         render_type=sp["render_type"], shading=sp["shading"], cull=sp.get("cull", True),
                         max_distance=cam["max_distance"], tri_per_bb=sp.get("tri_per_bb", 16), ads_depth=sp.get("ads_depth", 10))
    assert out["ppm_md5"] == k["ppm_md5"]
    assert out["search_calls"] == k["search_calls"] and out["tri_tests"] == k["tri_tests"]
    assert out["nodes"] == k["nodes"] and out["leaves"] == k["leaves"]


def test_published_tree_shape(orc, scenes):
    """report/report.tex:178-186 of the reference: 1,584 nodes / 1,300 leaves / 5.22 / 34 / 7.03."""
    ts = orc.Octree(scenes["sphere"], 16, 10).stats()
    assert (ts["nodes"], ts["leaves"], ts["max_tris_in_leaf"], ts["max_depth"]) == (1584, 1300, 34, 10)
    assert round(ts["avg_leaf_depth"], 2) == 5.22 and round(ts["avg_tris_in_leaf"], 2) == 7.03


@pytest.mark.skipif(not os.path.isdir("/root/reference/res"), reason="reference checkout absent")
def test_scene_dumps_match_obj_files(orc, scenes):
    import numpy as np
    for key, fn in (("original", "CornellBox-Original.obj"), ("sphere", "CornellBox-Sphere.obj")):
        s = orc.Scene.from_obj("/root/reference/res/" + fn, "/root/reference/res/")
        d = scenes[key]
        assert np.array_equal(s.pos, d.pos) and np.array_equal(s.matid, d.matid)
        assert np.array_equal(s.materials, d.materials) and np.array_equal(s.norm, d.norm)
        assert list(s.light_tri) == list(d.light_tri)
