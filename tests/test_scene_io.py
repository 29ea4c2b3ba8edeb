"""CPU tests of the host-side scene input / image output and of the Renderer mirror's config handling."""
import hashlib
import os

import numpy as np
import pytest

from raytracer_amd import renderer as rr

OBJ = """# a box-ish test scene: quads (both split cases), triangles, relative and absolute indices, v//vn and v/vt/vn
mtllib test.mtl
v 0 0 0
v 4 0 0
v 4 1 0
v 0 1 0
v 0 0 3
v 1 0 3
v 1 5 3   # comment after data
v 0 5 3
vn 0 0 1
vn 0 0 -1
vt 0.5 0.5
g wide
usemtl red
f -8 -7 -6 -5
g tall
usemtl glow
f 5//1 6//1 7//1 8//1
s 1
f 1/1/2 2/1/2 3/1/2
usemtl red
f 1 3 4
"""
MTL = """newmtl red
  Ka 0.1 0.2 0.3
  Kd 0.63 0.065 0.05
  Ks 0.5 0.5 0.5
  Ns 10.0000
  Ni 1.5
newmtl glow
\tKd 0.78 0.78 0.78
    Ke 17 12 4
  Tf 1.0 0.9 0.8
  d 0.5
"""


def write_scene(tmp_path):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    return str(tmp_path) + "/"


def test_obj_loader_matches_oracle_loader(orc, tmp_path):
    base = write_scene(tmp_path)
    s = rr.load_scene({"basepath": base, "triangulate": True}, {"source_file": "test.obj"})
    o = orc.Scene.from_obj(base + "test.obj", base)
    assert s.n_triangles == o.n_tri == 6
    assert np.array_equal(s.pos, o.pos) and np.array_equal(s.norm, o.norm)
    assert np.array_equal(s.matid, o.matid) and np.array_equal(s.materials, o.materials)
    assert list(s.light_triangles()) == list(o.light_tri) == [2, 3, 4]
    # quad 1 (4 x 1): both diagonals have the same length -> the "else" split [0,1,3],[1,2,3]
    assert np.array_equal(s.pos[0], [[0, 0, 0], [4, 0, 0], [0, 1, 0]]) and np.array_equal(s.pos[1], [[4, 0, 0], [4, 1, 0], [0, 1, 0]])
    # material defaults: shininess = ior = dissolve = 1 unless set
    assert s.materials[1, 15] == 1.0 and s.materials[1, 16] == 1.0 and s.materials[1, 17] == 0.5
    assert np.allclose(s.materials[1, 9:12], [1.0, 0.9, 0.8]) and s.materials[0, 16] == 1.5


def test_obj_loader_errors(tmp_path):
    base = write_scene(tmp_path)
    with pytest.raises(RuntimeError, match="Failed to load/parse .obj"):
        rr.load_scene({"basepath": base, "triangulate": True}, {"source_file": "missing.obj"})
    (tmp_path / "nomat.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(RuntimeError, match="without a valid material"):
        rr.load_scene({"basepath": base, "triangulate": True}, {"source_file": "nomat.obj"})
    with pytest.raises(KeyError):
        rr.load_scene({"triangulate": True}, {"source_file": "test.obj"})


@pytest.mark.skipif(not os.path.isdir("/root/reference/res"), reason="reference checkout absent")
def test_obj_loader_on_shipped_scenes(scenes):
    for key, fn in (("original", "CornellBox-Original.obj"), ("sphere", "CornellBox-Sphere.obj")):
        s = rr.load_scene({"basepath": "/root/reference/res/", "triangulate": True}, {"source_file": fn})
        d = scenes[key]
        assert np.array_equal(s.pos, d.pos) and np.array_equal(s.matid, d.matid) and np.array_equal(s.materials, d.materials)
        assert s.has_normals == d.has_normals
        if d.has_normals:
            assert np.array_equal(s.norm, d.norm)


def test_ppm_writer_matches_reference_format(orc, tmp_path):
    rs = np.random.RandomState(0)
    img = rs.rand(13, 17, 3).astype(np.float32)
    img[0, 0] = [0.0, 1.0, 0.999999]
    img[0, 1] = [0.00390625, 0.5, 0.99609375]
    path = tmp_path / "out.ppm"
    rr.write_ppm(path, img)
    data = path.read_bytes()
    assert data == orc.ppm_bytes(img)
    assert data.startswith(b"P3\n17 13\n255\n0 255 255\n")
    assert hashlib.md5(data).hexdigest() == orc.ppm_md5(img)


def test_renderer_config_errors():
    good = {"max_depth": 10, "samples_per_triangle": 4, "render_type": "phong", "max_distance": 6.0, "shading_type": "flat",
            "cull_backfaces": True, "background_color": [0.01, 0.01, 0.01],
            "acceleratied_data_structure": {"name": "octree_gpu", "max_triangles_per_BB": 16, "max_depth": 10},
            "output": {"filename": "x.ppm"}}
    r = rr.Renderer(good)
    assert r.acceleration_data_structure_ == 3 and r.samples_per_triangle_ == 4
    for key, value, msg in (("render_type", "raster", "Invalid render type"), ("shading_type", "gouraud", "Invalid shading type")):
        with pytest.raises(RuntimeError, match=msg):
            rr.Renderer(dict(good, **{key: value}))
    with pytest.raises(RuntimeError, match="Invalid data structure"):
        rr.Renderer(dict(good, acceleratied_data_structure={"name": "kdtree", "max_triangles_per_BB": 16, "max_depth": 10}))
    with pytest.raises(KeyError):
        rr.Renderer({k: v for k, v in good.items() if k != "max_distance"})
    assert rr.Renderer(dict(good, acceleratied_data_structure=dict(good["acceleratied_data_structure"], name="octree"))) \
        .acceleration_data_structure_ == 1
    # the optional "builder" key of the GPU data structure is validated before any device is touched
    from raytracer_amd.ads import GpuOctree
    with pytest.raises(RuntimeError, match="Invalid builder"):
        GpuOctree(dict(good["acceleratied_data_structure"], builder="fpga"))
    with pytest.raises(KeyError):
        GpuOctree({"max_depth": 10})
