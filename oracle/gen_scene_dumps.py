#!/usr/bin/env python3
"""Parse the reference's two shipped OBJ scenes with the oracle's loader and store the resulting
triangle / material arrays as tests/golden/scenes/*.npz (data only: positions, normals, material ids,
material table).  Runs only where the reference checkout exists."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as orc

RES = os.environ.get("RT_REF_RES", "/root/reference/res/")
OUT = os.path.join(HERE, "..", "tests", "golden", "scenes")
for name, f in (("cornell_original", "CornellBox-Original.obj"), ("cornell_sphere", "CornellBox-Sphere.obj")):
    s = orc.Scene.from_obj(os.path.join(RES, f), RES)
    s.save_npz(os.path.join(OUT, name + ".npz"))
    print(name, s.n_tri, "triangles", s.n_mat, "materials", s.n_light, "lights", "normals" if s.has_normals else "no normals")
