"""Runs the unmodified reference binary (oracle/_ref/raytracer_ref, built by oracle/Makefile from the sources under
/root/reference) on a scene given as arrays.  TEST INFRASTRUCTURE ONLY: used by tests and by bench.py's cpu_baseline
leg to time the real reference on the same host.

The binary travels with the repository snapshot, the reference's resource files do not, so the scene is written
here as a temporary Wavefront OBJ / MTL pair: one `v` (and `vn`) line per triangle corner, float32 values printed with
9 significant digits (they read back exactly), one material per row of the 18-float material table
(ambient, diffuse, specular, transmittance, emission, shininess, ior, dissolve).  Triangle order, values and the
material ids are what the reference's loader then hands to its renderer.
"""
import hashlib
import json
import os
import re
import subprocess
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "raytracer_ref")

STAT_PATTERNS = {
    "search_calls": r"Search method call count: (\d+)",
    "tri_tests": r"Ray-triangle collision count: (\d+)",
    "nodes": r"Nodes count: (\d+)",
    "leaves": r"Leaf nodes count: (\d+)",
}


def available():
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def write_scene(dirname, pos, norm, matid, materials, stem="scene"):
    """pos, norm: (n, 3, 3) float32 (norm may be None); matid: (n,); materials: (m, 18).  Returns the OBJ file name."""
    g = lambda x: "%.9g" % float(x)
    with open(os.path.join(dirname, stem + ".mtl"), "w") as f:
        for i, m in enumerate(materials):
            f.write(f"newmtl m{i}\n")
            f.write("Ka %s %s %s\n" % tuple(g(x) for x in m[0:3]))
            f.write("Kd %s %s %s\n" % tuple(g(x) for x in m[3:6]))
            f.write("Ks %s %s %s\n" % tuple(g(x) for x in m[6:9]))
            f.write("Tf %s %s %s\n" % tuple(g(x) for x in m[9:12]))
            f.write("Ke %s %s %s\n" % tuple(g(x) for x in m[12:15]))
            f.write(f"Ns {g(m[15])}\nNi {g(m[16])}\nd {g(m[17])}\n")
    with open(os.path.join(dirname, stem + ".obj"), "w") as f:
        f.write(f"mtllib {stem}.mtl\n")
        cur = None
        for i in range(pos.shape[0]):
            if matid[i] != cur:
                cur = matid[i]
                f.write(f"usemtl m{int(cur)}\n")
            for k in range(3):
                f.write("v %s %s %s\n" % tuple(g(x) for x in pos[i, k]))
            if norm is not None:
                for k in range(3):
                    f.write("vn %s %s %s\n" % tuple(g(x) for x in norm[i, k]))
                f.write("f -3//-3 -2//-2 -1//-1\n")
            else:
                f.write("f -3 -2 -1\n")
    return stem + ".obj"


def run(pos, norm, matid, materials, cam_pos, cam_up, cam_dir, fov, width, height, spp=1, spt=1, render_type="distance",
        shading="flat", cull=True, max_depth=10, max_distance=2000.0, seed=42, tri_per_bb=16, ads_depth=10, ads="octree",
        timeout=600):
    """Renders one frame with the reference binary.  Returns a dict: wall_s (whole process: scene load, Build, frame,
    PPM write), ppm_md5, and the counters the reference prints."""
    with tempfile.TemporaryDirectory() as td:
        obj = write_scene(td, pos, norm, matid, materials)
        cfg = {
            "seed": seed,
            "obj_loader": {"triangulate": True, "basepath": td + "/"},
            "renderer": {
                "max_depth": max_depth, "samples_per_triangle": spt, "render_type": render_type, "max_distance": max_distance,
                "shading_type": shading, "cull_backfaces": bool(cull), "background_color": [0.01, 0.01, 0.01],
                "acceleratied_data_structure": {"name": ads, "max_triangles_per_BB": tri_per_bb, "max_depth": ads_depth},
                "output": {"filename": "output.ppm"},
            },
            "scene": {"source_file": obj, "point_lights": [],
                      "camera": {"pos": list(cam_pos), "up": list(cam_up), "dir": list(cam_dir), "fov": fov, "width": width,
                                 "height": height, "samples_per_pixel": spp}},
        }
        with open(os.path.join(td, "cfg.json"), "w") as f:
            json.dump(cfg, f)
        t0 = time.perf_counter()
        p = subprocess.run([REF_BIN, "cfg.json"], cwd=td, capture_output=True, text=True, timeout=timeout)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError(f"reference binary exited {p.returncode}: {(p.stderr or p.stdout)[-400:]}")
        with open(os.path.join(td, "output.ppm"), "rb") as f:
            md5 = hashlib.md5(f.read()).hexdigest()
    out = {"wall_s": wall, "ppm_md5": md5}
    log = (p.stderr + p.stdout).replace("\r", "\n")
    for k, pat in STAT_PATTERNS.items():
        m = re.search(pat, log)
        if m:
            out[k] = int(m.group(1))
    return out
