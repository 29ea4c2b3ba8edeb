#!/usr/bin/env python3
"""Run the unmodified reference binary (oracle/_ref/raytracer_ref, built by
oracle/Makefile from the sources under /root/reference) on a set of known-answer
configurations and record, for each, the md5 of the PPM it writes and the
counters it prints.  Output: tests/golden/ref_kats.json (data only).

This is test infrastructure; it runs only where the reference checkout exists.
"""
import hashlib, json, os, re, subprocess, sys, tempfile, time

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "raytracer_ref")
REF_RES = os.environ.get("RT_REF_RES", "/root/reference/res/")

CAM_ORIGINAL = dict(pos=[278.0, 273.0, -1000.0], up=[0.0, 1.0, 0.0], dir=[0.0, 0.0, 1.0], fov=0.6)
CAM_SPHERE = dict(pos=[0.0, 1.0, 4.42], up=[0.0, 1.0, 0.0], dir=[0.0, 0.0, -1.0], fov=0.6)


def make_config(scene, render_type, shading, ads, w, h, spt=1, spp=1, cull=True, seed=42,
                max_depth=10, tri_per_bb=16, ads_depth=10):
    if scene == "original":
        src, cam, maxd = "CornellBox-Original.obj", CAM_ORIGINAL, 2000.0
    else:
        src, cam, maxd = "CornellBox-Sphere.obj", CAM_SPHERE, 6.0
    camera = dict(cam)
    camera.update(width=w, height=h, samples_per_pixel=spp)
    return {
        "seed": seed,
        "obj_loader": {"triangulate": True, "basepath": REF_RES},
        "renderer": {
            "max_depth": max_depth,
            "samples_per_triangle": spt,
            "render_type": render_type,
            "max_distance": maxd,
            "shading_type": shading,
            "cull_backfaces": cull,
            "background_color": [0.01, 0.01, 0.01],
            "acceleratied_data_structure": {"name": ads, "max_triangles_per_BB": tri_per_bb, "max_depth": ads_depth},
            "output": {"filename": "output.ppm"},
        },
        "scene": {"source_file": src, "camera": camera, "point_lights": []},
    }


KATS = {
    "K1": dict(scene="original", render_type="distance", shading="flat", ads="octree", w=640, h=480),
    "K2": dict(scene="original", render_type="distance", shading="flat", ads="octree", w=128, h=128),
    "K2n": dict(scene="original", render_type="distance", shading="flat", ads="none", w=128, h=128),
    "K3": dict(scene="sphere", render_type="diffusion", shading="smooth", ads="octree", w=128, h=128),
    "K4": dict(scene="sphere", render_type="distance", shading="flat", ads="octree", w=128, h=128, cull=False),
    "K5": dict(scene="original", render_type="blinn_phong", shading="flat", ads="octree", w=96, h=96, spt=4),
    "K6": dict(scene="sphere", render_type="phong", shading="smooth", ads="octree", w=64, h=64, spt=2),
    "K7": dict(scene="sphere", render_type="blinn_phong", shading="smooth", ads="octree", w=64, h=64, spt=2, spp=4),
    "K8": dict(scene="original", render_type="blinn_phong", shading="flat", ads="octree", w=256, h=256, spt=16),
    "K9": dict(scene="sphere", render_type="blinn_phong", shading="smooth", ads="octree", w=128, h=128, spt=4),
    "K10": dict(scene="original", render_type="phong", shading="flat", ads="octree", w=64, h=96, spt=2),
    "K11": dict(scene="sphere", render_type="distance", shading="flat", ads="octree", w=96, h=96,
                tri_per_bb=4, ads_depth=3),
}

STAT_PATTERNS = {
    "max_depth": r"Max depth: (\d+)",
    "nodes": r"Nodes count: (\d+)",
    "leaves": r"Leaf nodes count: (\d+)",
    "avg_leaf_depth": r"Average depth of leaf nodes: ([\d.eE+-]+)",
    "max_tris_in_leaf": r"Max triangles in leaf nodes: (\d+)",
    "avg_tris_in_leaf": r"Average triangles in leaf nodes: ([\d.eE+-]+)",
    "search_calls": r"Search method call count: (\d+)",
    "search_nodes": r"Search node count: (\d+)",
    "search_returned": r"Search return count: (\d+)",
    "search_leaves": r"Search leaves visited: (\d+)",
    "tri_tests": r"Ray-triangle collision count: (\d+)",
}


def run_one(name, spec):
    cfg = make_config(**spec)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "cfg.json"), "w") as f:
            json.dump(cfg, f)
        t0 = time.time()
        p = subprocess.run([REF_BIN, "cfg.json"], cwd=td, capture_output=True, text=True)
        dt = time.time() - t0
        if p.returncode != 0:
            raise RuntimeError(f"{name}: reference exited {p.returncode}: {p.stderr[-400:]}")
        with open(os.path.join(td, "output.ppm"), "rb") as f:
            ppm = f.read()
    log = p.stderr.replace("\r", "\n")
    out = {"spec": spec, "ppm_md5": hashlib.md5(ppm).hexdigest(), "wall_s": round(dt, 3)}
    for k, pat in STAT_PATTERNS.items():
        m = re.search(pat, log)
        if m:
            out[k] = float(m.group(1)) if "." in m.group(1) or "e" in m.group(1) else int(m.group(1))
    return out


def main():
    names = sys.argv[1:] or list(KATS)
    path = os.path.join(HERE, "..", "tests", "golden", "ref_kats.json")
    res = {}
    if os.path.exists(path):
        with open(path) as f:
            res = json.load(f)
    for n in names:
        res[n] = run_one(n, KATS[n])
        print(n, json.dumps(res[n]))
    with open(path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
