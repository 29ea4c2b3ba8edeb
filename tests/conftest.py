import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Camera set-ups of the two shipped Cornell scenes (res/CornellBox-Original.txt:2-6 and
# default_config.json:11,30-46 of the reference).
CAMERAS = {
    "original": dict(cam_pos=[278.0, 273.0, -1000.0], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=2000.0),
    "sphere": dict(cam_pos=[0.0, 1.0, 4.42], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, -1.0], max_distance=6.0),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def orc():
    import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def scenes(orc):
    return {k: orc.Scene.from_npz(os.path.join(GOLDEN, "scenes", f"cornell_{k}.npz")) for k in CAMERAS}


def soup_that_breaks_the_build_invariant():
    """54,688 random triangles on which Octree::Build's own check fails (octree.cc:124-152: a triangle of a subdivided
    node passes the triangle/box test of the node but of none of its octants; the reference prints and exits there)."""
    import numpy as np
    rs = np.random.RandomState(6085)
    rs.randint(20, 400), rs.randint(300, 2500), rs.choice([4, 8, 16]), rs.randint(3, 6)  # (the draws tools/light_cull_fuzz.py makes before this one)
    n_tri = int(rs.randint(20000, 60000))
    centre = rs.rand(n_tri, 1, 3) * 10.0 - 5.0
    size = np.exp(rs.uniform(np.log(0.02), np.log(0.4), (n_tri, 1, 1)))
    pos = (centre + rs.randn(n_tri, 3, 3) * size).astype(np.float32)
    pos[: n_tri // 3, :, 1] = np.round(pos[: n_tri // 3, :, 1])
    return pos
