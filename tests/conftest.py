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
