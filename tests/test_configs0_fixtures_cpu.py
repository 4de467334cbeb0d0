"""CPU: the configs[0] data fixtures (tests/golden/configs0/) are the reference's own files -- checked byte for byte
whenever the reference tree is present (this container); on the GPU box only their content is checked."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "configs0")
REF = "/root/reference"


def test_fixture_content():
    import re
    text = open(os.path.join(FIX, "config", "kitti_stereo.cfg")).read()
    keys = dict(re.findall(r"^(\w+)\s*=\s*(\S+)\s*$", text, flags=re.M))
    assert keys["slidingWindowSize"] == "5" and keys["patchRadius"] == "1" and keys["maxNumPoints"] == "4096"
    assert keys["minScore"] == "0.65" and keys["robustThreshold"] == "0.05"
    assert keys["Trajectory"] == "../data/kitti_init_poor/00.txt"
    poses = np.loadtxt(os.path.join(FIX, "data", "kitti_init_poor", "00.txt")).reshape(-1, 3, 4)
    assert poses.shape[0] == 12
    assert np.array_equal(poses[0], np.eye(4)[:3])
    # frame-to-frame poses of a forward-driving car: ~0.7 m per frame along -z (inverse of the motion), tiny rotations
    assert np.all(poses[1:, 2, 3] < -0.6) and np.all(poses[1:, 2, 3] > -0.9)
    assert np.abs(poses[:, :, :3] - np.eye(3)).max() < 0.02


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_fixtures_are_the_reference_files():
    assert open(os.path.join(FIX, "config", "kitti_stereo.cfg"), "rb").read() == open(os.path.join(REF, "config", "kitti_stereo.cfg"), "rb").read()
    ref_lines = open(os.path.join(REF, "data", "kitti_init_poor", "00.txt")).read().split("\n")[:12]
    assert open(os.path.join(FIX, "data", "kitti_init_poor", "00.txt")).read() == "\n".join(ref_lines) + "\n"
