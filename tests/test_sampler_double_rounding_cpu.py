"""CPU: tests/golden/sampler_double_rounding.json -- inputs on which fusing the vertical blend of SampleLinear
(reference src/sample_eigen.h:82-83) into an FMA changes the float result.  Pins the ORACLE to the reference's
FMA-free arithmetic (CMakeLists.txt:25: -msse4.1, no -mfma) and checks that every case really separates the forms."""
import json
import os
import struct

import numpy as np

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cases():
    with open(os.path.join(ROOT, "tests", "golden", "sampler_double_rounding.json")) as f:
        return json.load(f)["cases"]


def texel_values(case):
    if case["mode"] == 0:
        return [float(v) for v in case["texels"]]
    return [struct.unpack("<f", struct.pack("<I", v))[0] for v in case["texels"]]


def bits32(v):
    return struct.unpack("<I", struct.pack("<f", float(np.float32(v))))[0]


def position(case):
    return np.float32(2.0 - case["ky"] * 2.0 ** -23), np.float32(2.0 - case["kx"] * 2.0 ** -23)


def plane(case):
    p = np.zeros((8, 8), np.float32)
    p[1, 1], p[1, 2], p[2, 1], p[2, 2] = texel_values(case)
    return p


def test_fixture_is_meaningful():
    cases = load_cases()
    assert sum(c["mode"] == 0 for c in cases) >= 5 and sum(c["mode"] == 1 for c in cases) >= 20
    for c in cases:
        assert c["expected"] != c["fused_dy_top"] or c["expected"] != c["fused_omdy_bot"]
        y, x = position(c)
        assert 1.0 < y < 2.0 and 1.0 < x < 2.0
        assert np.float32(2.0) - y == np.float32(c["ky"] * 2.0 ** -23) and np.float32(2.0) - x == np.float32(c["kx"] * 2.0 ** -23)
    assert any(c["expected"] != c["fused_dy_top"] for c in cases) and any(c["expected"] != c["fused_omdy_bot"] for c in cases)


def test_oracle_sampler_is_the_unfused_form():
    z = np.zeros((8, 8), np.float32)
    for c in load_cases():
        y, x = position(c)
        got = oracle.sample_linear(np.stack([plane(c), z, z]), y, x)
        assert bits32(got[0]) == c["expected"], c
