"""The workload constants of bench.py / the tests / the C clients are the reference's own example settings
(tests/golden/reference_settings.json, extracted from Examples/*/*.yaml by tools/gen_settings_fixture.py); when the reference
checkout is present the fixture is re-derived from its text."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_settings.json")))


def test_bench_constants_are_the_references():
    import bench
    k = FIX["KITTI00-02 stereo"]
    assert bench.BF == k["Camera.bf"] and bench.FX == k["Camera.fx"]                       # Examples/Stereo/KITTI00-02.yaml:8,25
    assert (bench.WORKLOADS["stereo2000"][0], bench.WORKLOADS["stereo2000"][1]) == (int(k["Camera.width"]), int(k["Camera.height"]))
    assert bench.WORKLOADS["stereo2000"][2] == int(k["ORBextractor.nFeatures"])             # BASELINE configs 2 / 5
    assert bench.WORKLOADS["euroc_bow"][2] == int(FIX["EuRoC mono"]["ORBextractor.nFeatures"])   # config 3
    assert bench.NLEVELS == int(k["ORBextractor.nLevels"])
    for name in FIX:    # every example uses the same pyramid / FAST settings the tests hard-code (1.2, 8, 20, 7)
        f = FIX[name]
        assert (f["ORBextractor.scaleFactor"], f["ORBextractor.nLevels"], f["ORBextractor.iniThFAST"], f["ORBextractor.minThFAST"]) == (1.2, 8, 20, 7)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "1.2, NLEVELS, 20, 7" in src


def test_c_clients_use_the_reference_constants():
    for f in ("examples/stereo_frame.c", "examples/stereo_stream.c"):
        txt = open(os.path.join(ROOT, f)).read()
        assert "386.1448f" in txt and ("718.856f" in txt or "0.5372f" in txt), f


@pytest.mark.skipif(not os.path.exists("/root/reference/Examples/Stereo/KITTI00-02.yaml"), reason="reference checkout absent")
def test_fixture_rederived_from_reference_text():
    from tools import gen_settings_fixture as g
    for name, rel in g.FILES.items():
        got = g.parse(os.path.join(g.REF, rel))
        exp = {k: v for k, v in FIX[name].items() if k != "file"}
        assert got == exp, name
