#!/usr/bin/env python3
"""Extracts the ORB / camera settings of the reference's example configurations (data, from the text of
Examples/*/*.yaml) into tests/golden/reference_settings.json: the numbers bench.py, the tests and the C examples use for
BASELINE configs 1-5 (nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST, image size, fx, bf).  Build container only."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Examples"
FILES = {"KITTI00-02 stereo": "Stereo/KITTI00-02.yaml", "TUM1 mono": "Monocular/TUM1.yaml", "EuRoC mono": "Monocular/EuRoC.yaml",
         "EuRoC stereo": "Stereo/EuRoC.yaml"}
KEYS = ("Camera.fx", "Camera.bf", "Camera.width", "Camera.height", "Camera.fps", "ThDepth", "ORBextractor.nFeatures", "ORBextractor.scaleFactor",
        "ORBextractor.nLevels", "ORBextractor.iniThFAST", "ORBextractor.minThFAST")


def parse(path):
    txt = open(path, encoding="utf-8", errors="replace").read()
    out = {}
    for k in KEYS:
        m = re.search(r"^%s\s*:\s*([-+0-9.eE]+)" % re.escape(k), txt, re.M)
        if m:
            out[k] = float(m.group(1))
    return out


if __name__ == "__main__":
    doc = {name: dict(parse(os.path.join(REF, rel)), file="Examples/" + rel) for name, rel in FILES.items()}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "reference_settings.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1))
