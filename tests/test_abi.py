"""The C-ABI library loads and exports every symbol include/orbx.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback exists in the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    import __graft_entry__ as ge
    ge.build()
    L = C.CDLL(pkg.lib_path())
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/orbx.h but not exported by liborbx.so"


def test_header_is_plain_c():
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "orbx.h"\nint main(void){ orbx_keypoint k; return sizeof k == 28 ? 0 : 1; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "t.o")])


def test_keypoint_layout(pkg):
    assert pkg.KP_DTYPE.itemsize == 28
    assert [pkg.KP_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]


def test_host_hamming_needs_no_gpu(pkg):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 32), dtype=np.uint8); b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    for i in range(64):
        assert pkg.ORBmatcher.DescriptorDistance(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())


def test_no_silent_cpu_fallback(pkg):
    """on a box without a GPU every compute path must raise, never compute on the host"""
    if pkg.lib().orbx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.OrbxError) as ei:
        pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code == -4
    fs = dict(desc=np.zeros((4, 32), np.uint8), node_id=np.array([1], np.uint32), node_off=np.array([0, 4], np.int32),
              feat=np.arange(4, dtype=np.uint32), flag=np.ones(4, np.uint8), angle=np.zeros(4, np.float32))
    with pytest.raises(pkg.OrbxError) as ei:
        pkg.ORBmatcher(0.75, True).SearchByBoW(fs, fs)
    assert ei.value.code == -4


def test_product_never_touches_oracle():
    """nothing under the package or include/ may reference oracle/"""
    for base in ("orb-slam2_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    assert "oracle_py" not in txt and "orb_oracle" not in txt and "liborb_oracle" not in txt, os.path.join(dp, f)


def test_invalid_arguments_rejected(pkg):
    L = pkg.lib()
    h = C.c_void_p()
    for args in [(0, 1.2, 8, 20, 7), (1000, 1.0, 8, 20, 7), (1000, 1.2, 0, 20, 7), (1000, 1.2, 99, 20, 7), (1000, 1.2, 8, 7, 20)]:
        rc = L.orbx_extractor_create(C.byref(h), args[0], args[1], args[2], args[3], args[4], 0, 640, 480, 1)
        assert rc == -1, args
        assert b"invalid" in L.orbx_last_error()


def test_no_silent_cpu_fallback_next_rows(pkg):
    """the rows either side of the path fail the same way without a GPU"""
    if pkg.lib().orbx_device_count() > 0:
        pytest.skip("a GPU is present")
    f32 = np.float32
    n = 8
    frame = dict(x=np.linspace(10, 300, n).astype(f32), y=np.linspace(10, 200, n).astype(f32), octave=np.zeros(n, np.int32),
                 angle=np.zeros(n, f32), u_right=np.full(n, -1, f32), desc=np.zeros((n, 32), np.uint8), occupied=np.zeros(n, np.uint8),
                 bounds=(0.0, 0.0, 320.0, 240.0))
    pts = dict(u=frame["x"], v=frame["y"], aux=np.ones(n, f32), level=np.zeros(n, np.int32), angle=np.zeros(n, f32),
               view_cos=np.ones(n, f32), desc=frame["desc"], valid=np.ones(n, np.uint8), has_obs=np.ones(n, np.uint8))
    sf = np.array([1.0, 1.2], f32)
    m = pkg.ORBmatcher(0.75, True)
    calls = [lambda: m.SearchByProjectionLastFrame(frame, pts, sf, 7.0),
             lambda: m.SearchByProjectionMapPoints(frame, pts, sf, 3.0),
             lambda: m.SearchByProjectionKeyFrame(frame, pts, sf, 10.0, 100),
             lambda: m.SearchByProjectionSim3(frame, pts, sf, 10.0),
             lambda: m.Fuse(frame, pts, sf, None, 3.0),
             lambda: m.SearchBySim3(frame, frame, pts, pts, sf, sf, 7.5),
             lambda: m.SearchForInitialization(frame, frame, np.stack([frame["x"], frame["y"]], 1), 100),
             lambda: pkg.UndistortKeyPoints(np.zeros((4, 2), f32), 500.0, 500.0, 320.0, 240.0, [0.1, 0.0, 0.0, 0.0]),
             lambda: pkg.Rectifier((320, 240), np.zeros((240, 320), f32), np.zeros((240, 320), f32)),
             lambda: pkg.BowFrames(2, 100),
             lambda: pkg.ComputeDistinctiveDescriptors([np.zeros((3, 32), np.uint8)])]
    for i, c in enumerate(calls):
        with pytest.raises(pkg.OrbxError) as ei:
            c()
        assert ei.value.code == -4, i


def _build_c_example(tmpdir):
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(tmpdir, "stereo_frame")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "stereo_frame.c"),
                           "-L", os.path.join(ROOT, "orb-slam2_amd"), "-lorbx", "-Wl,-rpath," + os.path.join(ROOT, "orb-slam2_amd"), "-o", exe])
    return exe


def test_c_example_compiles_and_links(tmp_path):
    """a plain C client links against liborbx.so through include/orbx.h alone (no torch, no Python)"""
    _build_c_example(str(tmp_path))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    import subprocess
    exe = _build_c_example(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "stereo matches" in out.stdout
