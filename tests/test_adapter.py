"""The C++ drop-in adaptor (adapter/*.cc: ORB_SLAM2::ORBextractor, Frame::ComputeStereoMatches, the three BoW-guided
ORBmatcher searches over the C ABI) is real code that a compiler sees:

* CPU: every adaptor source passes `g++ -std=c++11 -Wall -Wextra -Werror -fsyntax-only` against tests/cvstub (OpenCV is not
  installed here; the stub holds only the members the adaptor touches) and tests/adapter_driver.cc links with liborbx.so and
  round-trips a DBoW2::FeatureVector through flatten() into an orbx_featset -- also against the reference's REAL
  FeatureVector.{h,cpp} when the reference checkout is present.
* GPU: the driver runs a stereo frame through the compiled adaptor classes exactly as Tracking / LocalMapping /
  LoopClosing call them; every output is compared with the CPU oracle bit for bit.
Reference interfaces: include/ORBextractor.h:58-139, include/ORBmatcher.h:41-103, src/Frame.cc:577-751."""
import os
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = [os.path.join(ROOT, "adapter", f) for f in ("ORBextractor.cc", "Frame_stereo.cc", "ORBmatcher_bow.cc", "ORBmatcher_proj.cc", "ORBmatcher_fuse.cc",
                                                       "Frame_bow.cc", "MapPoint_distinctive.cc", "ORBmatcher_batch.cc")]
INC = ["-I", os.path.join(ROOT, "adapter"), "-I", os.path.join(ROOT, "tests", "cvstub"), "-I", os.path.join(ROOT, "include")]
REF = "/root/reference"


def _build_driver(tmpdir, real_dbow2=False):
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(tmpdir, "adapter_driver" + ("_ref" if real_dbow2 else ""))
    inc, extra = list(INC), []
    if real_dbow2:   # "Thirdparty/DBoW2/DBoW2/FeatureVector.h" now resolves to the reference's own header
        inc = ["-I", REF] + inc
        extra = [os.path.join(REF, "Thirdparty/DBoW2/DBoW2", f) for f in ("FeatureVector.cpp", "BowVector.cpp")]
    # ORBX_ADAPTER_CAPTURE: the projection adaptors keep a copy of what they hand to the ABI (the `track` mode writes it out)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DORBX_ADAPTER_CAPTURE"] + inc + [os.path.join(ROOT, "tests", "adapter_driver.cc")] + ADAPTER + extra +
                          ["-L", os.path.join(ROOT, "orb-slam2_amd"), "-lorbx", "-Wl,-rpath," + os.path.join(ROOT, "orb-slam2_amd"), "-o", exe])
    return exe


@pytest.mark.parametrize("src", ADAPTER)
def test_adapter_sources_compile(src):
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + INC + [src])


def test_cv_keypoint_stub_is_28_bytes(tmp_path):
    src = os.path.join(tmp_path, "t.cc")
    open(src, "w").write('#include <opencv2/core/core.hpp>\n#include <orbx.h>\nstatic_assert(sizeof(cv::KeyPoint) == 28 && sizeof(orbx_keypoint) == 28, "");\nint main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only"] + INC + [src])


def _decls(text, names):
    """{method name: [normalised parameter lists]} of the member declarations `int name(...);` in a header"""
    import re
    out = {}
    for m in re.finditer(r"\bint\s+(\w+)\s*\(([^;{]*)\)\s*;", text):
        if m.group(1) in names:
            params = []
            for prm in m.group(2).split(","):
                prm = re.sub(r"=[^,]*$", "", prm)                      # default value
                prm = re.sub(r"\bstd::", "", prm)
                prm = re.sub(r"\s*([*&<>])\s*", r"\1", " ".join(prm.split()))
                prm = re.sub(r"[*&]?\s*\w+$", lambda t: t.group(0)[0] if t.group(0)[0] in "*&" else "", prm).strip()   # drop the parameter name
                params.append(prm)
            out.setdefault(m.group(1), []).append(tuple(params))
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include/ORBmatcher.h")), reason="reference checkout absent")
def test_signatures_match_the_reference_header():
    """every ORBmatcher method an adaptor defines is declared in tests/cvstub/ORBmatcher.h with the parameter types the reference's
    include/ORBmatcher.h (:41-83) gives it: the stand-in header cannot drift from the interface the drop-in has to meet"""
    names = {"SearchByProjection", "SearchByBoW", "SearchForInitialization", "SearchForTriangulation", "SearchBySim3", "Fuse"}
    ref = _decls(open(os.path.join(REF, "include/ORBmatcher.h"), errors="replace").read(), names)
    stub = _decls(open(os.path.join(ROOT, "tests/cvstub/ORBmatcher.h")).read(), names)
    assert stub, "no declarations parsed"
    for name, sigs in stub.items():
        for sig in sigs:
            assert sig in ref.get(name, []), f"{name}{sig} is not a signature of the reference header: {ref.get(name)}"


CSR_EXPECT = ["id 0 3 7 900000", "off 0 1 4 7 8", "feat 16 11 14 15 10 12 17 13"]


def _check_csr(exe, have_gpu):
    out = subprocess.run([exe, "csr"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[:3] == CSR_EXPECT
    assert lines[3] == ("rc 0" if have_gpu else "rc -4")      # through the ABI: ORBX_OK on a GPU box, ORBX_E_NO_DEVICE here; no host fallback


def test_feature_vector_roundtrip(pkg, tmp_path):
    _check_csr(_build_driver(str(tmp_path)), pkg.lib().orbx_device_count() > 0)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Thirdparty/DBoW2/DBoW2/FeatureVector.cpp")), reason="reference checkout absent")
def test_feature_vector_roundtrip_with_reference_dbow2(pkg, tmp_path):
    _check_csr(_build_driver(str(tmp_path), real_dbow2=True), pkg.lib().orbx_device_count() > 0)


# ------------------------------------------------------------------------------------------------ GPU

def _parse(path):
    out = {}
    for line in open(path):
        t = line.split()
        out[t[0]] = np.array([int(v) for v in t[2:]], np.int64)
        assert len(out[t[0]]) == int(t[1])
    return out


def _f32(a):
    return a.astype(np.uint32).view(np.float32)


def _keys(a, pkg):
    a = a.reshape(-1, 7)
    k = np.zeros(len(a), pkg.KP_DTYPE)
    for j, f in enumerate(("x", "y", "size", "angle", "response")):
        k[f] = _f32(a[:, j])
    k["octave"] = a[:, 5]; k["class_id"] = a[:, 6]
    return k


def _featset(desc, kp, flag, u_right, nodes=16):
    keep = desc[:, 1] % 8 != 0
    node = 100 + ((desc[:, 0] & 15) if nodes == 16 else desc[:, 0] % nodes).astype(np.uint32)
    ids = np.unique(node[keep])
    feat = np.concatenate([np.nonzero(keep & (node == i))[0] for i in ids]).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum([int((keep & (node == i)).sum()) for i in ids])]).astype(np.int32)
    return dict(desc=desc, node_id=ids.astype(np.uint32), node_off=off, feat=feat, flag=flag.astype(np.uint8), angle=kp["angle"].copy(),
                x=kp["x"].copy(), y=kp["y"].copy(), octave=kp["octave"].copy(), u_right=u_right.astype(np.float32))


@pytest.mark.gpu
def test_adapter_runs_and_matches_oracle(pkg, oracle, tmp_path):
    w, h = 1241, 376
    left, right, _ = synth.stereo_pair(515, w, h)
    inp, outp = os.path.join(tmp_path, "in.bin"), os.path.join(tmp_path, "out.txt")
    with open(inp, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes()); f.write(left.tobytes()); f.write(right.tobytes())
    exe = _build_driver(str(tmp_path))
    run = subprocess.run([exe, "run", inp, outp], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    r = _parse(outp)
    bf, fx = np.float32(386.1448), np.float32(718.856)
    oL, oR = oracle.Oracle(1000, 1.2, 8, 20, 7), oracle.Oracle(1000, 1.2, 8, 20, 7)
    kL, dL = oL.extract(left); kR, dR = oR.extract(right)
    # ORBextractor::operator() x 2, getters, mvImagePyramid
    assert _keys(r["keysL"], pkg).tobytes() == kL.tobytes() and _keys(r["keysR"], pkg).tobytes() == kR.tobytes()
    assert (r["descL"].reshape(-1, 32) == dL).all() and (r["descR"].reshape(-1, 32) == dR).all()
    assert _f32(r["scaleFactors"]).tobytes() == np.asarray(oL.scale_factors(), np.float32).tobytes()
    assert _f32(r["levelSigma2"]).tobytes() == np.asarray(oL.level_sigma2(), np.float32).tobytes()
    assert _f32(r["invScaleFactors"]).tobytes() == np.asarray(oL.inv_scale_factors(), np.float32).tobytes()
    assert _f32(r["invLevelSigma2"]).tobytes() == np.asarray(oL.inv_level_sigma2(), np.float32).tobytes()
    lv3 = oL.level(3)
    assert r["pyramidDims"].reshape(-1, 2)[3].tolist() == [lv3.shape[1], lv3.shape[0]] and (r["pyramid3"].reshape(lv3.shape) == lv3).all()
    # Frame::ComputeStereoMatches with mb = mbf / fx
    our, odp = oracle.stereo_match(oL, oR, kL, dL, kR, dR, float(bf), float(bf / fx))
    assert _f32(r["uRight"]).tobytes() == our.tobytes() and _f32(r["depth"]).tobytes() == odp.tobytes() and (our >= 0).sum() > 50
    assert _keys(r["oneCallKeysL"], pkg).tobytes() == kL.tobytes() and (r["oneCallDescR"].reshape(-1, 32) == dR).all()
    assert _f32(r["oneCallURight"]).tobytes() == our.tobytes()
    # ORBmatcher::SearchByBoW x 2 and SearchForTriangulation with the driver's MapPoint pattern
    iR, iL = np.arange(len(kR)), np.arange(len(kL))
    has_r, bad_r = iR % 3 != 2, iR % 5 == 1
    has_l, bad_l = iL % 3 != 2, iL % 6 == 1
    none_r = np.full(len(kR), -1.0, np.float32)
    f_frame = _featset(dL, kL, np.zeros(len(kL)), our)
    kf_r = _featset(dR, kR, has_r & ~bad_r, none_r)
    kf_l = _featset(dL, kL, has_l & ~bad_l, our)
    exp1, n1 = oracle.search_by_bow_kf_f(kf_r, f_frame, 0.75, True)
    assert int(r["bowKfF_n"][0]) == n1 and (r["bowKfF"] == exp1).all() and n1 > 20
    exp2, n2 = oracle.search_by_bow_kf_kf(kf_r, kf_l, 0.75, True)
    assert int(r["bowKfKf_n"][0]) == n2 and (r["bowKfKf"] == exp2).all() and n2 > 10
    t_r = _featset(dR, kR, has_r, none_r); t_l = _featset(dL, kL, has_l, our)
    c2 = np.array([0.5372, 0.0, 0.02], np.float32)
    invz = np.float32(1.0) / c2[2]
    ex = np.float32(np.float32(fx * c2[0]) * invz) + np.float32(607.1928)
    ey = np.float32(np.float32(fx * c2[1]) * invz) + np.float32(185.2157)
    F12 = np.array([0, 0, 0, 0, 0, np.float32(-1.0) / fx, 0, np.float32(1.0) / fx, 0], np.float32)
    exp3 = oracle.search_for_triangulation(t_r, t_l, F12, float(ex), float(ey), np.asarray(oL.scale_factors(), np.float32),
                                           np.asarray(oL.level_sigma2(), np.float32), 0.6, False, False)
    assert int(r["tri_n"][0]) == len(exp3) and (r["triPairs"].reshape(-1, 2) == exp3.reshape(-1, 2)).all() and len(exp3) > 5


def _u8(a):
    return a.astype(np.uint8)


def _cur(r, pre):
    return dict(x=_f32(r[pre + "cur_x"]), y=_f32(r[pre + "cur_y"]), octave=r[pre + "cur_octave"].astype(np.int32), angle=_f32(r[pre + "cur_angle"]),
                u_right=_f32(r[pre + "cur_uright"]), desc=_u8(r[pre + "cur_desc"]).reshape(-1, 32), occupied=_u8(r[pre + "cur_occupied"]),
                bounds=[float(v) for v in _f32(r[pre + "cur_bounds"])])


def _pts(r, pre):
    return dict(u=_f32(r[pre + "pts_u"]), v=_f32(r[pre + "pts_v"]), aux=_f32(r[pre + "pts_aux"]), level=r[pre + "pts_level"].astype(np.int32),
                angle=_f32(r[pre + "pts_angle"]), view_cos=_f32(r[pre + "pts_view"]), desc=_u8(r[pre + "pts_desc"]).reshape(-1, 32),
                valid=_u8(r[pre + "pts_valid"]), has_obs=_u8(r[pre + "pts_has_obs"]))


@pytest.mark.gpu
def test_adapter_tracking_searches(pkg, oracle, tmp_path):
    """Tracking's per-frame searches, Frame::ComputeBoW / UndistortKeyPoints and MapPoint::ComputeDistinctiveDescriptors through the
    COMPILED adaptors (adapter/ORBmatcher_proj.cc, Frame_bow.cc, MapPoint_distinctive.cc), against the CPU oracle fed with exactly
    the inputs each adaptor handed to the ABI"""
    w, h = 1241, 376
    left, right, _ = synth.stereo_pair(517, w, h)
    inp, outp, vocp = os.path.join(tmp_path, "in.bin"), os.path.join(tmp_path, "out.txt"), os.path.join(tmp_path, "voc.txt")
    with open(inp, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes()); f.write(left.tobytes()); f.write(right.tobytes())
    oL, oR = oracle.Oracle(1000, 1.2, 8, 20, 7), oracle.Oracle(1000, 1.2, 8, 20, 7)
    kL, dL = oL.extract(left); kR, dR = oR.extract(right)
    par, leaf, nd, wt = synth.vocab_tree(77, 10, 3, stop_frac=0.02, data=dL)
    synth.write_vocab_text(vocp, 10, 3, par, leaf, nd, wt)
    exe = _build_driver(str(tmp_path))
    run = subprocess.run([exe, "track", inp, vocp, outp], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    r = _parse(outp)
    assert _keys(r["keysCur"], pkg).tobytes() == kL.tobytes() and _keys(r["keysLast"], pkg).tobytes() == kR.tobytes()
    sf = np.asarray(oL.scale_factors(), np.float32)
    nL, nR = len(kL), len(kR)
    # ---- SearchByProjection(CurrentFrame, LastFrame): both pose variants (plain window / bForward)
    for pre, direction in (("lastA_", 0), ("lastB_", 1)):
        cur, pts = _cur(r, pre), _pts(r, pre)
        assert len(cur["x"]) == nL and len(pts["u"]) == nR and cur["desc"].tobytes() == dL.tobytes()
        iR = np.arange(nR)
        assert (pts["valid"] == ((iR % 5 != 3) & (iR % 9 != 4))).all()                      # has a point and is no outlier
        ok = pts["valid"] == 1
        assert (pts["desc"][ok] == dR[ok]).all() and (pts["level"][ok] == kR["octave"][ok]).all()
        # the prologue's projection: each point was placed to project 12 px right of its right-eye position (float32 round trips)
        assert np.abs(pts["u"][ok] - (kR["x"][ok] + 12)).max() < 0.05 and np.abs(pts["v"][ok] - kR["y"][ok]).max() < 0.05
        assert np.abs(1.0 / pts["aux"][ok] - (8 + iR[ok] % 7)).max() < 1e-3
        exp, en = oracle.search_by_projection_last(cur, pts, sf, 15.0, direction, 386.1448, 3)   # 3: orientation check + cleared features marked -2
        held = r[pre + "held"]
        before = np.where(np.arange(nL) % 11 == 0, -2, -1)        # (-2 in `held`: the feature still holds the point it held before the search)
        # a feature the rotation filter cleared ends NULL as in the reference (:1526-1545), also when it held a point before
        assert int(r[pre + "n"][0]) == en and (held == np.where(exp >= 0, exp, np.where(exp == -2, -1, before))).all(), pre
        assert en > 30
    # ---- SearchByProjection(F, vpMapPoints)
    cur, pts = _cur(r, "local_"), _pts(r, "local_")
    iR = np.arange(nR)
    assert (pts["valid"] == ((iR % 7 != 3) & (iR % 13 != 5))).all()
    exp, en = oracle.search_by_projection_points(cur, pts, sf, 3.0, 0.8)
    before = np.where(np.arange(nL) % 11 == 0, -2, -1)
    assert int(r["local_n"][0]) == en and (r["local_held"] == np.where(exp >= 0, exp, before)).all() and en > 30
    # ---- SearchForInitialization(F1 = current, F2 = last)
    f2 = _cur(r, "init_")
    f1 = dict(x=kL["x"], y=kL["y"], octave=kL["octave"], angle=kL["angle"], u_right=np.where(np.arange(nL) % 4 == 0, kL["x"] - 7.5, -1).astype(np.float32),
              desc=dL, occupied=np.zeros(nL, np.uint8), bounds=f2["bounds"])
    prev = np.stack([kL["x"], kL["y"]], axis=1)
    exp, en = oracle.search_for_initialization(f1, f2, prev, 100, 0.9, True)
    assert int(r["init_n"][0]) == en and (r["init_m12"] == exp).all() and en > 20
    after = _f32(r["init_prev_after"]).reshape(-1, 2)
    want = prev.copy(); m = exp >= 0
    want[m, 0] = kR["x"][exp[m]]; want[m, 1] = kR["y"][exp[m]]
    assert after.tobytes() == want.astype(np.float32).tobytes()
    # ---- Frame::ComputeBoW
    ovoc = oracle.Vocabulary(path=vocp)
    t = ovoc.transform(dL, 4)
    assert (r["bow_id"] == t["bow_id"]).all() and r["bow_val"].astype(np.uint64).view(np.float64).tobytes() == t["bow_val"].tobytes()
    assert (r["fv_id"] == t["fv_node_id"]).all() and (r["fv_off"] == t["fv_node_off"]).all() and (r["fv_feat"] == t["fv_feat"]).all()
    # ---- Frame::UndistortKeyPoints
    xy = np.stack([kL["x"], kL["y"]], axis=1)
    exp_xy = oracle.undistort_points(xy, 517.3, 516.5, 318.6, 255.3, np.array([0.2624, -0.9531, -0.0054, 0.0026, 1.1633], np.float32))
    assert _f32(r["undist_xy"]).tobytes() == exp_xy.astype(np.float32).tobytes()
    # ---- MapPoint::ComputeDistinctiveDescriptors (single and batched form)
    kf_desc = [dL, dR, _u8(r["kf2_desc"]).reshape(-1, 32)]
    obs = r["distinct_obs"].reshape(-1, 7); got = r["distinct_desc"].reshape(-1, 33)
    for p in range(len(obs)):
        rows = [kf_desc[k][row] for k, row in obs[p, 1:].reshape(3, 2) if k >= 0 and k != 1]    # keyframe 1 is bad: its observation is left out
        if obs[p, 0] or not rows:
            assert got[p, 0] == 0, p                                                            # a bad point / no usable observation: untouched
            continue
        best = oracle.distinctive_descriptor(np.array(rows, np.uint8))
        assert got[p, 0] == 1 and (got[p, 1:] == rows[best]).all(), p


@pytest.mark.gpu
def test_adapter_map_searches(pkg, oracle, tmp_path):
    """relocalisation, local-mapping and loop-closing searches through the COMPILED adapter/ORBmatcher_fuse.cc: SearchByProjection(Frame, KeyFrame, ...),
    SearchByProjection(KeyFrame, Scw, ...), both Fuse overloads (with the map surgery the adaptor applies in the reference's order, replayed here
    from the oracle's answer) and SearchBySim3 -- each against the CPU oracle fed with exactly what the adaptor handed to the ABI"""
    w, h = 1241, 376
    left, right, _ = synth.stereo_pair(519, w, h)
    inp, outp = os.path.join(tmp_path, "in.bin"), os.path.join(tmp_path, "out.txt")
    with open(inp, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes()); f.write(left.tobytes()); f.write(right.tobytes())
    exe = _build_driver(str(tmp_path))
    run = subprocess.run([exe, "map", inp, outp], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    r = _parse(outp)
    oL, oR = oracle.Oracle(1000, 1.2, 8, 20, 7), oracle.Oracle(1000, 1.2, 8, 20, 7)
    kL, dL = oL.extract(left); kR, dR = oR.extract(right)
    assert _keys(r["keysL"], pkg).tobytes() == kL.tobytes() and _keys(r["keysR"], pkg).tobytes() == kR.tobytes()
    nL, nR = len(kL), len(kR)
    sf = np.asarray(oL.scale_factors(), np.float32); inv2 = np.asarray(oL.inv_level_sigma2(), np.float32)
    iL, iR = np.arange(nL), np.arange(nR)
    has_L, bad_L = iL % 3 != 2, iL % 6 == 1          # tests/adapter_driver.cc: fill_keyframe(kfL, ..., bad_every 6, none_every 3)
    has_R, bad_R = iR % 3 != 2, iR % 5 == 1
    # ---- 1. relocalisation
    cur, pts = _cur(r, "reloc_"), _pts(r, "reloc_")
    assert cur["desc"].tobytes() == dL.tobytes() and (cur["occupied"] == (iL % 13 == 0)).all()
    assert not (pts["valid"] & ~(has_R & ~bad_R & (iR % 19 != 0))).any() and pts["valid"].sum() > 300
    ok = pts["valid"] == 1
    assert np.abs(pts["u"][ok] - (kR["x"][ok] + 9)).max() < 0.05 and np.abs(pts["v"][ok] - kR["y"][ok]).max() < 0.05   # the projection prologue
    assert (np.abs(pts["level"][ok] - kR["octave"][ok]) <= 1).all()                                                       # PredictScale
    exp, en = oracle.search_by_projection_keyframe(cur, pts, sf, 10.0, 100, 3)
    before = np.where(iL % 13 == 0, -2, -1)
    assert int(r["reloc_n"][0]) == en and (r["reloc_held"] == np.where(exp >= 0, exp, np.where(exp == -2, -1, before))).all() and en > 30
    # ---- 2. SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)
    kf, pts = _cur(r, "sim3p_"), _pts(r, "sim3p_")
    occ = (iL % 9 == 0); occ[1] = True
    assert (kf["occupied"] == occ).all()
    assert not (pts["valid"] & ~(~bad_R & (iR != 40) & (iR % 10 != 7) & (iR % 17 != 3))).any() and pts["valid"].sum() > 300
    exp, en = oracle.search_by_projection_sim3(kf, pts, sf, 10.0)
    before = np.where(iL % 9 == 0, -2, -1); before[1] = 40
    assert int(r["sim3p_n"][0]) == en and (r["sim3p_held"] == np.where(exp >= 0, exp, before)).all() and en > 30
    # ---- 3. Fuse(pKF, vpMapPoints, th): replay the adaptor's map surgery (src/ORBmatcher.cc:1011-1033) from the oracle's best keypoints
    kf, pts = _cur(r, "fuse_"), _pts(r, "fuse_")
    best, _bd, _nf = oracle.window_best(kf, pts, sf, inv2, 3.0, 1, 50)
    in_kf = np.where((iR >= 5) & ((iR - 5) % 29 == 0), 0, -1)
    assert not (pts["valid"] & ~((iR % 23 != 0) & ~bad_R & (in_kf < 0))).any()
    badR, badL = bad_R.copy(), bad_L.copy()
    obsR, obsL = 1 + iR % 4, 1 + iL % 5
    repR, repL = np.full(nR, -1), np.full(nL, -1)
    kfpt = [("L", j) if has_L[j] else None for j in range(nL)]
    nfused = 0
    for i in range(nR):
        if best[i] < 0 or i % 23 == 0 or badR[i] or in_kf[i] >= 0:
            continue
        holder = kfpt[best[i]]
        if holder is not None:
            kind, j = holder
            hbad = badL[j] if kind == "L" else badR[j]
            if not hbad:
                hobs = obsL[j] if kind == "L" else obsR[j]
                if hobs > obsR[i]:
                    badR[i] = True; repR[i] = -2 if kind == "L" else j
                elif kind == "L":
                    badL[j] = True; repL[j] = -2
                else:
                    badR[j] = True; repR[j] = i
        else:
            in_kf[i] = best[i]; obsR[i] += 1; kfpt[best[i]] = ("R", i)
        nfused += 1
    assert int(r["fuse_n"][0]) == nfused and nfused > 30
    gR, gL = r["fuse_ptsR"].reshape(-1, 4), r["fuse_ptsL"].reshape(-1, 4)
    assert (gR[:, 0] == badR).all() and (gR[:, 1] == repR).all() and (gR[:, 2] == in_kf).all() and (gR[:, 3] == obsR).all()
    assert (gL[:, 0] == badL).all() and (gL[:, 1] == repL).all()
    assert (r["fuse_kf_points"] == np.array([-1 if p_ is None else (-2 if p_[0] == "L" else p_[1]) for p_ in kfpt])).all()
    # ---- 4. Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    kf, pts = _cur(r, "fuse2_"), _pts(r, "fuse2_")
    best, _bd, _nf = oracle.window_best(kf, pts, sf, None, 4.0, 0, 50)
    in_kf = np.full(nR, -1); obsR = 1 + iR % 4
    kfpt = [("L", j) if has_L[j] else None for j in range(nL)]
    repl = np.full(nR, -1)
    nfused = 0
    for i in range(nR):
        if best[i] < 0:
            continue
        holder = kfpt[best[i]]
        if holder is not None:
            kind, j = holder
            if not (bad_L[j] if kind == "L" else bad_R[j]):
                repl[i] = j if kind == "L" else -2
        else:
            in_kf[i] = best[i]; obsR[i] += 1; kfpt[best[i]] = ("R", i)
        nfused += 1
    assert int(r["fuse2_n"][0]) == nfused and nfused > 30 and (r["fuse2_replace"] == repl).all()
    gR = r["fuse2_ptsR"].reshape(-1, 4)
    assert (gR[:, 2] == in_kf).all() and (gR[:, 3] == obsR).all()
    assert (r["fuse2_kf_points"] == np.array([-1 if p_ is None else (-2 if p_[0] == "L" else p_[1]) for p_ in kfpt])).all()
    # ---- 5. SearchBySim3: slot 0 = (KF2 features, KF1's points projected into KF2), slot 1 = (KF1 features, KF2's points projected into KF1)
    kf2, p12 = _cur(r, "sim3_12_"), _pts(r, "sim3_12_")
    kf1, p21 = _cur(r, "sim3_21_"), _pts(r, "sim3_21_")
    assert kf1["desc"].tobytes() == dL.tobytes() and kf2["desc"].tobytes() == dR.tobytes()
    pre = np.where((iL % 31 == 0) & has_L, (iL * 3) % nR, -1)
    assert not (p12["valid"] & ~(has_L & ~bad_L & (pre < 0))).any() and p12["valid"].sum() > 200 and p21["valid"].sum() > 200
    exp, en = oracle.search_by_sim3(kf1, kf2, p12, p21, sf, sf, 7.5)
    assert int(r["sim3_n"][0]) == en and (r["sim3_held"] == np.where(exp >= 0, exp, pre)).all() and en > 20


@pytest.mark.gpu
def test_adapter_batched_resident_matchers(pkg, oracle, tmp_path):
    """adapter/ORBmatcher_batch.cc: the loops of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:241-309), LoopClosing::ComputeSim3
    (src/LoopClosing.cc:293-323) and Tracking::Relocalization (src/Tracking.cc:1661-1682) over 20 keyframes, each as ONE call on resident
    keyframes (orbx_adapter::KeyFrameCache), against the oracle's per-pair loop on the data the driver wrote out; the single-pair adaptors
    give the same answers whether the keyframes are resident or not, and after a drop()"""
    w, h = 1241, 376
    left, right, _ = synth.stereo_pair(616, w, h)
    inp, outp = os.path.join(tmp_path, "in.bin"), os.path.join(tmp_path, "out.txt")
    with open(inp, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes()); f.write(left.tobytes()); f.write(right.tobytes())
    exe = _build_driver(str(tmp_path))
    run = subprocess.run([exe, "batch", inp, outp, "30"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    print(run.stdout.strip())
    r = _parse(outp)
    NK = 20
    assert int(r["cache_size"][0]) == NK + 1 and int(r["single_equals_batch"][0]) == 1 and int(r["after_drop_equal"][0]) == 1
    sf, s2 = _f32(r["scaleFactors"]), _f32(r["levelSigma2"])
    fx, cx, cy = np.float32(718.856), np.float32(607.1928), np.float32(185.2157)

    def kf(pre, flag_key):
        kp = _keys(r[pre + "keys"], pkg)
        return _featset(_u8(r[pre + "desc"]).reshape(-1, 32), kp, r[pre + flag_key], _f32(r[pre + "uright"]), nodes=100)
    cur_tri, cur_bow = kf("cur_", "has"), kf("cur_", "good")
    frame = dict(cur_bow); frame["flag"] = np.zeros(len(frame["desc"]), np.uint8)     # the frame of the relocalisation loop = the left eye as extracted
    ow = _f32(r["cur_Ow"]); F12s = _f32(r["F12s"]).reshape(NK, 9)
    total = 0
    for k in range(NK):
        pre = f"nb{k}_"
        nb_tri, nb_bow = kf(pre, "has"), kf(pre, "good")
        c2 = (ow + _f32(r[pre + "tcw"])).astype(np.float32)          # R2w = identity
        invz = np.float32(1.0) / c2[2]
        ex = np.float32(np.float32(fx * c2[0]) * invz) + cx
        ey = np.float32(np.float32(fx * c2[1]) * invz) + cy
        exp = oracle.search_for_triangulation(cur_tri, nb_tri, F12s[k], float(ex), float(ey), sf, s2, 0.6, False, False)
        assert (r[f"tri{k}"].reshape(-1, 2) == exp.reshape(-1, 2)).all() and len(r[f"tri{k}"]) == exp.size, f"triangulation, neighbour {k}"
        total += len(exp.reshape(-1, 2))
        if k % 7 == 0:
            exps = oracle.search_for_triangulation(cur_tri, nb_tri, F12s[k], float(ex), float(ey), sf, s2, 0.6, False, True)
            assert (r[f"triStereo{k}"].reshape(-1, 2) == exps.reshape(-1, 2)).all() and len(r[f"triStereo{k}"]) == exps.size, f"bOnlyStereo, neighbour {k}"
        e12, n12 = oracle.search_by_bow_kf_kf(cur_bow, nb_bow, 0.75, True)
        assert int(r["bow12_n"][k]) == n12 and (r[f"bow12_{k}"] == e12).all(), f"SearchByBoW(KF, KF), candidate {k}"
        ef, nf = oracle.search_by_bow_kf_f(nb_bow, frame, 0.75, True)
        assert int(r["bowF_n"][k]) == nf and (r[f"bowF_{k}"] == ef).all(), f"SearchByBoW(KF, F), candidate {k}"
    assert total > 100 and int(r["bow12_n"].sum()) > 200 and int(r["bowF_n"].sum()) > 200      # not vacuous
    t = r["time_ns"]
    assert len(t) == 8 and (t > 0).all()
