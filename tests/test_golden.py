"""Committed fixtures (tests/golden, made by tools/gen_golden.py from the CPU oracle).
CPU: the oracle still reproduces them (regression pin).  GPU: the HIP path reproduces them too,
without executing anything under oracle/ at test time."""
import glob
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXTRACT = sorted(glob.glob(os.path.join(GOLD, "extract_*.npz")))


def _load(path):
    return np.load(path, allow_pickle=False)


def _featset(z, pre, flag=None):
    return dict(desc=z[pre + "_desc"], node_id=z[pre + "_node_id"], node_off=z[pre + "_node_off"], feat=z[pre + "_feat"],
                flag=z[pre + "_flag"] if flag is None else flag, angle=z[pre + "_angle"], x=z[pre + "_x"], y=z[pre + "_y"],
                octave=z[pre + "_octave"], u_right=z[pre + "_u_right"])


def test_fixtures_present():
    assert len(EXTRACT) == 3
    for f in ("stereo_640x480_s104.npz", "bow_s105.npz", "projection_s201.npz", "frontend_s204.npz"):
        assert os.path.exists(os.path.join(GOLD, f))


@pytest.mark.parametrize("path", EXTRACT, ids=[os.path.basename(p) for p in EXTRACT])
def test_oracle_reproduces_extract_fixture(oracle, path):
    z = _load(path)
    o = oracle.Oracle(int(z["nfeatures"]), 1.2, int(z["nlevels"]), 20, 7)
    k, d = o.extract(z["image"])
    assert k.tobytes() == z["kps"].tobytes() and d.tobytes() == z["desc"].tobytes()
    for l in range(int(z["nlevels"])):
        assert zlib.crc32(o.level(l).tobytes()) == z["level_crc"][l]
        assert len(o.candidates(l)[0]) == z["ncand"][l]
        assert zlib.crc32(np.stack(o.candidates(l)).astype(np.int32).tobytes()) == z["cand_crc"][l]
        assert o.nkeypoints(l) == z["nkp"][l]


def test_oracle_reproduces_stereo_and_bow_fixtures(oracle):
    z = _load(os.path.join(GOLD, "stereo_640x480_s104.npz"))
    oL, oR = oracle.Oracle(1000), oracle.Oracle(1000)
    kL, dL = oL.extract(z["left"]); kR, dR = oR.extract(z["right"])
    assert kL.tobytes() == z["kL"].tobytes() and dR.tobytes() == z["dR"].tobytes()
    ur, dp = oracle.stereo_match(oL, oR, kL, dL, kR, dR, float(z["bf"]), float(z["b"]))
    assert ur.tobytes() == z["u_right"].tobytes() and dp.tobytes() == z["depth"].tobytes()
    b = _load(os.path.join(GOLD, "bow_s105.npz"))
    kf, fr = _featset(b, "kf"), _featset(b, "fr")
    m, n = oracle.search_by_bow_kf_f(kf, fr, 0.75, True)
    assert n == int(b["n_kf_f"]) and (m == b["m_kf_f"]).all()
    m, n = oracle.search_by_bow_kf_kf(kf, fr, 0.75, True)
    assert n == int(b["n_kf_kf"]) and (m == b["m_kf_kf"]).all()
    p = oracle.search_for_triangulation(_featset(b, "kf", b["tri_flag_kf"]), _featset(b, "fr", b["tri_flag_fr"]), b["F12"],
                                        300.0, 200.0, b["sf"], b["sig2"], 0.6, False, False)
    assert (p == b["tri_pairs"]).all()


def _search_by_bow_python(kf, f, ratio):
    """independent plain-Python restatement of SearchByBoW(KF,F) (reference src/ORBmatcher.cc:171-303)
    without the orientation filter, with dicts standing in for std::map"""
    pop = np.array([bin(i).count("1") for i in range(256)])
    fv_kf = {int(i): kf["feat"][kf["node_off"][j]:kf["node_off"][j + 1]] for j, i in enumerate(kf["node_id"])}
    fv_f = {int(i): f["feat"][f["node_off"][j]:f["node_off"][j + 1]] for j, i in enumerate(f["node_id"])}
    match = np.full(len(f["desc"]), -1, np.int32)
    for node in sorted(set(fv_kf) & set(fv_f)):
        for i_kf in fv_kf[node]:
            if not kf["flag"][i_kf]:
                continue
            b1, b2, bi = 256, 256, -1
            for i_f in fv_f[node]:
                if match[i_f] >= 0:
                    continue
                dist = int(pop[kf["desc"][i_kf] ^ f["desc"][i_f]].sum())
                if dist < b1:
                    b2, b1, bi = b1, dist, i_f
                elif dist < b2:
                    b2 = dist
            if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                match[bi] = i_kf
    return match


def test_bow_oracle_vs_plain_python(oracle):
    b = _load(os.path.join(GOLD, "bow_s105.npz"))
    kf, fr = _featset(b, "kf"), _featset(b, "fr")
    m, n = oracle.search_by_bow_kf_f(kf, fr, 0.75, False)
    assert (m == _search_by_bow_python(kf, fr, 0.75)).all() and n == (m >= 0).sum()


# ---------------------------------------------------------------------------------- GPU vs fixtures

@pytest.mark.gpu
@pytest.mark.parametrize("path", EXTRACT, ids=[os.path.basename(p) for p in EXTRACT])
def test_hip_reproduces_extract_fixture(pkg, path):
    z = _load(path)
    h, w = z["image"].shape
    ex = pkg.ORBextractor(int(z["nfeatures"]), 1.2, int(z["nlevels"]), 20, 7, device=0, max_size=(w, h))
    k, d = ex(z["image"])
    assert k.tobytes() == z["kps"].tobytes() and d.tobytes() == z["desc"].tobytes()
    for l in range(int(z["nlevels"])):
        assert zlib.crc32(ex.pyramid_level(l).tobytes()) == z["level_crc"][l]
        assert zlib.crc32(np.stack(ex.debug_candidates(l)).astype(np.int32).tobytes()) == z["cand_crc"][l]
    assert (ex.debug_level_counts() == z["nkp"]).all()


@pytest.mark.gpu
def test_hip_reproduces_stereo_and_bow_fixtures(pkg):
    z = _load(os.path.join(GOLD, "stereo_640x480_s104.npz"))
    exL = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(640, 480))
    exR = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(640, 480))
    kL, dL = exL(z["left"]); kR, dR = exR(z["right"])
    assert kL.tobytes() == z["kL"].tobytes() and dL.tobytes() == z["dL"].tobytes()
    assert kR.tobytes() == z["kR"].tobytes() and dR.tobytes() == z["dR"].tobytes()
    ur, dp = pkg.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, float(z["bf"]), float(z["b"]))
    assert ur.tobytes() == z["u_right"].tobytes() and dp.tobytes() == z["depth"].tobytes()
    b = _load(os.path.join(GOLD, "bow_s105.npz"))
    kf, fr = _featset(b, "kf"), _featset(b, "fr")
    m = pkg.ORBmatcher(0.75, True)
    got, n = m.SearchByBoW(kf, fr)
    assert n == int(b["n_kf_f"]) and (got == b["m_kf_f"]).all()
    fr2 = dict(fr); fr2["kind"] = "keyframe"
    got, n = m.SearchByBoW(kf, fr2)
    assert n == int(b["n_kf_kf"]) and (got == b["m_kf_kf"]).all()
    p = pkg.ORBmatcher(0.6, False).SearchForTriangulation(_featset(b, "kf", b["tri_flag_kf"]), _featset(b, "fr", b["tri_flag_fr"]),
                                                         b["F12"], 300.0, 200.0, b["sf"], b["sig2"])
    assert (p == b["tri_pairs"]).all()


# ---------------------------------------------------------------- fixtures of the next rows (f1-f4)

def _group(z, pre):
    d = {k[len(pre) + 1:]: z[k] for k in z.files if k.startswith(pre + "_")}
    if "bounds" in d:
        d["bounds"] = tuple(float(v) for v in d["bounds"])
    return d


def _projection_cases(z, api):
    """(name, got) pairs; api = the oracle module or a dict of GPU callables with the same signatures"""
    cur, pts, sf, inv = _group(z, "cur"), _group(z, "pts"), z["sf"], z["inv_s2"]
    f32 = np.float32
    for d in (0, 1, 2):
        yield f"last_{d}", api["last"](cur, pts, sf, 7.0, d, 40.0)
    p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
    yield "points", api["points"](cur, p2, sf, 3.0, 0.8)
    yield "kf", api["kf"](cur, pts, sf, 10.0, 100)
    yield "sim3p", api["sim3p"](cur, pts, sf, 10.0)
    yield "sim3", api["sim3"](_group(z, "s1"), _group(z, "s2"), _group(z, "p12"), _group(z, "p21"), sf, 7.5)
    yield "init", api["init"](_group(z, "i1"), _group(z, "i2"), z["init_prev"], 100, 0.9)


def _check_projection(z, api, best):
    for name, (m, n) in _projection_cases(z, api):
        assert n == int(z[name + "_n"]) and (np.asarray(m) == z[name + "_m"]).all(), name
    cur, pts, sf, inv = _group(z, "cur"), _group(z, "pts"), z["sf"], z["inv_s2"]
    p3 = dict(pts); p3["aux"] = (pts["u"] - 8).astype(np.float32)
    for chi2 in (0, 1):
        bi, bd, n = best(cur, p3, sf, inv, 4.0, chi2, 50)
        assert n == int(z[f"best_{chi2}_n"]) and (bi == z[f"best_{chi2}_idx"]).all() and (bd == z[f"best_{chi2}_dist"]).all(), chi2


def test_oracle_reproduces_next_row_fixtures(oracle):
    z = _load(os.path.join(GOLD, "projection_s201.npz"))
    O = oracle
    api = dict(last=lambda c, p, sf, th, d, mbf: O.search_by_projection_last(c, p, sf, th, d, mbf, True),
               points=lambda c, p, sf, th, r: O.search_by_projection_points(c, p, sf, th, r),
               kf=lambda c, p, sf, th, od: O.search_by_projection_keyframe(c, p, sf, th, od, True),
               sim3p=lambda c, p, sf, th: O.search_by_projection_sim3(c, p, sf, th),
               sim3=lambda a, b, p, q, sf, th: O.search_by_sim3(a, b, p, q, sf, sf, th),
               init=lambda a, b, prev, win, r: O.search_for_initialization(a, b, prev, win, r, True))
    _check_projection(z, api, O.window_best)
    f = _load(os.path.join(GOLD, "frontend_s204.npz"))
    t = O.Vocabulary(10, 3, f["voc_parent"], f["voc_leaf"], f["voc_desc"], f["voc_weight"]).transform(f["desc"], 1)
    for k in ("word_id", "node_id", "bow_id", "fv_node_id", "fv_node_off", "fv_feat"):
        assert (t[k] == f["bow_" + k]).all(), k
    assert t["bow_val"].tobytes() == f["bow_bow_val"].tobytes() and t["word_weight"].tobytes() == f["bow_word_weight"].tobytes()
    off = f["obs_off"]
    assert [O.distinctive_descriptor(f["obs_flat"][off[i]:off[i + 1]]) for i in range(len(off) - 1)] == f["obs_best"].tolist()
    assert (O.cvt_gray(f["rgb"], 1) == f["gray_rgb"]).all() and (O.cvt_gray(f["rgb"], 0) == f["gray_bgr"]).all()
    assert (O.remap_bilinear(f["img"], f["map_x"], f["map_y"]) == f["remapped"]).all()
    c = f["und_cam"]
    assert O.undistort_points(f["und_xy"], float(c[0]), float(c[1]), float(c[2]), float(c[3]), f["und_dist"]).tobytes() == f["und_out"].tobytes()


@pytest.mark.gpu
def test_hip_reproduces_next_row_fixtures(pkg):
    """the HIP paths of rows f1-f4 against the committed fixtures: nothing under oracle/ runs in this test"""
    z = _load(os.path.join(GOLD, "projection_s201.npz"))
    m = pkg.ORBmatcher(0.8, True); m9 = pkg.ORBmatcher(0.9, True)
    api = dict(last=lambda c, p, sf, th, d, mbf: m.SearchByProjectionLastFrame(c, p, sf, th, d, mbf),
               points=lambda c, p, sf, th, r: pkg.ORBmatcher(r, True).SearchByProjectionMapPoints(c, p, sf, th),
               kf=lambda c, p, sf, th, od: m.SearchByProjectionKeyFrame(c, p, sf, th, od),
               sim3p=lambda c, p, sf, th: m.SearchByProjectionSim3(c, p, sf, th),
               sim3=lambda a, b, p, q, sf, th: m.SearchBySim3(a, b, p, q, sf, sf, th),
               init=lambda a, b, prev, win, r: pkg.ORBmatcher(r, True).SearchForInitialization(a, b, prev, win)[:2])
    _check_projection(z, api, lambda c, p, sf, inv, th, chi2, md: m.Fuse(c, p, sf, inv if chi2 else None, th, md))
    f = _load(os.path.join(GOLD, "frontend_s204.npz"))
    t = pkg.ORBVocabulary(10, 3, f["voc_parent"], f["voc_leaf"], f["voc_desc"], f["voc_weight"]).transform(f["desc"], 1)
    for k in ("word_id", "node_id", "bow_id", "fv_node_id", "fv_node_off", "fv_feat"):
        assert (t[k] == f["bow_" + k]).all(), k
    assert t["bow_val"].tobytes() == f["bow_bow_val"].tobytes()
    off = f["obs_off"]
    best = pkg.ComputeDistinctiveDescriptors([f["obs_flat"][off[i]:off[i + 1]] for i in range(len(off) - 1)])
    assert (best == f["obs_best"]).all()
    h, w = f["img"].shape
    ex = pkg.ORBextractor(300, 1.2, 4, 20, 7, device=0, max_size=(w, h))
    assert (ex.extract_color(f["rgb"], rgb=True, want_gray=True)[2] == f["gray_rgb"]).all()
    assert (ex.extract_color(f["rgb"], rgb=False, want_gray=True)[2] == f["gray_bgr"]).all()
    rect = pkg.Rectifier((w, h), f["map_x"], f["map_y"])
    rh, rw = f["map_x"].shape
    ex2 = pkg.ORBextractor(300, 1.2, 4, 20, 7, device=0, max_size=(rw, rh))
    assert (ex2.extract_rectified(rect, f["img"], want_rect=True)[2] == f["remapped"]).all()
    c = f["und_cam"]
    assert pkg.UndistortKeyPoints(f["und_xy"], float(c[0]), float(c[1]), float(c[2]), float(c[3]), f["und_dist"]).tobytes() == f["und_out"].tobytes()
