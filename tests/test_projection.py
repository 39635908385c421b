"""the two per-frame projection-guided matchers (reference src/ORBmatcher.cc:48-129 and
:1396-1553) with the Frame grid (src/Frame.cc:261-279, :386-457).  CPU: the oracle against an independent
plain-Python restatement.  GPU: the exact parallel fixpoint of the claims against the sequential oracle."""
import math

import numpy as np
import pytest

from tools import synth

POP = np.array([bin(i).count("1") for i in range(256)])
f32 = np.float32


def _scene(seed, n_cur=900, n_pts=700, w=752, h=480, dense=False, stereo_frac=0.5, obs_frac=0.7, occ_frac=0.1):
    rng = np.random.Generator(np.random.PCG64(seed))
    if dense:   # many features in a few clusters: long conflict chains, crowded windows
        cx = rng.uniform(100, w - 100, 6); cy = rng.uniform(80, h - 80, 6); k = rng.integers(0, 6, n_cur)
        x = (cx[k] + rng.normal(0, 14, n_cur)).astype(f32); y = (cy[k] + rng.normal(0, 14, n_cur)).astype(f32)
    else:
        x = rng.uniform(-5, w + 5, n_cur).astype(f32); y = rng.uniform(-5, h + 5, n_cur).astype(f32)
    octave = rng.integers(0, 8, n_cur).astype(np.int32)
    desc = rng.integers(0, 256, (n_cur, 32), dtype=np.uint8)
    if dense:
        desc[:] = synth.flip_bits(rng, np.repeat(desc[:8], n_cur // 8 + 1, axis=0)[:n_cur], 0.15)   # confusable descriptors
    cur = dict(x=x, y=y, octave=octave, angle=rng.uniform(0, 360, n_cur).astype(f32),
               u_right=np.where(rng.random(n_cur) < stereo_frac, x - rng.uniform(1, 30, n_cur), -1).astype(f32),
               desc=desc, occupied=(rng.random(n_cur) < occ_frac).astype(np.uint8), bounds=(0.0, 0.0, float(w), float(h)))
    src = rng.integers(0, n_cur, n_pts)
    u = (x[src] + rng.normal(0, 4, n_pts)).astype(f32); v = (y[src] + rng.normal(0, 4, n_pts)).astype(f32)
    invz = rng.uniform(0.02, 0.5, n_pts).astype(f32)
    invz[rng.random(n_pts) < 0.03] = -0.1
    pts = dict(u=u, v=v, aux=invz, level=np.clip(octave[src] + rng.integers(-1, 2, n_pts), 0, 7).astype(np.int32),
               angle=((cur["angle"][src] + rng.normal(0, 8, n_pts)) % 360).astype(f32),
               view_cos=rng.uniform(0.99, 1.0, n_pts).astype(f32), desc=synth.flip_bits(rng, desc[src], 0.06),
               valid=(rng.random(n_pts) < 0.92).astype(np.uint8), has_obs=(rng.random(n_pts) < obs_frac).astype(np.uint8))
    sf = np.array([f32(1.2) ** i for i in range(8)], f32)
    return cur, pts, sf


def _py_area(cur, grid, inv_w, inv_h, x, y, r, min_l, max_l):
    mnx, mny = f32(cur["bounds"][0]), f32(cur["bounds"][1])
    a = int(math.floor(f32(f32(f32(x - mnx) - r) * inv_w))); cx0 = max(0, a)
    if cx0 >= 64: return []
    cx1 = min(63, int(math.ceil(f32(f32(f32(x - mnx) + r) * inv_w))))
    if cx1 < 0: return []
    b = int(math.floor(f32(f32(f32(y - mny) - r) * inv_h))); cy0 = max(0, b)
    if cy0 >= 48: return []
    cy1 = min(47, int(math.ceil(f32(f32(f32(y - mny) + r) * inv_h))))
    if cy1 < 0: return []
    out = []
    check = min_l > 0 or max_l >= 0
    for ix in range(cx0, cx1 + 1):
        for iy in range(cy0, cy1 + 1):
            for k in grid.get((ix, iy), ()):
                if check:
                    if cur["octave"][k] < min_l: continue
                    if max_l >= 0 and cur["octave"][k] > max_l: continue
                if abs(f32(cur["x"][k] - x)) < r and abs(f32(cur["y"][k] - y)) < r:
                    out.append(k)
    return out


def _py_search(cur, pts, sf, mode, th, direction=0, mbf=0.0, ratio=0.8, check_ori=True):
    mnx, mny, mxx, mxy = [f32(v) for v in cur["bounds"]]
    inv_w = f32(f32(64) / f32(mxx - mnx)); inv_h = f32(f32(48) / f32(mxy - mny))
    grid = {}
    for i in range(len(cur["x"])):
        px = int(np.round(f32(f32(cur["x"][i] - mnx) * inv_w))) if True else 0
        px = int(math.floor(abs(float(f32(f32(cur["x"][i] - mnx) * inv_w))) + 0.5)) * (1 if f32(cur["x"][i] - mnx) >= 0 else -1)   # roundf: half away from zero
        py = int(math.floor(abs(float(f32(f32(cur["y"][i] - mny) * inv_h))) + 0.5)) * (1 if f32(cur["y"][i] - mny) >= 0 else -1)
        if 0 <= px < 64 and 0 <= py < 48:
            grid.setdefault((px, py), []).append(i)
    n = len(cur["x"])
    match = np.full(n, -1, np.int32); blocked = cur["occupied"].astype(bool).copy()
    nm = 0; rot = []
    for i in range(len(pts["u"])):
        if not pts["valid"][i]: continue
        u, v, lvl = f32(pts["u"][i]), f32(pts["v"][i]), int(pts["level"][i])
        if mode == 0:
            if pts["aux"][i] < 0: continue
            if u < mnx or u > mxx or v < mny or v > mxy: continue
            r = f32(f32(th) * sf[lvl])
            rng_l = (lvl, -1) if direction == 1 else (0, lvl) if direction == 2 else (lvl - 1, lvl + 1)
        else:
            rr = f32(2.5) if float(pts["view_cos"][i]) > 0.998 else f32(4.0)
            if float(f32(th)) != 1.0: rr = f32(rr * f32(th))
            r = f32(rr * sf[lvl]); rng_l = (lvl - 1, lvl)
        cand = _py_area(cur, grid, inv_w, inv_h, u, v, r, *rng_l)
        b1 = b2 = 256; l1 = l2 = -1; bi = -1
        for k in cand:
            if blocked[k]: continue
            if cur["u_right"][k] > 0:
                er = abs(f32(f32(u - f32(f32(mbf) * pts["aux"][i])) - cur["u_right"][k])) if mode == 0 else abs(f32(pts["aux"][i] - cur["u_right"][k]))
                if er > r: continue
            d = int(POP[pts["desc"][i] ^ cur["desc"][k]].sum())
            if d < b1: b2, l2, b1, l1, bi = b1, l1, d, int(cur["octave"][k]), k
            elif d < b2: b2, l2 = d, int(cur["octave"][k])
        if b1 <= 100:
            if mode == 1 and l1 == l2 and f32(b1) > f32(f32(ratio) * f32(b2)): continue
            match[bi] = i; blocked[bi] = bool(pts["has_obs"][i]); nm += 1
            if mode == 0 and check_ori:
                rt = f32(pts["angle"][i] - cur["angle"][bi])
                if rt < 0: rt = f32(rt + f32(360))
                b = int(math.floor(float(f32(rt * f32(f32(1) / f32(30)))) + 0.5))
                rot.append((0 if b == 30 else b, bi))
    if mode == 0 and check_ori:
        hist = np.bincount([b for b, _ in rot], minlength=30)
        m1 = m2 = m3 = 0; i1 = i2 = i3 = -1
        for i, s_ in enumerate(hist):
            if s_ > m1: m3, m2, m1, i3, i2, i1 = m2, m1, s_, i2, i1, i
            elif s_ > m2: m3, m2, i3, i2 = m2, s_, i2, i
            elif s_ > m3: m3, i3 = s_, i
        if f32(m2) < f32(0.1) * f32(m1): i2 = i3 = -1
        elif f32(m3) < f32(0.1) * f32(m1): i3 = -1
        for b, f_ in rot:
            if b not in (i1, i2, i3):
                match[f_] = -1; nm -= 1
    return match, nm


@pytest.mark.parametrize("seed,dense", [(1, False), (2, True)])
def test_oracle_vs_python(oracle, seed, dense):
    cur, pts, sf = _scene(seed, 500, 400, dense=dense)
    for direction in (0, 1, 2):
        m, n = oracle.search_by_projection_last(cur, pts, sf, 7.0, direction, 40.0, True)
        pm, pn = _py_search(cur, pts, sf, 0, 7.0, direction, 40.0)
        assert n == pn and (m == pm).all(), direction
    p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
    for th in (1.0, 3.0):
        m, n = oracle.search_by_projection_points(cur, p2, sf, th, 0.8)
        pm, pn = _py_search(cur, p2, sf, 1, th, ratio=0.8)
        assert n == pn and (m == pm).all(), th


@pytest.mark.gpu
@pytest.mark.parametrize("seed,dense,obs", [(3, False, 0.7), (4, True, 0.7), (5, True, 1.0), (6, True, 0.0), (7, False, 0.3)])
def test_hip_projection_parity(pkg, oracle, seed, dense, obs):
    cur, pts, sf = _scene(seed, 1200, 1000, dense=dense, obs_frac=obs)
    for ori in (True, False):
        mt = pkg.ORBmatcher(0.9, ori)
        for direction, th in ((0, 15.0), (1, 7.0), (2, 7.0)):
            got, n = mt.SearchByProjectionLastFrame(cur, pts, sf, th, direction, 40.0)
            exp, en = oracle.search_by_projection_last(cur, pts, sf, th, direction, 40.0, ori)
            assert n == en, (direction, n, en)
            assert (got == exp).all(), (direction, np.nonzero(got != exp)[0][:5])
    p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.6)):
        got, n = pkg.ORBmatcher(ratio, True).SearchByProjectionMapPoints(cur, p2, sf, th)
        exp, en = oracle.search_by_projection_points(cur, p2, sf, th, ratio)
        assert n == en and (got == exp).all(), th
        assert n > 50


@pytest.mark.gpu
def test_hip_projection_on_extracted_frames(pkg, oracle):
    """two consecutive synthetic frames: last-frame points projected with the known image shift"""
    seq = synth.sequence(21, 752, 480, 2)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(752, 480))
    k0, d0 = ex(seq[0]); k1, d1 = ex(seq[1])
    rng = np.random.Generator(np.random.PCG64(22))
    cur = dict(x=k1["x"], y=k1["y"], octave=k1["octave"], angle=k1["angle"], u_right=np.full(len(k1), -1, f32), desc=d1,
               occupied=np.zeros(len(k1), np.uint8), bounds=(0.0, 0.0, 752.0, 480.0))
    pts = dict(u=(k0["x"] - 2).astype(f32), v=(k0["y"] - 1).astype(f32), aux=np.full(len(k0), 0.1, f32), level=k0["octave"],
               angle=k0["angle"], view_cos=np.ones(len(k0), f32), desc=d0, valid=np.ones(len(k0), np.uint8),
               has_obs=(rng.random(len(k0)) < 0.8).astype(np.uint8))
    sf = ex.GetScaleFactors()
    got, n = pkg.ORBmatcher(0.9, True).SearchByProjectionLastFrame(cur, pts, sf, 15.0, 0, 0.0)
    exp, en = oracle.search_by_projection_last(cur, pts, sf, 15.0, 0, 0.0, True)
    assert n == en and (got == exp).all() and n > 300


@pytest.mark.gpu
def test_hip_projection_edge_cases(pkg, oracle):
    cur, pts, sf = _scene(8, 300, 200)
    mt = pkg.ORBmatcher(0.9, True)
    # no valid point / no feature
    p0 = dict(pts); p0["valid"] = np.zeros(len(pts["u"]), np.uint8)
    got, n = mt.SearchByProjectionLastFrame(cur, p0, sf, 7.0)
    assert n == 0 and (got == -1).all()
    # every feature occupied
    c1 = dict(cur); c1["occupied"] = np.ones(len(cur["x"]), np.uint8)
    got, n = mt.SearchByProjectionLastFrame(c1, pts, sf, 7.0)
    assert n == 0 and (got == -1).all()
    # all points on one feature: the classic chain (each point's best is taken by the previous one)
    m = 40
    c2 = dict(cur)
    p2 = {k_: (np.repeat(v[:1], m, axis=0) if isinstance(v, np.ndarray) else v) for k_, v in pts.items()}
    p2["valid"] = np.ones(m, np.uint8); p2["has_obs"] = np.ones(m, np.uint8); p2["aux"] = np.full(m, 0.1, f32)
    p2["u"] = np.full(m, cur["x"][5], f32); p2["v"] = np.full(m, cur["y"][5], f32); p2["level"] = np.full(m, cur["octave"][5], np.int32)
    got, n = mt.SearchByProjectionLastFrame(c2, p2, sf, 15.0)
    exp, en = oracle.search_by_projection_last(c2, p2, sf, 15.0, 0, 0.0, True)
    assert n == en and (got == exp).all()
    with pytest.raises(pkg.OrbxError):
        bad = dict(pts); bad["level"] = pts["level"].copy(); bad["level"][0] = 12
        mt.SearchByProjectionLastFrame(cur, bad, sf, 7.0)


# ---------------------------------------------------------------- the other projection-type searches of row f1:
# SearchByProjection(Frame, KeyFrame, ...) :1555-1685, SearchByProjection(KeyFrame, Scw, ...) :305-415,
# Fuse x2 :873-1164 (search half), SearchBySim3 :1166-1394

def _py_grid(cur):
    mnx, mny, mxx, mxy = [f32(v) for v in cur["bounds"]]
    inv_w = f32(f32(64) / f32(mxx - mnx)); inv_h = f32(f32(48) / f32(mxy - mny))
    grid = {}
    for i in range(len(cur["x"])):
        px = int(math.floor(abs(float(f32(f32(cur["x"][i] - mnx) * inv_w))) + 0.5)) * (1 if f32(cur["x"][i] - mnx) >= 0 else -1)
        py = int(math.floor(abs(float(f32(f32(cur["y"][i] - mny) * inv_h))) + 0.5)) * (1 if f32(cur["y"][i] - mny) >= 0 else -1)
        if 0 <= px < 64 and 0 <= py < 48:
            grid.setdefault((px, py), []).append(i)
    return grid, inv_w, inv_h


def _py_search2(kf, pts, sf, kind, th, max_dist, inv_sigma2=None):
    """kind: 'reloc' (Frame bounds, levels l-1..l+1, blocking), 'loop' (IsInImage, l-1..l, blocking),
    'best' (IsInImage, l-1..l, independent, optional chi2)"""
    mnx, mny, mxx, mxy = [f32(v) for v in kf["bounds"]]
    grid, inv_w, inv_h = _py_grid(kf)
    n = len(kf["x"])
    match = np.full(n, -1, np.int32)
    blocked = kf["occupied"].astype(bool).copy() if kind != "best" else np.zeros(n, bool)
    bi_out = np.full(len(pts["u"]), -1, np.int32); bd_out = np.full(len(pts["u"]), 256, np.int32)
    nm = 0
    for i in range(len(pts["u"])):
        if not pts["valid"][i]: continue
        u, v, lvl = f32(pts["u"][i]), f32(pts["v"][i]), int(pts["level"][i])
        if kind == "reloc":
            if u < mnx or u > mxx or v < mny or v > mxy: continue
        else:
            if not (u >= mnx and u < mxx and v >= mny and v < mxy): continue
        r = f32(f32(th) * sf[lvl])
        cand = _py_area(kf, grid, inv_w, inv_h, u, v, r, lvl - 1, lvl + 1) if kind == "reloc" else _py_area(kf, grid, inv_w, inv_h, u, v, r, -1, -1)
        b1 = 256; bi = -1
        for k in cand:
            if blocked[k]: continue
            kl = int(kf["octave"][k])
            if kind != "reloc" and (kl < lvl - 1 or kl > lvl): continue
            if inv_sigma2 is not None:
                ex = f32(u - kf["x"][k]); ey = f32(v - kf["y"][k])
                if kf["u_right"][k] >= 0:
                    er = f32(pts["aux"][i] - kf["u_right"][k])
                    e2 = f32(f32(f32(ex * ex) + f32(ey * ey)) + f32(er * er))
                    if float(f32(e2 * inv_sigma2[kl])) > 7.8: continue
                else:
                    e2 = f32(f32(ex * ex) + f32(ey * ey))
                    if float(f32(e2 * inv_sigma2[kl])) > 5.99: continue
            d = int(POP[pts["desc"][i] ^ kf["desc"][k]].sum())
            if d < b1: b1, bi = d, k
        if b1 <= max_dist:
            nm += 1
            bi_out[i] = bi; bd_out[i] = b1
            if kind != "best":
                match[bi] = i; blocked[bi] = True
    return match, nm, bi_out, bd_out


def _sim3_scene(seed, n=700):
    """two keyframes seeing the same synthetic structure; point i of each side belongs to keypoint i"""
    c1, p1, sf = _scene(seed, n, n, stereo_frac=0.0)
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    perm = rng.permutation(n)
    c2 = dict(c1)
    for k in ("x", "y", "octave", "angle", "u_right", "desc", "occupied"):
        c2[k] = c1[k][perm].copy()
    c2["x"] = (c2["x"] + rng.normal(0, 1.0, n)).astype(f32); c2["y"] = (c2["y"] + rng.normal(0, 1.0, n)).astype(f32)
    c2["desc"] = synth.flip_bits(rng, c2["desc"], 0.05)
    inv = np.argsort(perm)
    def side(src_kf, dst_kf, dst_of_src):
        m = len(src_kf["x"])
        return dict(u=(dst_kf["x"][dst_of_src] + rng.normal(0, 2.5, m)).astype(f32), v=(dst_kf["y"][dst_of_src] + rng.normal(0, 2.5, m)).astype(f32),
                    aux=np.zeros(m, f32), level=np.clip(dst_kf["octave"][dst_of_src] + rng.integers(0, 2, m), 0, 7).astype(np.int32),
                    angle=np.zeros(m, f32), view_cos=np.ones(m, f32), desc=src_kf["desc"], valid=(rng.random(m) < 0.85).astype(np.uint8),
                    has_obs=np.ones(m, np.uint8))
    return c1, c2, side(c1, c2, inv), side(c2, c1, perm), sf


@pytest.mark.parametrize("seed,dense", [(11, False), (12, True)])
def test_oracle_vs_python_other_searches(oracle, seed, dense):
    cur, pts, sf = _scene(seed, 500, 400, dense=dense)
    inv_s2 = (1.0 / (sf * sf)).astype(f32)
    m, n = oracle.search_by_projection_keyframe(cur, pts, sf, 10.0, 100, False)
    pm, pn, _, _ = _py_search2(cur, pts, sf, "reloc", 10.0, 100)
    assert n == pn and (m == pm).all()
    m, n = oracle.search_by_projection_sim3(cur, pts, sf, 10.0)
    pm, pn, _, _ = _py_search2(cur, pts, sf, "loop", 10.0, 50)
    assert n == pn and (m == pm).all()
    p2 = dict(pts); p2["aux"] = (pts["u"] - 8).astype(f32)
    for chi2 in (0, 1):
        bi, bd, n = oracle.window_best(cur, p2, sf, inv_s2, 4.0, chi2, 50)
        _, pn, pbi, pbd = _py_search2(cur, p2, sf, "best", 4.0, 50, inv_s2 if chi2 else None)
        assert n == pn and (bi == pbi).all() and (bd == pbd).all(), chi2
    assert n > 20


def test_oracle_sim3_agreement(oracle):
    c1, c2, p12, p21, sf = _sim3_scene(13, 400)
    m12, n = oracle.search_by_sim3(c1, c2, p12, p21, sf, sf, 7.5)
    _, _, b1, _ = _py_search2(c2, p12, sf, "best", 7.5, 100)
    _, _, b2, _ = _py_search2(c1, p21, sf, "best", 7.5, 100)
    exp = np.array([b1[i] if b1[i] >= 0 and b2[b1[i]] == i else -1 for i in range(len(b1))], np.int32)
    assert (m12 == exp).all() and n == (exp >= 0).sum() and n > 100


@pytest.mark.gpu
@pytest.mark.parametrize("seed,dense", [(14, False), (15, True), (16, True)])
def test_hip_other_searches_parity(pkg, oracle, seed, dense):
    cur, pts, sf = _scene(seed, 1500, 1200, dense=dense)
    inv_s2 = (1.0 / (sf * sf)).astype(f32)
    for ori in (True, False):
        mt = pkg.ORBmatcher(0.9, ori)
        for th, od in ((10.0, 100), (3.0, 64)):
            got, n = mt.SearchByProjectionKeyFrame(cur, pts, sf, th, od)
            exp, en = oracle.search_by_projection_keyframe(cur, pts, sf, th, od, ori)
            assert n == en and (got == exp).all(), (ori, th)
    mt = pkg.ORBmatcher(0.9, True)
    for th in (10.0, 4.0):
        got, n = mt.SearchByProjectionSim3(cur, pts, sf, th)
        exp, en = oracle.search_by_projection_sim3(cur, pts, sf, th)
        assert n == en and (got == exp).all(), th
    p2 = dict(pts); p2["aux"] = (pts["u"] - 8).astype(f32)
    for th, chi2, md in ((3.0, 1, 50), (4.0, 0, 50), (7.5, 0, 100), (12.0, 1, 50)):
        bi, bd, n = mt.Fuse(cur, p2, sf, inv_s2 if chi2 else None, th, md)
        ebi, ebd, en = oracle.window_best(cur, p2, sf, inv_s2, th, chi2, md)
        assert n == en and (bi == ebi).all() and (bd == ebd).all(), (th, chi2)
    assert n > 30


@pytest.mark.gpu
def test_hip_search_by_sim3(pkg, oracle):
    for seed in (17, 18):
        c1, c2, p12, p21, sf = _sim3_scene(seed, 1100)
        got, n = pkg.ORBmatcher(0.75, True).SearchBySim3(c1, c2, p12, p21, sf, sf, 7.5)
        exp, en = oracle.search_by_sim3(c1, c2, p12, p21, sf, sf, 7.5)
        assert n == en and (got == exp).all() and n > 300
    with pytest.raises(pkg.OrbxError):
        bad = {k: v[:-1] if isinstance(v, np.ndarray) else v for k, v in p12.items()}
        pkg.ORBmatcher(0.75, True).SearchBySim3(c1, c2, bad, p21, sf, sf, 7.5)


# ---------------------------------------------------------------- SearchForInitialization (:430-556)

def _init_scene(seed, n=900, shift=6.0, noise=0.04):
    c1, _, sf = _scene(seed, n, 10, stereo_frac=0.0)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    c1["octave"] = np.where(rng.random(n) < 0.55, 0, rng.integers(1, 8, n)).astype(np.int32)
    perm = rng.permutation(n)
    c2 = dict(c1)
    for k in ("x", "y", "octave", "angle", "u_right", "desc", "occupied"):
        c2[k] = c1[k][perm].copy()
    c2["x"] = (c2["x"] + shift + rng.normal(0, 1.5, n)).astype(f32); c2["y"] = (c2["y"] + rng.normal(0, 1.5, n)).astype(f32)
    c2["angle"] = ((c2["angle"] + rng.normal(0, 6, n)) % 360).astype(f32)
    c2["desc"] = synth.flip_bits(rng, c2["desc"], noise)
    # a handful of look-alike features so that matches get stolen and the ratio test bites
    dup = rng.integers(0, n, 60); src = rng.integers(0, n, 60)
    c2["desc"][dup] = synth.flip_bits(rng, c2["desc"][src], 0.02); c2["octave"][dup] = c2["octave"][src]
    prev = np.stack([c1["x"], c1["y"]], 1).astype(f32)
    return c1, c2, prev


def _py_init(f1, f2, prev, win, ratio, check_ori):
    grid, inv_w, inv_h = _py_grid(f2)
    n1, n2 = len(f1["x"]), len(f2["x"])
    m12 = np.full(n1, -1, np.int32); m21 = np.full(n2, -1, np.int32); md = np.full(n2, 2**31 - 1, np.int64)
    nm = 0; rot = []
    for i1 in range(n1):
        if f1["octave"][i1] > 0: continue
        cand = _py_area(f2, grid, inv_w, inv_h, f32(prev[i1, 0]), f32(prev[i1, 1]), f32(win), 0, 0)
        b1 = b2 = 2**31 - 1; bi = -1
        for k in cand:
            d = int(POP[f1["desc"][i1] ^ f2["desc"][k]].sum())
            if md[k] <= d: continue
            if d < b1: b2, b1, bi = b1, d, k
            elif d < b2: b2 = d
        if b1 <= 50 and f32(b1) < f32(f32(b2) * f32(ratio)):
            if m21[bi] >= 0: m12[m21[bi]] = -1; nm -= 1
            m12[i1] = bi; m21[bi] = i1; md[bi] = b1; nm += 1
            if check_ori:
                rt = f32(f1["angle"][i1] - f2["angle"][bi])
                if rt < 0: rt = f32(rt + f32(360))
                b = int(math.floor(float(f32(rt * f32(f32(1) / f32(30)))) + 0.5))
                rot.append((0 if b == 30 else b, i1))
    if check_ori:
        hist = np.bincount([b for b, _ in rot], minlength=30)
        mx1 = mx2 = mx3 = 0; i1 = i2 = i3 = -1
        for i, s_ in enumerate(hist):
            if s_ > mx1: mx3, mx2, mx1, i3, i2, i1 = mx2, mx1, s_, i2, i1, i
            elif s_ > mx2: mx3, mx2, i3, i2 = mx2, s_, i2, i
            elif s_ > mx3: mx3, i3 = s_, i
        if f32(mx2) < f32(0.1) * f32(mx1): i2 = i3 = -1
        elif f32(mx3) < f32(0.1) * f32(mx1): i3 = -1
        for b, idx1 in rot:
            if b not in (i1, i2, i3) and m12[idx1] >= 0:
                m12[idx1] = -1; nm -= 1
    return m12, nm


def test_oracle_vs_python_initialization(oracle):
    f1, f2, prev = _init_scene(31, 500)
    for ori in (True, False):
        m, n = oracle.search_for_initialization(f1, f2, prev, 100, 0.9, ori)
        pm, pn = _py_init(f1, f2, prev, 100, 0.9, ori)
        assert n == pn and (m == pm).all()
    assert n > 100


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,win,ratio", [(32, 1500, 100, 0.9), (33, 2000, 100, 0.9), (34, 800, 30, 0.7), (35, 1200, 300, 0.95)])
def test_hip_search_for_initialization(pkg, oracle, seed, n, win, ratio):
    f1, f2, prev = _init_scene(seed, n)
    for ori in (True, False):
        got, gn, new_xy = pkg.ORBmatcher(ratio, ori).SearchForInitialization(f1, f2, prev, win)
        exp, en = oracle.search_for_initialization(f1, f2, prev, win, ratio, ori)
        assert gn == en and (got == exp).all(), (ori, gn, en, np.nonzero(got != exp)[0][:5])
        m = exp >= 0
        assert (new_xy[m, 0] == f2["x"][exp[m]]).all() and (new_xy[~m] == prev[~m]).all()
    assert en > 0.25 * n * 0.5
