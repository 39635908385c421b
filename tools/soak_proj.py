"""Randomized parity soak of the eight projection-type searches against the oracle (scenes of tests/test_projection.py
with random sizes / densities / thresholds).  Run on the GPU box: python tools/soak_proj.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
from oracle import oracle_py as O
import test_projection as TP
pkg = ge.load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 99))
f32 = np.float32
t0 = time.time(); trial = 0
while time.time() - t0 < budget:
    trial += 1
    nc = int(rng.integers(1, 2500)); npnt = int(rng.integers(1, 2500)); dense = bool(trial % 3 == 0)
    cur, pts, sf = TP._scene(10000 + trial, nc, npnt, dense=dense, stereo_frac=float(rng.choice([0, 0.5, 1])),
                             obs_frac=float(rng.choice([0, 0.3, 0.7, 1])), occ_frac=float(rng.choice([0, 0.1, 0.5])))
    inv = (1.0 / (sf * sf)).astype(f32)
    th = float(rng.choice([1.0, 3.0, 7.0, 15.0, 40.0])); ori = bool(trial % 2); ratio = float(rng.choice([0.6, 0.8, 0.9]))
    m = pkg.ORBmatcher(ratio, ori)
    tag = f"trial {trial} nc {nc} np {npnt} dense {dense} th {th}"
    d = int(rng.integers(0, 3))
    g, n = m.SearchByProjectionLastFrame(cur, pts, sf, th, d, 40.0); e, en = O.search_by_projection_last(cur, pts, sf, th, d, 40.0, ori)
    assert n == en and (g == e).all(), "last " + tag
    p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
    g, n = m.SearchByProjectionMapPoints(cur, p2, sf, th); e, en = O.search_by_projection_points(cur, p2, sf, th, ratio)
    assert n == en and (g == e).all(), "points " + tag
    od = int(rng.choice([50, 64, 100]))
    g, n = m.SearchByProjectionKeyFrame(cur, pts, sf, th, od); e, en = O.search_by_projection_keyframe(cur, pts, sf, th, od, ori)
    assert n == en and (g == e).all(), "kf " + tag
    g, n = m.SearchByProjectionSim3(cur, pts, sf, th); e, en = O.search_by_projection_sim3(cur, pts, sf, th)
    assert n == en and (g == e).all(), "sim3p " + tag
    chi2 = trial % 2; md = int(rng.choice([50, 100]))
    bi, bd, n = m.Fuse(cur, p2, sf, inv if chi2 else None, th, md); ebi, ebd, en = O.window_best(cur, p2, sf, inv, th, chi2, md)
    assert n == en and (bi == ebi).all() and (bd == ebd).all(), "best " + tag
    if trial % 4 == 0:
        k = int(rng.integers(2, 1500))
        c1, c2, p12, p21, s2 = TP._sim3_scene(20000 + trial, k)
        g, n = m.SearchBySim3(c1, c2, p12, p21, s2, s2, th); e, en = O.search_by_sim3(c1, c2, p12, p21, s2, s2, th)
        assert n == en and (g == e).all(), "sim3 " + tag
        f1, f2, prev = TP._init_scene(30000 + trial, max(k, 70))
        win = int(rng.choice([10, 50, 100, 300]))
        g, n, _ = m.SearchForInitialization(f1, f2, prev, win); e, en = O.search_for_initialization(f1, f2, prev, win, ratio, ori)
        assert n == en and (g == e).all(), "init " + tag
    if trial % 50 == 0: print(f"{time.time() - t0:6.1f}s trials {trial}", flush=True)
print(f"projection soak done: {trial} random scenes, every search equal to the oracle")
