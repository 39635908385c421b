"""Per-call and batched forms of the three north-star matchers (orb-slam2_amd/csrc/orbx_match.hip) against the CPU oracle,
bit-exact: host-pointer calls, the batched loops of LocalMapping::CreateNewMapPoints (reference src/LocalMapping.cc:241-309)
and LoopClosing::ComputeSim3 (src/LoopClosing.cc:293-323), and the same searches on resident keyframes (orbx_kf_*).
Edge cases as the reference meets them: neighbours of unequal size, bOnlyStereo, a neighbour whose features all hold map
points, an empty FeatureVector, vocabulary nodes larger than one 64 x 64 tile / than the 4096-column register form."""
import threading

import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


def _mk(rng, voc, desc, x, y, octave, angle, flag_p=0.6, stereo_p=0.3):
    ids, off, feat = voc.feature_vector(desc)
    n = len(desc)
    return dict(desc=np.ascontiguousarray(desc), node_id=ids, node_off=off, feat=feat, flag=(rng.random(n) < flag_p).astype(np.uint8),
                angle=angle.astype(np.float32), x=x.astype(np.float32), y=y.astype(np.float32), octave=octave.astype(np.int32),
                u_right=np.where(rng.random(n) < stereo_p, 5.0, -1.0).astype(np.float32))


def _scene(seed, n1=1000, n2s=(1000, 700, 1300), k=10, flip=0.04):
    """current keyframe + neighbours seen from a camera translated along x (epipolar lines = rows), per-pair F12 / epipole"""
    rng = np.random.Generator(np.random.PCG64(seed))
    nmax = max((n1,) + tuple(n2s))
    d = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
    voc = synth.Vocab2(seed + 1, k=k)
    if k > 1:
        voc.seed_from(d, rng)
    x = rng.uniform(20, 1220, nmax); y = rng.uniform(20, 356, nmax)
    octave = rng.integers(0, 8, nmax); angle = rng.uniform(0, 360, nmax)
    cur = _mk(rng, voc, d[:n1], x[:n1], y[:n1], octave[:n1], angle[:n1])
    kfs, Fs, eps = [], [], []
    for n2 in n2s:
        perm = rng.permutation(nmax)[:n2]
        dk = synth.flip_bits(rng, d, flip)[perm]
        kfs.append(_mk(rng, voc, dk, x[perm] + rng.uniform(-40, 40), y[perm] + rng.normal(0, 0.4, n2), octave[perm],
                       (angle[perm] + rng.normal(0, 5, n2)) % 360))
        F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + np.float32(2e-7) * rng.normal(0, 1, (3, 3)).astype(np.float32)
        Fs.append(F.astype(np.float32)); eps.append((float(rng.uniform(-500, 1700)), float(rng.uniform(0, 376))))
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    return cur, kfs, Fs, eps, sf, (sf * sf).astype(np.float32)


def _tri_flags(s, mod, lo):
    t = dict(s); t["flag"] = (np.arange(len(s["desc"])) % mod < lo).astype(np.uint8)   # "already has a MapPoint"
    return t


@pytest.mark.parametrize("only_stereo,ori", [(False, True), (False, False), (True, True)])
def test_triangulation_batch(pkg, oracle, only_stereo, ori):
    cur, kfs, Fs, eps, sf, sg = _scene(11, 1000, (1000, 700, 1300, 900, 40))
    cur = _tri_flags(cur, 5, 2)
    kfs = [_tri_flags(k, 5, 1) for k in kfs]
    kfs[3]["flag"][:] = 1                                            # every feature of this neighbour already has a map point
    kfs[4] = dict(kfs[4], node_id=np.zeros(0, np.uint32), node_off=np.zeros(1, np.int32), feat=np.zeros(0, np.uint32))  # empty FeatureVector
    m = pkg.ORBmatcher(0.6, ori)
    got = m.SearchForTriangulationBatch(cur, kfs, Fs, eps, sf, sg, bOnlyStereo=only_stereo)
    total = 0
    for i, kf in enumerate(kfs):
        exp = oracle.search_for_triangulation(cur, kf, Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, ori, only_stereo)
        assert got[i].shape == exp.shape and (got[i] == exp).all(), f"neighbour {i}"
        one = m.SearchForTriangulation(cur, kf, Fs[i], eps[i][0], eps[i][1], sf, sg, bOnlyStereo=only_stereo)
        assert one.shape == exp.shape and (one == exp).all(), f"neighbour {i} (single call)"
        total += len(exp)
    assert len(got[3]) == 0 and len(got[4]) == 0
    if not only_stereo:
        assert total > 100


def test_bow_kf_kf_batch(pkg, oracle):
    cur, kfs, *_ = _scene(12, 1000, (1000, 600, 1400, 50))
    kfs[3] = dict(kfs[3], node_id=np.zeros(0, np.uint32), node_off=np.zeros(1, np.int32), feat=np.zeros(0, np.uint32))
    kfs[1]["flag"][:] = 0
    for ratio, ori in ((0.75, True), (0.9, False)):
        m = pkg.ORBmatcher(ratio, ori)
        got, n = m.SearchByBoWKeyFrames(cur, kfs)
        for i, kf in enumerate(kfs):
            exp, en = oracle.search_by_bow_kf_kf(cur, kf, ratio, ori)
            assert n[i] == en and (got[i] == exp).all(), f"candidate {i}"
        assert n[0] > 50 and n[1] == 0 and n[3] == 0
        # the other direction, one call per pair, and (KF, F) for every keyframe against the current one
        for i, kf in enumerate(kfs):
            b = dict(cur); b["kind"] = "keyframe"
            g1, n1 = m.SearchByBoW(kf, b)
            e1, en1 = oracle.search_by_bow_kf_kf(kf, cur, ratio, ori)
            assert n1 == en1 and (g1 == e1).all(), i
        gf, nf = m.SearchByBoWBatch(kfs, cur)
        for i, kf in enumerate(kfs):
            ef, enf = oracle.search_by_bow_kf_f(kf, cur, ratio, ori)
            assert nf[i] == enf and (gf[i] == ef).all(), i


@pytest.mark.parametrize("k,n", [(2, 1000), (1, 300), (3, 2500)])
def test_large_nodes(pkg, oracle, k, n):
    """k = 2: four nodes of ~250 features (several column chunks and row tiles per node, triangulation nodes split by rows);
    k = 1: everything in one node"""
    cur, kfs, Fs, eps, sf, sg = _scene(13 + k, n, (n, n - 37), k=k)
    m = pkg.ORBmatcher(0.8, True)
    for i, kf in enumerate(kfs):
        g, c = m.SearchByBoW(kf, cur)
        e, ec = oracle.search_by_bow_kf_f(kf, cur, 0.8, True)
        assert c == ec and (g == e).all()
        b = dict(kf); b["kind"] = "keyframe"
        g, c = m.SearchByBoW(cur, b)
        e, ec = oracle.search_by_bow_kf_kf(cur, kf, 0.8, True)
        assert c == ec and (g == e).all()
    tc = _tri_flags(cur, 4, 1); tk = [_tri_flags(kf, 3, 1) for kf in kfs]
    got = m.SearchForTriangulationBatch(tc, tk, Fs, eps, sf, sg)
    for i, kf in enumerate(tk):
        exp = oracle.search_for_triangulation(tc, kf, Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, True, False)
        assert got[i].shape == exp.shape and (got[i] == exp).all()


@pytest.mark.parametrize("in_memory", [False, True])
@pytest.mark.parametrize("k,n", [(10, 1000), (2, 1000), (1, 300), (3, 2500)])
def test_single_pair_item_transports(pkg, oracle, k, n, in_memory):
    """A call with one pair hands its work items to the kernel by value (k_match_v: at most 144 items, in the kernel-argument segment)
    or, like a batch, through mapped host memory (k_match; also what 3 x 3 nodes of 278 features split by rows -- 180 items -- take):
    identical results, host pointers and resident keyframes"""
    cur, kfs, Fs, eps, sf, sg = _scene(61 + k, n, (n - 11,), k=k)
    kf = kfs[0]
    tc = _tri_flags(cur, 4, 1); tk = _tri_flags(kf, 3, 1)
    m = pkg.ORBmatcher(0.8, True)
    pkg.orbx.debug_set_match_items(in_memory)
    try:
        for rnd in range(2):                                           # (the second call reuses the thread's context: tickets, scratch rows back at -1)
            g, c = m.SearchByBoW(kf, cur)
            e, ec = oracle.search_by_bow_kf_f(kf, cur, 0.8, True)
            assert c == ec and (g == e).all()
            b = dict(kf); b["kind"] = "keyframe"
            g, c = m.SearchByBoW(cur, b)
            e, ec = oracle.search_by_bow_kf_kf(cur, kf, 0.8, True)
            assert c == ec and (g == e).all()
            for only_stereo in (False, True):
                got = m.SearchForTriangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg, bOnlyStereo=only_stereo)
                exp = oracle.search_for_triangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg, 0.6, True, only_stereo)
                assert got.shape == exp.shape and (got == exp).all()
        dcur = pkg.DeviceKeyFrame(cur); dk = pkg.DeviceKeyFrame(kf)
        g, c = m.SearchByBoWResident(dk, kf["flag"], dcur)
        e, ec = oracle.search_by_bow_kf_f(kf, cur, 0.8, True)
        assert c == ec and (g == e).all()
        got = m.SearchForTriangulationResident(dcur, tc["flag"], [dk], [tk["flag"]], Fs, eps, sf, sg)
        exp = oracle.search_for_triangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg, 0.6, True, False)
        assert got[0].shape == exp.shape and (got[0] == exp).all()
    finally:
        pkg.orbx.debug_set_match_items(False)


def test_node_beyond_register_form_takes_legacy_kernels(pkg, oracle):
    """one vocabulary node with more than 4096 second-side features: SearchByBoW falls back to the kernels of orbx_bow.hip"""
    cur, kfs, Fs, eps, sf, sg = _scene(21, 4300, (4200,), k=1, flip=0.02)
    m = pkg.ORBmatcher(0.8, True)
    g, c = m.SearchByBoW(kfs[0], cur)
    e, ec = oracle.search_by_bow_kf_f(kfs[0], cur, 0.8, True)
    assert c == ec and (g == e).all()
    tc = _tri_flags(cur, 4, 1); tk = _tri_flags(kfs[0], 3, 1)
    got = m.SearchForTriangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg)
    exp = oracle.search_for_triangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg, 0.6, True, False)
    assert got.shape == exp.shape and (got == exp).all()


def test_resident_keyframes(pkg, oracle):
    cur, kfs, Fs, eps, sf, sg = _scene(31, 1000, (1000, 800, 1200))
    dcur = pkg.DeviceKeyFrame(cur)
    dk = [pkg.DeviceKeyFrame(k) for k in kfs]
    m = pkg.ORBmatcher(0.75, True)
    rng = np.random.default_rng(5)
    for rnd in range(3):                                             # the flags change between calls, the resident data does not
        for s in [cur] + kfs:
            s["flag"] = (rng.random(len(s["desc"])) < 0.3 + 0.2 * rnd).astype(np.uint8)
        for i, kf in enumerate(kfs):
            g, c = m.SearchByBoWResident(dk[i], kf["flag"], dcur)
            e, ec = oracle.search_by_bow_kf_f(kf, cur, 0.75, True)
            assert c == ec and (g == e).all(), (rnd, i)
        g, c = m.SearchByBoWKeyFramesFrameResident(dk, [k["flag"] for k in kfs], cur)     # Tracking::Relocalization's loop as one call
        for i, kf in enumerate(kfs):
            e, ec = oracle.search_by_bow_kf_f(kf, cur, 0.75, True)
            assert c[i] == ec and (g[i] == e).all(), (rnd, i)
        g, c = m.SearchByBoWKeyFramesResident(dcur, cur["flag"], dk, [k["flag"] for k in kfs])
        for i, kf in enumerate(kfs):
            e, ec = oracle.search_by_bow_kf_kf(cur, kf, 0.75, True)
            assert c[i] == ec and (g[i] == e).all(), (rnd, i)
        for only_stereo in (False, True):
            got = m.SearchForTriangulationResident(dcur, cur["flag"], dk, [k["flag"] for k in kfs], Fs, eps, sf, sg, bOnlyStereo=only_stereo)
            for i, kf in enumerate(kfs):
                exp = oracle.search_for_triangulation(cur, kf, Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, True, only_stereo)
                assert got[i].shape == exp.shape and (got[i] == exp).all(), (rnd, i, only_stereo)
    # no flags at all: no feature has a map point yet
    got = m.SearchForTriangulationResident(dcur, None, dk, None, Fs, eps, sf, sg)
    for i, kf in enumerate(kfs):
        exp = oracle.search_for_triangulation(dict(cur, flag=np.zeros(1000, np.uint8)), dict(kf, flag=np.zeros(len(kf["desc"]), np.uint8)),
                                              Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, True, False)
        assert got[i].shape == exp.shape and (got[i] == exp).all()


def test_two_threads_call_concurrently(pkg, oracle):
    """the tracking and the local-mapping thread of the reference match at the same time: per-thread contexts"""
    cur, kfs, Fs, eps, sf, sg = _scene(41, 900, (900, 950))
    tc = _tri_flags(cur, 5, 2); tk = [_tri_flags(k, 5, 1) for k in kfs]
    exp_b = [oracle.search_by_bow_kf_f(k, cur, 0.7, True) for k in kfs]
    exp_t = [oracle.search_for_triangulation(tc, tk[i], Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, True, False) for i in range(2)]
    errs = []

    def bow():
        m = pkg.ORBmatcher(0.7, True)
        for it in range(60):
            g, c = m.SearchByBoW(kfs[it & 1], cur)
            if c != exp_b[it & 1][1] or not (g == exp_b[it & 1][0]).all():
                errs.append(("bow", it))

    def tri():
        m = pkg.ORBmatcher(0.6, True)
        for it in range(60):
            g = m.SearchForTriangulation(tc, tk[it & 1], Fs[it & 1], eps[it & 1][0], eps[it & 1][1], sf, sg)
            if g.shape != exp_t[it & 1].shape or not (g == exp_t[it & 1]).all():
                errs.append(("tri", it))
    ts = [threading.Thread(target=bow), threading.Thread(target=tri)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs[:5]


def test_capacity_error(pkg, oracle):
    import ctypes as C
    cur, kfs, Fs, eps, sf, sg = _scene(51, 600, (600,))
    tc = _tri_flags(cur, 5, 0); tk = _tri_flags(kfs[0], 5, 0)
    exp = oracle.search_for_triangulation(tc, tk, Fs[0], eps[0][0], eps[0][1], sf, sg, 0.6, False, False)
    assert len(exp) > 8
    a, ka = pkg.orbx.make_featset(tc); b, kb = pkg.orbx.make_featset(tk)
    pairs = np.full((8, 2), -7, np.int32); n = C.c_int()
    F = np.ascontiguousarray(Fs[0].reshape(9))
    rc = pkg.lib().orbx_search_for_triangulation(0, C.byref(a), C.byref(b), F.ctypes.data_as(C.c_void_p), eps[0][0], eps[0][1],
                                                 sf.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p), 8, 0, 0,
                                                 pairs.ctypes.data_as(C.c_void_p), 8, C.byref(n))
    assert rc == -2 and n.value == len(exp) and (pairs == exp[:8]).all()
