"""Pin the CPU oracle against every known answer the reference's own text holds and
against independent brute-force definitions of its primitives.  The reference has no tests and no
golden vectors, and cannot be built here (OpenCV absent): "parity unpinned" beyond these."""
import math

import numpy as np
import pytest

from tools import synth


def test_umax_table(oracle):
    # comment in reference src/ORBextractor.cc:80,526
    assert oracle.Oracle(1000).umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # circular patch of 749 pixels
    u = oracle.Oracle(1000).umax()
    assert (2 * u[0] + 1) + 2 * sum(2 * int(x) + 1 for x in u[1:]) == 749


def test_feature_quotas_4000(oracle):
    # comments in reference src/ORBextractor.cc:471,478-491: nDesired0 = 868.698, 869+724+603+503+419+349+291 = 3758, last 242
    q = oracle.Oracle(4000, 1.2, 8, 20, 7).features_per_level()
    assert q.tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    assert q[:7].sum() == 3758 and q.sum() == 4000


def test_feature_quotas_1000_2000(oracle):
    # SURVEY.md Appendix D (computed with the reference's arithmetic)
    assert oracle.Oracle(1000).features_per_level().tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.Oracle(2000).features_per_level().tolist() == [434, 362, 302, 251, 209, 175, 145, 122]


def test_scale_tables(oracle):
    o = oracle.Oracle(1000, 1.2, 8, 20, 7)
    sf = o.scale_factors()
    s = np.float32(1.0)
    for i in range(8):
        assert sf[i] == s
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))  # float * double -> float (src/ORBextractor.cc:450)
    assert (o.level_sigma2() == sf * sf).all()
    assert (o.inv_scale_factors() == np.float32(1.0) / sf).all()
    assert (o.inv_level_sigma2() == np.float32(1.0) / (sf * sf)).all()


def test_level_dims_appendix_d(oracle):
    dims = {(1241, 376): [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)],
            (640, 480): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
            (752, 480): [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]}
    for (w, h), exp in dims.items():
        o = oracle.Oracle(1000)
        o.extract(np.zeros((h, w), np.uint8))
        assert [o.level(l).shape[::-1] for l in range(8)] == exp
    total = sum(a * b for a, b in dims[(1241, 376)])
    assert total == 1444097  # P_tot of SURVEY.md 8d


def test_thresholds_and_popcount(oracle):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (500, 32), dtype=np.uint8); b = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    for i in range(500):
        assert oracle.hamming(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, f) == 256


def test_cv_round_half_even(oracle):
    L = oracle.lib()
    for v, e in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2)]:
        assert L.oracle_cv_round_f(v) == e


def test_fast_atan2_close_to_atan2(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-100000, 100000, 2))
        a = L.oracle_fast_atan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref); d = min(d, 360 - d)
        assert d < 0.3, (y, x, a, ref)          # OpenCV documents ~0.3 degree accuracy
        assert 0.0 <= a <= 360.0
    assert L.oracle_fast_atan2(0.0, 0.0) == 0.0
    assert L.oracle_fast_atan2(0.0, 5.0) == 0.0
    assert abs(L.oracle_fast_atan2(5.0, 0.0) - 90.0) < 1e-3


def test_sincos_is_correctly_rounded_almost_everywhere(oracle):
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), np.linspace(0, 6.3, 5000)]).astype(np.float32)
    bad = 0
    for x in xs:
        s, c = C.c_float(), C.c_float()
        L.oracle_sincos(float(x), C.byref(s), C.byref(c))
        bad += (np.float32(s.value) != np.float32(math.sin(float(x)))) + (np.float32(c.value) != np.float32(math.cos(float(x))))
    assert bad <= 2  # fp64 evaluation rounded once: equals the correctly rounded fp32 value except in 1e-8-rare ties


def test_gaussian_taps(oracle):
    # taps [18,34,49,55,49,34,18], sum 257 (not renormalised)
    img = np.zeros((21, 21), np.uint8); img[10, 10] = 255
    out = np.zeros_like(img)
    oracle.lib().oracle_gaussian_blur7(img.ctypes.data, 21, 21, 21, out.ctypes.data, 21)
    taps = np.array([18, 34, 49, 55, 49, 34, 18])
    exp = (np.outer(taps, taps) * 255 + (1 << 15)) >> 16
    assert (out[7:14, 7:14] == exp).all() and out.sum() == exp.sum()
    # saturation: 255 * 257 * 257 >> 16 = 257 -> 255
    full = np.full((16, 16), 255, np.uint8); out2 = np.zeros_like(full)
    oracle.lib().oracle_gaussian_blur7(full.ctypes.data, 16, 16, 16, out2.ctypes.data, 16)
    assert (out2 == 255).all()


def test_gaussian_taps_both_opencv_profiles(oracle, pkg):
    """the 7 taps are one table per OpenCV generation, and the product's table is the oracle's.  Profile 0 =
    OpenCV <= 3.4.1 (cvRound(k * 256), sum 257), profile 1 = OpenCV >= 3.4.2 (error-diffused, sums to exactly 256)."""
    t0, t1 = np.array([18, 34, 49, 55, 49, 34, 18]), np.array([18, 34, 48, 56, 48, 34, 18])
    assert (oracle.gaussian_taps(0) == t0).all() and (oracle.gaussian_taps(1) == t1).all()
    assert (pkg.orbx.gaussian_taps(pkg.orbx.CV_PROFILE_3_2) == t0).all() and (pkg.orbx.gaussian_taps(pkg.orbx.CV_PROFILE_3_4_2) == t1).all()
    assert t1.sum() == 256
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (24, 30), dtype=np.uint8)
    outs = []
    for prof, taps in ((0, t0), (1, t1)):
        out = np.zeros_like(img)
        oracle.lib().oracle_gaussian_blur7_profile(img.ctypes.data, 30, 24, 30, out.ctypes.data, 30, prof)
        pad = np.pad(img.astype(np.int64), 3, mode="reflect")
        rows = sum(int(taps[k]) * pad[:, k:k + 30] for k in range(7))
        full = sum(int(taps[k]) * rows[k:k + 24, :] for k in range(7))
        assert (out == np.minimum((full + (1 << 15)) >> 16, 255)).all()
        outs.append(out)
    assert (outs[0] != outs[1]).any()      # the two generations do blur differently
    flat = np.full((16, 16), 200, np.uint8); o1 = np.zeros_like(flat)
    oracle.lib().oracle_gaussian_blur7_profile(flat.ctypes.data, 16, 16, 16, o1.ctypes.data, 16, 1)
    assert (o1 == 200).all()               # a kernel that sums to 256 keeps a flat image (the 257 one brightens it: 200 -> 201)


def test_blur_reflect101(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 30), dtype=np.uint8)
    out = np.zeros_like(img)
    oracle.lib().oracle_gaussian_blur7(img.ctypes.data, 30, 24, 30, out.ctypes.data, 30)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    taps = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    rows = sum(taps[k] * pad[:, k:k + 30] for k in range(7))
    full = sum(taps[k] * rows[k:k + 24, :] for k in range(7))
    exp = np.minimum((full + (1 << 15)) >> 16, 255)
    assert (out == exp).all()


def test_resize_constant_and_bounds(oracle):
    L = oracle.lib()
    for v in (0, 1, 127, 255):
        src = np.full((100, 120), v, np.uint8); dst = np.zeros((83, 100), np.uint8)
        L.oracle_resize_linear(src.ctypes.data, 120, 100, 120, dst.ctypes.data, 100, 83, 100)
        assert (dst == v).all()
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (60, 72), dtype=np.uint8); dst = np.zeros((50, 60), np.uint8)
    L.oracle_resize_linear(src.ctypes.data, 72, 60, 72, dst.ctypes.data, 60, 50, 60)
    # independent float bilinear (pixel-centre aligned) agrees within the fixed-point error
    xs = (np.arange(60) + 0.5) * (72 / 60) - 0.5; ys = (np.arange(50) + 0.5) * (60 / 50) - 0.5
    x0 = np.clip(np.floor(xs).astype(int), 0, 70); y0 = np.clip(np.floor(ys).astype(int), 0, 58)
    fx = np.clip(xs - x0, 0, 1); fy = np.clip(ys - y0, 0, 1)
    s = src.astype(np.float64)
    ref = ((1 - fy)[:, None] * ((1 - fx) * s[y0][:, x0] + fx * s[y0][:, x0 + 1]) +
           fy[:, None] * ((1 - fx) * s[y0 + 1][:, x0] + fx * s[y0 + 1][:, x0 + 1]))
    assert np.abs(dst.astype(np.float64) - ref).max() <= 1.0


def test_resize_exact_half_is_area_average(oracle):
    """both scale factors exactly 2: cv::resize turns INTER_LINEAR into INTER_AREA = the rounded mean of each 2x2 block"""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (60, 72), dtype=np.uint8); dst = np.zeros((30, 36), np.uint8)
    L.oracle_resize_linear(src.ctypes.data, 72, 60, 72, dst.ctypes.data, 36, 30, 36)
    s = src.astype(np.int32)
    assert (dst == (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).all()
    # one dimension off by a pixel: back on the bilinear path (differs from the block mean somewhere)
    dst2 = np.zeros((30, 35), np.uint8)
    L.oracle_resize_linear(src.ctypes.data, 72, 60, 72, dst2.ctypes.data, 35, 30, 35)
    assert (dst2 != dst[:, :35]).any()


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _fast_bruteforce(patch, t):
    """FAST-9/16 by definition: corner iff 9 contiguous ring pixels all darker than v-t or all brighter
    than v+t; score = largest threshold for which it is still a corner (== OpenCV cornerScore)."""
    v = int(patch[3, 3])
    ring = [int(patch[3 + dy, 3 + dx]) for dx, dy in RING]

    def is_corner(th):
        for sign in (1, -1):
            flags = [(sign * (v - r)) > th for r in ring]
            run = 0
            for f in flags + flags[:8]:
                run = run + 1 if f else 0
                if run >= 9:
                    return True
        return False
    if not is_corner(t):
        return 0
    s = t
    while s < 255 and is_corner(s + 1):
        s += 1
    return s


def test_fast_score_matches_definition(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    hits = 0
    for trial in range(3000):
        base = rng.integers(0, 256)
        patch = np.clip(base + rng.integers(-60, 60, (7, 7)), 0, 255).astype(np.uint8)
        if trial % 3 == 0:   # plant an arc
            k0, ln = rng.integers(0, 16), rng.integers(7, 13)
            delta = int(rng.integers(8, 120)) * (1 if rng.random() < 0.5 else -1)
            for k in range(ln):
                dx, dy = RING[(k0 + k) % 16]
                patch[3 + dy, 3 + dx] = np.clip(int(patch[3, 3]) + delta + rng.integers(0, 30) * np.sign(delta), 0, 255)
        patch = np.ascontiguousarray(patch)
        for t in (7, 20):
            got = L.oracle_fast_score(patch.ctypes.data + 3 * 7 + 3, 7, t)
            exp = _fast_bruteforce(patch, t)
            assert got == exp, (trial, t, got, exp)
            hits += exp > 0
    assert hits > 200


def test_candidates_order_and_nms(oracle):
    """candidates come cell-row-major then row-major inside a cell; each is a strict 3x3 maximum of its cell"""
    img = synth.image(6, 400, 300)
    o = oracle.Oracle(500)
    o.extract(img)
    x, y, r = o.candidates(0)
    assert len(x) > 200 and r.min() >= 7
    width, height = 400 - 32, 300 - 32
    n_cols, n_rows = int(width / 30), int(height / 30)
    w_cell, h_cell = math.ceil(width / n_cols), math.ceil(height / n_rows)
    cell = ((y - 3) // h_cell) * n_cols + ((x - 3) // w_cell)
    key = cell.astype(np.int64) * 10 ** 8 + y.astype(np.int64) * 10 ** 4 + x
    assert (np.diff(key) > 0).all()
    # coordinates are relative to (16,16) and lie in the FAST-detectable area
    assert x.min() >= 3 and y.min() >= 3 and x.max() <= width - 4 and y.max() <= height - 4


def test_quadtree_model_matches_sequential_oracle(oracle):
    """the data-parallel formulation used by the HIP kernel == the sequential list algorithm"""
    from quadtree_model import quadtree_model
    L = oracle.lib()
    rng = np.random.default_rng(7)
    for trial in range(120):
        W = int(rng.integers(40, 1300)); H = int(rng.integers(40, 500))
        if round(W / H) < 1:
            continue
        n = int(rng.integers(1, 4000)); N = int(rng.integers(0, 700))
        if trial % 2:
            xs = rng.integers(0, W - 3, n); ys = rng.integers(0, H - 3, n)
        else:
            cx = rng.integers(0, W - 3, 6); cy = rng.integers(0, H - 3, 6); k = rng.integers(0, 6, n)
            xs = np.clip(cx[k] + rng.normal(0, W / 25, n).astype(int), 0, W - 4)
            ys = np.clip(cy[k] + rng.normal(0, H / 25, n).astype(int), 0, H - 4)
        p = np.unique(np.stack([ys, xs], 1), axis=0)
        ys, xs = np.ascontiguousarray(p[:, 0], np.int32), np.ascontiguousarray(p[:, 1], np.int32)
        r = np.ascontiguousarray(rng.integers(7, 40 if trial % 3 else 255, len(xs)), np.int32)
        out = np.zeros(len(xs) + 1, np.int32)
        cnt = L.oracle_distribute_octtree(xs.ctypes.data, ys.ctypes.data, r.ctypes.data, len(xs), 16, 16 + W, 16, 16 + H, N, out.ctypes.data, len(out))
        model = quadtree_model(xs, ys, r, 16, 16 + W, 16, 16 + H, N)
        assert cnt == len(model) and (out[:cnt] == model).all(), trial
        assert cnt <= max(N + 2, 4 * max(round(W / H), 1)) or cnt == len(xs)   # SURVEY.md A.4 output bound


def test_extract_output_invariants(oracle):
    img = synth.image(8, 640, 480)
    o = oracle.Oracle(1000)
    k, d = o.extract(img)
    assert 900 <= len(k) <= 1000 + 3 * 8
    assert (np.diff(k["octave"]) >= 0).all()                     # levels concatenated 0..7
    sf = o.scale_factors()
    assert (k["size"] == np.floor(31 * sf[k["octave"]])).all()
    assert ((k["angle"] >= 0) & (k["angle"] < 360.0001)).all() and (k["class_id"] == -1).all()
    lv_x = k["x"] / sf[k["octave"]]; lv_y = k["y"] / sf[k["octave"]]
    for l in range(8):
        m = k["octave"] == l
        h, w = o.level(l).shape
        assert lv_x[m].min() >= 18.99 and lv_x[m].max() <= w - 19 and lv_y[m].min() >= 18.99 and lv_y[m].max() <= h - 19
    assert d.shape == (len(k), 32) and 60 < np.unpackbits(d, axis=1).sum(axis=1).mean() < 196


def test_three_maxima(oracle):
    import ctypes as C
    L = oracle.lib()

    def tm(h):
        a = np.ascontiguousarray(h, np.int32); i1, i2, i3 = C.c_int(), C.c_int(), C.c_int()
        L.oracle_three_maxima(a.ctypes.data, len(a), C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value
    h = np.zeros(30, int); h[3] = 100; h[7] = 50; h[9] = 20
    assert tm(h) == (3, 7, 9)
    h[7] = 9; h[9] = 5          # second < 10 % of first -> both dropped
    assert tm(h) == (3, -1, -1)
    h[7] = 50; h[9] = 9         # third < 10 %
    assert tm(h) == (3, 7, -1)
    assert tm(np.zeros(30, int)) == (-1, -1, -1)
