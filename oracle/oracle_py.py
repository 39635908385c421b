"""ctypes wrapper over oracle/liborb_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never from the product package.  PARITY UNPINNED
(see oracle/orb_oracle.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "orb_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


class FrameFeats(C.Structure):
    _fields_ = [("n", C.c_int), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p),
                ("u_right", C.c_void_p), ("desc", C.c_void_p), ("occupied", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class ProjPoints(C.Structure):
    _fields_ = [("n", C.c_int), ("u", C.c_void_p), ("v", C.c_void_p), ("aux", C.c_void_p), ("level", C.c_void_p),
                ("angle", C.c_void_p), ("view_cos", C.c_void_p), ("desc", C.c_void_p), ("valid", C.c_void_p), ("has_obs", C.c_void_p)]


def make_frame_feats(d, cls=None):
    """dict(x,y,octave,angle,u_right,desc,occupied,bounds=(minx,miny,maxx,maxy)) -> (struct, keepalive)"""
    keep = {}
    def arr(k, dt):
        keep[k] = np.ascontiguousarray(d[k], dtype=dt); return keep[k].ctypes.data
    s = (cls or FrameFeats)()
    s.x = arr("x", np.float32); s.y = arr("y", np.float32); s.octave = arr("octave", np.int32); s.angle = arr("angle", np.float32)
    s.u_right = arr("u_right", np.float32); s.desc = arr("desc", np.uint8); s.occupied = arr("occupied", np.uint8)
    s.n = len(keep["x"])
    s.min_x, s.min_y, s.max_x, s.max_y = [float(v) for v in d["bounds"]]
    return s, keep


def make_proj_points(d, cls=None):
    keep = {}
    def arr(k, dt):
        keep[k] = np.ascontiguousarray(d[k], dtype=dt); return keep[k].ctypes.data
    s = (cls or ProjPoints)()
    s.u = arr("u", np.float32); s.v = arr("v", np.float32); s.aux = arr("aux", np.float32); s.level = arr("level", np.int32)
    s.angle = arr("angle", np.float32); s.view_cos = arr("view_cos", np.float32); s.desc = arr("desc", np.uint8)
    s.valid = arr("valid", np.uint8); s.has_obs = arr("has_obs", np.uint8)
    s.n = len(keep["u"])
    return s, keep


class FeatSet(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("nnodes", C.c_int), ("node_id", C.c_void_p),
                ("node_off", C.c_void_p), ("feat", C.c_void_p), ("flag", C.c_void_p), ("angle", C.c_void_p),
                ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p), ("u_right", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        for f in ("oracle_scale_factors", "oracle_inv_scale_factors", "oracle_level_sigma2", "oracle_inv_level_sigma2"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("oracle_features_per_level", "oracle_umax"):
            getattr(L, f).restype = C.POINTER(C.c_int)
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_extract.restype = C.c_int
        L.oracle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_level_pixels.restype = C.c_void_p
        L.oracle_level_pixels.argtypes = [C.c_void_p, C.c_int]
        L.oracle_level_blurred.restype = C.c_void_p
        L.oracle_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.oracle_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_level_nkeypoints.argtypes = [C.c_void_p, C.c_int]
        L.oracle_cv_round_f.argtypes = [C.c_float]
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_fast_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.oracle_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_gaussian_blur7_profile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_gaussian_taps.argtypes = [C.c_int, C.c_void_p]
        L.oracle_set_cv_profile.argtypes = [C.c_void_p, C.c_int]
        L.oracle_distribute_octtree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_search_by_bow_kf_f.argtypes = [C.POINTER(FeatSet), C.POINTER(FeatSet), C.c_float, C.c_int, C.c_void_p]
        L.oracle_search_by_bow_kf_kf.argtypes = [C.POINTER(FeatSet), C.POINTER(FeatSet), C.c_float, C.c_int, C.c_void_p]
        L.oracle_search_for_triangulation.argtypes = [C.POINTER(FeatSet), C.POINTER(FeatSet), C.c_void_p, C.c_float, C.c_float,
                                                      C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_vocab_create.restype = C.c_void_p
        L.oracle_vocab_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_vocab_load_text.restype = C.c_void_p
        L.oracle_vocab_load_text.argtypes = [C.c_char_p]
        L.oracle_vocab_destroy.argtypes = [C.c_void_p]
        L.oracle_vocab_nodes.argtypes = [C.c_void_p]
        L.oracle_vocab_words.argtypes = [C.c_void_p]
        L.oracle_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + [C.POINTER(C.c_int)]
        L.oracle_bow_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
        L.oracle_search_by_projection_last.argtypes = [C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_search_by_projection_points.argtypes = [C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_search_by_projection_keyframe.argtypes = [C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_search_by_projection_sim3.argtypes = [C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_window_best.argtypes = [C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_search_for_initialization.argtypes = [C.POINTER(FrameFeats), C.POINTER(FrameFeats), C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_search_by_sim3.argtypes = [C.POINTER(FrameFeats), C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.POINTER(ProjPoints),
                                            C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.oracle_cvt_gray.restype = None
        L.oracle_three_maxima.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One ORBextractor instance of the CPU oracle (keeps its pyramid after extract)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("oracle_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def set_cv_profile(self, profile):
        """0 = GaussianBlur taps summing to 257 (cvRound per tap; OpenCV 3.2, default), 1 = taps summing to 256 (error-diffused; later OpenCV releases: which one switched is parity unpinned)"""
        assert self.L.oracle_set_cv_profile(self.h, profile) == 0

    def _farr(self, fn, n=None):
        return np.ctypeslib.as_array(fn(self.h), shape=(n or self.nlevels,)).copy()

    def scale_factors(self): return self._farr(self.L.oracle_scale_factors)
    def inv_scale_factors(self): return self._farr(self.L.oracle_inv_scale_factors)
    def level_sigma2(self): return self._farr(self.L.oracle_level_sigma2)
    def inv_level_sigma2(self): return self._farr(self.L.oracle_inv_level_sigma2)
    def features_per_level(self): return self._farr(self.L.oracle_features_per_level)
    def umax(self): return self._farr(self.L.oracle_umax, 16)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 4 * self.nlevels + 64
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = self.L.oracle_extract(self.h, _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError(f"oracle_extract failed: {n}")
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l, blurred=False):
        w, h = C.c_int(), C.c_int()
        assert self.L.oracle_level_dims(self.h, l, C.byref(w), C.byref(h)) == 0
        ptr = (self.L.oracle_level_blurred if blurred else self.L.oracle_level_pixels)(self.h, l)
        if not ptr:
            return None
        buf = (C.c_uint8 * (w.value * h.value)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, w.value).copy()

    def candidates(self, l):
        n = self.L.oracle_level_candidates(self.h, l, None, None, None, 0)
        x = np.zeros(n, np.int32); y = np.zeros(n, np.int32); r = np.zeros(n, np.int32)
        self.L.oracle_level_candidates(self.h, l, _p(x), _p(y), _p(r), n)
        return x, y, r

    def nkeypoints(self, l):
        return self.L.oracle_level_nkeypoints(self.h, l)


def stereo_match(oL, oR, kL, dL, kR, dR, bf, min_z):
    kL = np.ascontiguousarray(kL); kR = np.ascontiguousarray(kR)
    dL = np.ascontiguousarray(dL); dR = np.ascontiguousarray(dR)
    ur = np.zeros(len(kL), np.float32); dp = np.zeros(len(kL), np.float32)
    rc = lib().oracle_stereo_match(oL.h, oR.h, _p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), bf, min_z, _p(ur), _p(dp))
    assert rc == 0
    return ur, dp


def gaussian_taps(profile):
    t = np.zeros(7, np.int32)
    lib().oracle_gaussian_taps(profile, t.ctypes.data)
    return t


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_hamming(_p(a), _p(b))


def make_featset(fs):
    """fs: dict with desc [n,32] u8, node_id u32, node_off i32, feat u32, flag u8, angle f32,
    optional x,y f32, octave i32, u_right f32.  Returns (FeatSet, keepalive)."""
    keep = {}
    def arr(k, dt):
        if fs.get(k) is None:
            return None
        keep[k] = np.ascontiguousarray(fs[k], dtype=dt)
        return keep[k].ctypes.data
    s = FeatSet()
    s.desc = arr("desc", np.uint8)
    s.n = len(keep["desc"])
    s.node_id = arr("node_id", np.uint32); s.node_off = arr("node_off", np.int32); s.feat = arr("feat", np.uint32)
    s.nnodes = len(keep["node_id"])
    s.flag = arr("flag", np.uint8); s.angle = arr("angle", np.float32)
    s.x = arr("x", np.float32); s.y = arr("y", np.float32)
    s.octave = arr("octave", np.int32); s.u_right = arr("u_right", np.float32)
    return s, keep


def search_by_bow_kf_f(kf, f, nnratio, check_ori):
    a, ka = make_featset(kf); b, kb = make_featset(f)
    out = np.full(b.n, -1, np.int32)
    n = lib().oracle_search_by_bow_kf_f(C.byref(a), C.byref(b), nnratio, int(check_ori), _p(out))
    return out, n


def search_by_bow_kf_kf(k1, k2, nnratio, check_ori):
    a, ka = make_featset(k1); b, kb = make_featset(k2)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_bow_kf_kf(C.byref(a), C.byref(b), nnratio, int(check_ori), _p(out))
    return out, n


def search_for_triangulation(k1, k2, F12, ex, ey, sf2, sig2, nnratio, check_ori, only_stereo):
    a, ka = make_featset(k1); b, kb = make_featset(k2)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf2 = np.ascontiguousarray(sf2, np.float32); sig2 = np.ascontiguousarray(sig2, np.float32)
    pairs = np.zeros((a.n, 2), np.int32)
    n = lib().oracle_search_for_triangulation(C.byref(a), C.byref(b), _p(F), ex, ey, _p(sf2), _p(sig2),
                                              nnratio, int(check_ori), int(only_stereo), _p(pairs), a.n)
    return pairs[:n].copy()


class Vocabulary:
    """DBoW2 vocabulary tree of the CPU oracle (TF_IDF weights, L1 norm)"""

    def __init__(self, k=None, L=None, parent=None, is_leaf=None, desc=None, weight=None, path=None):
        self.lib = lib()
        if path is not None:
            self.h = self.lib.oracle_vocab_load_text(path.encode())
        else:
            parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
            self.h = self.lib.oracle_vocab_create(k, L, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight))
        if not self.h:
            raise ValueError("vocabulary rejected")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_vocab_destroy(self.h); self.h = None

    def nodes(self): return self.lib.oracle_vocab_nodes(self.h)
    def words(self): return self.lib.oracle_vocab_words(self.h)

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8); n = len(desc)
        wid = np.zeros(n, np.uint32); ww = np.zeros(n, np.float64); nid = np.zeros(n, np.uint32)
        bid = np.zeros(n, np.uint32); bval = np.zeros(n, np.float64); nb = C.c_int()
        fid = np.zeros(n, np.uint32); foff = np.zeros(n + 1, np.int32); ffeat = np.zeros(n, np.uint32); fn = C.c_int()
        rc = self.lib.oracle_bow_transform(self.h, _p(desc), n, levelsup, _p(wid), _p(ww), _p(nid), _p(bid), _p(bval), C.byref(nb),
                                           _p(fid), _p(foff), _p(ffeat), C.byref(fn))
        assert rc == 0
        return dict(word_id=wid, word_weight=ww, node_id=nid, bow_id=bid[:nb.value].copy(), bow_val=bval[:nb.value].copy(),
                    fv_node_id=fid[:fn.value].copy(), fv_node_off=foff[:fn.value + 1].copy(), fv_feat=ffeat[:foff[fn.value]].copy())


def bow_accumulate(word_id, word_weight, node_id):
    """the accumulation half of TemplatedVocabulary::transform (BowVector::addWeight / normalize(L1), FeatureVector::addFeature)"""
    wid = np.ascontiguousarray(word_id, np.uint32); ww = np.ascontiguousarray(word_weight, np.float64)
    nid = np.ascontiguousarray(node_id, np.uint32); n = len(wid)
    bid = np.zeros(max(n, 1), np.uint32); bval = np.zeros(max(n, 1), np.float64); nb = C.c_int()
    fid = np.zeros(max(n, 1), np.uint32); foff = np.zeros(n + 1, np.int32); ffeat = np.zeros(max(n, 1), np.uint32); fn = C.c_int()
    rc = lib().oracle_bow_accumulate(_p(wid), _p(ww), _p(nid), n, _p(bid), _p(bval), C.byref(nb), _p(fid), _p(foff), _p(ffeat), C.byref(fn))
    assert rc == 0
    return dict(bow_id=bid[:nb.value].copy(), bow_val=bval[:nb.value].copy(), fv_node_id=fid[:fn.value].copy(),
                fv_node_off=foff[:fn.value + 1].copy(), fv_feat=ffeat[:foff[fn.value]].copy())


def distinctive_descriptor(desc):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    return lib().oracle_distinctive_descriptor(_p(desc), len(desc))


def search_by_projection_last(cur, pts, scale_factors, th, direction, mbf, check_ori):
    a, ka = make_frame_feats(cur); b, kb = make_proj_points(pts)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_projection_last(C.byref(a), C.byref(b), _p(sf), th, direction, mbf, int(check_ori), _p(out))
    return out, n


def search_by_projection_points(cur, pts, scale_factors, th, nnratio):
    a, ka = make_frame_feats(cur); b, kb = make_proj_points(pts)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_projection_points(C.byref(a), C.byref(b), _p(sf), th, nnratio, _p(out))
    return out, n


def search_by_projection_keyframe(cur, pts, scale_factors, th, orb_dist, check_ori):
    a, ka = make_frame_feats(cur); b, kb = make_proj_points(pts)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_projection_keyframe(C.byref(a), C.byref(b), _p(sf), th, int(orb_dist), int(check_ori), _p(out))
    return out, n


def search_by_projection_sim3(kf, pts, scale_factors, th):
    a, ka = make_frame_feats(kf); b, kb = make_proj_points(pts)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_projection_sim3(C.byref(a), C.byref(b), _p(sf), th, _p(out))
    return out, n


def window_best(kf, pts, scale_factors, inv_sigma2, th, chi2, max_dist):
    a, ka = make_frame_feats(kf); b, kb = make_proj_points(pts)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    sg = np.ascontiguousarray(inv_sigma2 if inv_sigma2 is not None else np.zeros(len(sf)), np.float32)
    bi = np.full(b.n, -1, np.int32); bd = np.full(b.n, 256, np.int32)
    n = lib().oracle_window_best(C.byref(a), C.byref(b), _p(sf), _p(sg), th, int(chi2), int(max_dist), _p(bi), _p(bd))
    return bi, bd, n


def search_by_sim3(kf1, kf2, pts12, pts21, sf1, sf2, th):
    a, ka = make_frame_feats(kf1); b, kb = make_frame_feats(kf2); p, kp = make_proj_points(pts12); q, kq = make_proj_points(pts21)
    s1 = np.ascontiguousarray(sf1, np.float32); s2 = np.ascontiguousarray(sf2, np.float32)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_by_sim3(C.byref(a), C.byref(b), C.byref(p), C.byref(q), _p(s1), _p(s2), th, _p(out))
    return out, n


def search_for_initialization(f1, f2, prev_xy, window_size, nnratio, check_ori):
    a, ka = make_frame_feats(f1); b, kb = make_frame_feats(f2)
    xy = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    out = np.full(a.n, -1, np.int32)
    n = lib().oracle_search_for_initialization(C.byref(a), C.byref(b), _p(xy), int(window_size), nnratio, int(check_ori), _p(out))
    return out, n


def remap_bilinear(img, map_x, map_y):
    img = np.ascontiguousarray(img, np.uint8); sh, sw = img.shape
    mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32); dh, dw = mx.shape
    out = np.zeros((dh, dw), np.uint8)
    L = lib()
    L.oracle_remap_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    L.oracle_remap_bilinear.restype = None
    L.oracle_remap_bilinear(_p(img), sw, sh, sw, _p(mx), _p(my), _p(out), dw, dh, dw)
    return out


def undistort_points(xy, fx, fy, cx, cy, dist):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(xy)
    L = lib()
    L.oracle_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_undistort_points.restype = None
    L.oracle_undistort_points(_p(xy), len(xy), fx, fy, cx, cy, _p(d), len(d), _p(out))
    return out


def cvt_gray(img, rgb_order):
    img = np.ascontiguousarray(img, np.uint8); h, w, ch = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().oracle_cvt_gray(_p(img), w, h, img.strides[0], ch, int(rgb_order), _p(out), out.strides[0])
    return out
