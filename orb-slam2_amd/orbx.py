"""ctypes binding of liborbx.so + mirror classes named after the reference's C++ interface."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORBX_SO") or os.path.join(_HERE, "liborbx.so")  # ORBX_SO: diagnostic builds only

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])  # == cv::KeyPoint, 28 B
STAGES = ("resize", "fast", "tree", "desc", "stereo", "stereo_cut")


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"orbx error {code}: {msg}")
        self.code = code


class FeatSet(C.Structure):
    """orbx_featset (include/orbx.h)"""
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("nnodes", C.c_int), ("node_id", C.c_void_p),
                ("node_off", C.c_void_p), ("feat", C.c_void_p), ("flag", C.c_void_p), ("angle", C.c_void_p),
                ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p), ("u_right", C.c_void_p)]


class FrameFeats(C.Structure):
    """orbx_frame_feats (include/orbx.h)"""
    _fields_ = [("n", C.c_int), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p),
                ("u_right", C.c_void_p), ("desc", C.c_void_p), ("occupied", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class ProjPoints(C.Structure):
    """orbx_proj_points (include/orbx.h)"""
    _fields_ = [("n", C.c_int), ("u", C.c_void_p), ("v", C.c_void_p), ("aux", C.c_void_p), ("level", C.c_void_p),
                ("angle", C.c_void_p), ("view_cos", C.c_void_p), ("desc", C.c_void_p), ("valid", C.c_void_p), ("has_obs", C.c_void_p)]


_lib = None


def lib_path():
    return _SO


def lib():
    """Load liborbx.so; fail loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise OrbxError(-4, f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(_SO)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_device_count.restype = i
    L.orbx_device_identity.argtypes = [i, C.c_char_p, i]
    L.orbx_extractor_create.argtypes = [C.POINTER(vp), i, f, i, i, i, i, i, i, i]
    L.orbx_extractor_set_cv_profile.argtypes = [vp, i]
    L.orbx_extractor_set_pyramid_group_limit.argtypes = [vp, i]
    L.orbx_gaussian_taps.argtypes = [i, vp]
    L.orbx_extractor_destroy.argtypes = [vp]
    L.orbx_extractor_destroy.restype = None
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f
    L.orbx_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_get_features_per_level.argtypes = [vp, vp]
    L.orbx_max_keypoints.argtypes = [vp, i, i]
    L.orbx_extract.argtypes = [vp, vp, i, i, sz, vp, vp, i, ip]
    L.orbx_extract_stereo.argtypes = [vp, vp, vp, i, i, sz, f, f, vp, vp, i, vp, vp, vp]
    L.orbx_extract_stereo_submit.argtypes = [vp, vp, vp, i, i, sz, f, f, ip]
    L.orbx_extract_stereo_wait.argtypes = [vp, i, vp, vp, i, vp, vp, vp]
    L.orbx_extract_submit.argtypes = [vp, vp, i, i, sz, ip]
    L.orbx_pipeline_warm.argtypes = [vp, i, i]
    L.orbx_extract_wait.argtypes = [vp, i, vp, vp, i, vp]
    L.orbx_pinned_alloc.argtypes = [sz]; L.orbx_pinned_alloc.restype = vp
    L.orbx_pinned_free.argtypes = [vp]; L.orbx_pinned_free.restype = None
    L.orbx_extract_color.argtypes = [vp, vp, i, i, sz, i, i, vp, vp, i, ip, vp, sz]
    L.orbx_extract_batch.argtypes = [vp, vp, i, i, i, sz, vp, vp, i, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, sz, sz, i, i, i, vp, vp, i, vp, vp]
    L.orbx_sync.argtypes = [vp, vp]
    L.orbx_pyramid_level.argtypes = [vp, i, i, vp, sz, ip, ip]
    L.orbx_stereo_match.argtypes = [vp, vp, vp, vp, i, vp, vp, i, f, f, vp, vp]
    L.orbx_stereo_match_batch_device.argtypes = [vp, i, vp, i, i, vp, vp, vp, vp, vp, vp, i, f, f, vp, vp, i, vp]
    L.orbx_stereo_row_table_available.argtypes = [vp, vp, i, i, i]
    L.orbx_hamming.argtypes = [vp, vp]
    FS = C.POINTER(FeatSet)
    L.orbx_search_by_bow_kf_f.argtypes = [i, FS, FS, f, i, vp, ip]
    L.orbx_search_by_bow_kf_f_batch.argtypes = [i, FS, i, FS, f, i, vp, vp]
    L.orbx_search_by_bow_kf_kf.argtypes = [i, FS, FS, f, i, vp, ip]
    L.orbx_search_for_triangulation.argtypes = [i, FS, FS, vp, f, f, vp, vp, i, i, i, vp, i, ip]
    L.orbx_search_by_bow_kf_kf_batch.argtypes = [i, FS, FS, i, f, i, vp, vp]
    L.orbx_search_for_triangulation_batch.argtypes = [i, FS, FS, i, vp, vp, vp, vp, i, i, i, vp, i, vp]
    L.orbx_kf_create.argtypes = [i, FS, C.POINTER(vp)]
    L.orbx_kf_destroy.argtypes = [vp]; L.orbx_kf_destroy.restype = None
    L.orbx_kf_size.argtypes = [vp]
    L.orbx_kf_search_by_bow_kf_f.argtypes = [vp, vp, vp, f, i, vp, ip]
    L.orbx_kf_search_by_bow_kf_kf.argtypes = [vp, vp, vp, vp, i, f, i, vp, vp]
    L.orbx_kf_search_by_bow_kfs_f.argtypes = [vp, vp, i, FS, f, i, vp, vp]
    L.orbx_kf_search_for_triangulation.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp, i, i, i, vp, i, vp]
    L.orbx_bowdb_create.argtypes = [i, FS, i, C.POINTER(vp)]
    L.orbx_bowdb_search.argtypes = [vp, FS, f, i, vp, vp]
    L.orbx_bowdb_size.argtypes = [vp]
    L.orbx_bowdb_destroy.argtypes = [vp]
    L.orbx_bowdb_destroy.restype = None
    L.orbx_vocab_create.argtypes = [i, i, i, i, vp, vp, vp, vp, C.POINTER(vp)]
    L.orbx_vocab_load_text.argtypes = [i, C.c_char_p, C.POINTER(vp)]
    L.orbx_vocab_info.argtypes = [vp, ip, ip, ip, ip]
    L.orbx_vocab_destroy.argtypes = [vp]
    L.orbx_vocab_destroy.restype = None
    L.orbx_bow_transform.argtypes = [vp, vp, i, i, vp, vp, vp, vp, vp, ip, vp, vp, vp, ip]
    L.orbx_search_by_projection_last_frame.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(ProjPoints), vp, i, f, i, f, i, vp, ip]
    L.orbx_search_by_projection_map_points.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(ProjPoints), vp, i, f, f, vp, ip]
    L.orbx_search_by_projection_keyframe.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(ProjPoints), vp, i, f, i, i, vp, ip]
    L.orbx_search_by_projection_sim3.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(ProjPoints), vp, i, f, vp, ip]
    L.orbx_window_best.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(ProjPoints), vp, vp, i, f, i, i, vp, vp, ip]
    L.orbx_bow_frames_create.argtypes = [i, i, i, C.POINTER(vp)]
    L.orbx_bow_frames_destroy.argtypes = [vp]; L.orbx_bow_frames_destroy.restype = None
    L.orbx_bow_transform_batch_device.argtypes = [vp, vp, vp, vp, vp, i, i, vp]
    L.orbx_bow_frames_read.argtypes = [vp, i, vp, vp, vp, ip, vp, vp, vp, ip]
    L.orbx_bowdb_search_batch_device.argtypes = [vp, vp, i, f, i, vp, vp, vp]
    L.orbx_bowdb_search_batch_device_compact.argtypes = [vp, vp, i, f, i, vp, i, vp, vp]
    L.orbx_undistort_keypoints.argtypes = [i, vp, i, f, f, f, f, vp, i, vp]
    L.orbx_rectifier_create.argtypes = [i, i, i, i, i, vp, vp, C.POINTER(vp)]
    L.orbx_rectifier_destroy.argtypes = [vp]; L.orbx_rectifier_destroy.restype = None
    L.orbx_rectifier_size.argtypes = [vp, ip, ip]
    L.orbx_remap_batch_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i, vp, C.c_size_t, C.c_size_t, vp]
    L.orbx_extract_rectified.argtypes = [vp, vp, vp, i, i, C.c_size_t, vp, vp, i, ip, vp, C.c_size_t]
    L.orbx_search_for_initialization.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(FrameFeats), vp, i, f, i, vp, ip]
    L.orbx_search_by_sim3.argtypes = [i, C.POINTER(FrameFeats), C.POINTER(FrameFeats), C.POINTER(ProjPoints), C.POINTER(ProjPoints),
                                      vp, vp, i, f, vp, ip]
    L.orbx_distinctive_descriptors.argtypes = [i, vp, vp, i, vp]
    L.orbx_profile_enable.argtypes = [vp, i]
    L.orbx_profile_stages.argtypes = [vp, C.c_uint]
    L.orbx_profile_read.argtypes = [vp, vp, vp, i]
    L.orbx_debug_candidates.argtypes = [vp, i, i, vp, vp, vp, i, ip]
    L.orbx_debug_level_counts.argtypes = [vp, i, vp]
    L.orbx_debug_fast_form.argtypes = [vp]
    L.orbx_debug_set_bow_form.argtypes = [i]
    L.orbx_debug_set_match_items.argtypes = [i]
    L.orbx_debug_match_timing.argtypes = [vp]
    _lib = L
    return L


def device_identity(device):
    """"<pci bus id> <uuid hex> <gcn arch>" of a visible device (orbx_device_identity)"""
    buf = C.create_string_buffer(160)
    _check(lib().orbx_device_identity(int(device), buf, len(buf)))
    return buf.value.decode(errors="replace")


def pipeline_depth():
    return lib().orbx_pipeline_depth()


class _Pinned:
    def __init__(self, nbytes):
        self.ptr = lib().orbx_pinned_alloc(nbytes)
        if not self.ptr:
            raise OrbxError(-5, lib().orbx_last_error().decode(errors="replace"))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().orbx_pinned_free(self.ptr); self.ptr = None


def pinned_array(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (orbx_pinned_alloc): uploads from it need no staging copy"""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    owner = _Pinned(nbytes)
    buf = (C.c_uint8 * nbytes).from_address(owner.ptr)
    buf._orbx_owner = owner              # the ctypes buffer is the base object of every view: the pages are freed with the last of them
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


CV_TAPS_257, CV_TAPS_256 = 0, 1                 # named by what they are: the 7 taps sum to 257 (cvRound(k * 256)) or to 256 (error-diffused)
CV_PROFILE_3_2, CV_PROFILE_3_4_2 = 0, 1         # older names; which OpenCV release has which table is parity unpinned (include/orbx.h)


def gaussian_taps(profile):
    t = np.zeros(7, np.int32)
    _check(lib().orbx_gaussian_taps(profile, _p(t)))
    return t


def debug_set_bow_form(form):
    """test hook: "auto" / "wave" / "table" form of the SearchByBoW kernels (orbx_debug_set_bow_form)"""
    _check(lib().orbx_debug_set_bow_form({"auto": 0, "wave": 1, "table": 2}[form]))


def debug_set_match_items(in_memory):
    """test hook: single-pair matcher calls read their work items from mapped host memory (True) or get them by value (False, default)"""
    _check(lib().orbx_debug_set_match_items(1 if in_memory else 0))


def _check(rc):
    if rc != 0:
        raise OrbxError(rc, lib().orbx_last_error().decode(errors="replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_featset(fs):
    """dict(desc[n,32]u8, node_id u32, node_off i32, feat u32, flag u8, angle f32[, x, y, octave, u_right])
    -> (FeatSet, keepalive dict)"""
    keep = {}

    def arr(k, dt):
        v = fs.get(k)
        if v is None:
            return None
        keep[k] = np.ascontiguousarray(v, dtype=dt)
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


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:58-139).

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) as in src/Tracking.cc:124-130;
    `device`, `max_size`, `max_batch` size the HBM workspace.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0, max_size=(4096, 4096), max_batch=1):
        self._L = lib()
        self._h = C.c_void_p()
        self.nfeatures, self.nlevels, self.max_batch, self.device = nfeatures, nlevels, max_batch, device
        self._pipe_shapes, self._pipe_last = {}, (max_size[0], max_size[1])   # image sizes of the tickets in flight (pipelined forms)
        _check(self._L.orbx_extractor_create(C.byref(self._h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                             device, max_size[0], max_size[1], max_batch))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbx_extractor_destroy(h)
            self._h = None

    # -- getters (include/ORBextractor.h:78-98)
    def set_cv_profile(self, profile):
        """which OpenCV generation's GaussianBlur the descriptors are computed on: CV_PROFILE_3_2 (default) or CV_PROFILE_3_4_2"""
        _check(self._L.orbx_extractor_set_cv_profile(self._h, profile))

    def set_pyramid_group_limit(self, max_images):
        """tuning only (results do not change): launches of at most max_images images build several pyramid levels per launch; 0 = never"""
        _check(self._L.orbx_extractor_set_pyramid_group_limit(self._h, int(max_images)))

    def pipeline_warm(self, w, h):
        """make and touch every pipeline slot / kernel lane of the pipelined forms for w x h images now (orbx_pipeline_warm)"""
        _check(self._L.orbx_pipeline_warm(self._h, int(w), int(h)))

    def debug_fast_form(self):
        """1 = k_fast, 2 = k_fast2 (a pair of cells per wave) ran in the most recent extraction"""
        return self._L.orbx_debug_fast_form(self._h)

    def GetLevels(self): return self._L.orbx_get_levels(self._h)
    def GetScaleFactor(self): return self._L.orbx_get_scale_factor(self._h)

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        _check(self._L.orbx_get_scale_tables(self._h, *[_p(a) for a in t]))
        return t

    def GetScaleFactors(self): return self._tables()[0]
    def GetInverseScaleFactors(self): return self._tables()[1]
    def GetScaleSigmaSquares(self): return self._tables()[2]
    def GetInverseScaleSigmaSquares(self): return self._tables()[3]

    def GetFeaturesPerLevel(self):
        q = np.zeros(self.nlevels, np.int32)
        _check(self._L.orbx_get_features_per_level(self._h, _p(q)))
        return q

    def max_keypoints(self, w, h):
        n = self._L.orbx_max_keypoints(self._h, w, h)
        if n < 0:
            _check(n)
        return n

    # -- operator() (include/ORBextractor.h:74-76): image -> (keypoints, descriptors)
    def __call__(self, image, mask=None):
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise OrbxError(-1, "image must be 2-D uint8 (CV_8UC1, reference src/ORBextractor.cc:1269)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        _check(self._L.orbx_extract(self._h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_stereo(self, imLeft, imRight, bf, min_z):
        """one stereo frame in one call: both eyes' operator() + Frame::ComputeStereoMatches (reference src/Frame.cc:82-97)
        -> (kpsL, descL, kpsR, descR, uRight, depth); the extractor needs max_batch >= 2"""
        L_ = np.ascontiguousarray(imLeft, np.uint8); R_ = np.ascontiguousarray(imRight, np.uint8)
        if L_.shape != R_.shape or L_.ndim != 2:
            raise OrbxError(-1, "left and right image must be equal 2-D uint8 arrays")
        h, w = L_.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros((2, cap), KP_DTYPE); desc = np.zeros((2, cap, 32), np.uint8); n = np.zeros(2, np.int32)
        ur = np.zeros(cap, np.float32); z = np.zeros(cap, np.float32)
        _check(self._L.orbx_extract_stereo(self._h, _p(L_), _p(R_), w, h, L_.strides[0], bf, min_z, _p(kps), _p(desc), cap, _p(n), _p(ur), _p(z)))
        return (kps[0, :n[0]].copy(), desc[0, :n[0]].copy(), kps[1, :n[1]].copy(), desc[1, :n[1]].copy(), ur[:n[0]].copy(), z[:n[0]].copy())

    def extract_submit(self, image):
        """pipelined monocular form of __call__ -> ticket"""
        if image.ndim != 2 or image.dtype != np.uint8 or image.strides[1] != 1:
            raise OrbxError(-1, "image must be a 2-D uint8 array with unit column stride")
        h, w = image.shape
        t = C.c_int()
        _check(self._L.orbx_extract_submit(self._h, image.ctypes.data, w, h, image.strides[0], C.byref(t)))
        self._pipe_shapes[t.value] = (w, h); self._pipe_last = (w, h)
        return t.value

    def extract_wait(self, ticket):
        w, h = self._pipe_shapes.pop(ticket, self._pipe_last)       # an unknown ticket is the library's error to report
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        _check(self._L.orbx_extract_wait(self._h, ticket, _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # -- pipelined form: up to pipeline_depth() frames in flight, uploads / downloads overlap the kernels of the neighbouring frames
    def extract_stereo_submit(self, imLeft, imRight, bf, min_z):
        """-> ticket; the images are copied (or, if they live in pinned_array() memory, must stay untouched until the wait)"""
        if imLeft.shape != imRight.shape or imLeft.ndim != 2 or imLeft.dtype != np.uint8 or imLeft.strides != imRight.strides or imLeft.strides[1] != 1:
            raise OrbxError(-1, "left and right image must be equal 2-D uint8 arrays with unit column stride")
        h, w = imLeft.shape
        t = C.c_int()
        _check(self._L.orbx_extract_stereo_submit(self._h, imLeft.ctypes.data, imRight.ctypes.data, w, h, imLeft.strides[0], bf, min_z, C.byref(t)))
        self._pipe_shapes[t.value] = (w, h); self._pipe_last = (w, h)
        return t.value

    def extract_stereo_wait(self, ticket, copy=True):
        w, h = self._pipe_shapes.pop(ticket, self._pipe_last)       # an unknown ticket is the library's error to report
        cap = self.max_keypoints(w, h)
        b = getattr(self, "_pipe_out", None)
        if b is None or b[0].shape[1] != cap:
            b = self._pipe_out = (np.zeros((2, cap), KP_DTYPE), np.zeros((2, cap, 32), np.uint8), np.zeros(2, np.int32),
                                  np.zeros(cap, np.float32), np.zeros(cap, np.float32))
        kps, desc, n, ur, z = b
        _check(self._L.orbx_extract_stereo_wait(self._h, ticket, _p(kps), _p(desc), cap, _p(n), _p(ur), _p(z)))
        if not copy:      # views into buffers that the next wait overwrites (measurement loops)
            return kps[0, :n[0]], desc[0, :n[0]], kps[1, :n[1]], desc[1, :n[1]], ur[:n[0]], z[:n[0]]
        return (kps[0, :n[0]].copy(), desc[0, :n[0]].copy(), kps[1, :n[1]].copy(), desc[1, :n[1]].copy(), ur[:n[0]].copy(), z[:n[0]].copy())

    def extract_color(self, image, rgb=True, want_gray=False):
        """colour frame (H x W x 3|4 uint8): cvtColor to grey on device (Tracking::GrabImage*, src/Tracking.cc:177-202), then operator()"""
        image = np.ascontiguousarray(image, np.uint8)
        h, w, ch = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        gray = np.zeros((h, w), np.uint8) if want_gray else None
        _check(self._L.orbx_extract_color(self._h, _p(image), w, h, image.strides[0], ch, int(rgb), _p(kps), _p(desc), cap, C.byref(n),
                                          _p(gray), gray.strides[0] if want_gray else 0))
        out = (kps[:n.value].copy(), desc[:n.value].copy())
        return out + (gray,) if want_gray else out

    def extract_rectified(self, rectifier, image, want_rect=False):
        """raw grey frame -> cv::remap on device (Examples/Stereo/stereo_euroc.cc:136-137) -> operator()"""
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        rw, rh = rectifier.size
        cap = self.max_keypoints(rw, rh)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        rect = np.zeros((rh, rw), np.uint8) if want_rect else None
        _check(self._L.orbx_extract_rectified(self._h, rectifier._h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n),
                                              _p(rect), rect.strides[0] if want_rect else 0))
        out = (kps[:n.value].copy(), desc[:n.value].copy())
        return out + (rect,) if want_rect else out

    def extract_batch(self, images):
        """list/array of equally sized uint8 images -> list of (keypoints, descriptors)"""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        h, w = imgs[0].shape
        B = len(imgs)
        cap = self.max_keypoints(w, h)
        ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
        kps = np.zeros((B, cap), KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); n = np.zeros(B, np.int32)
        _check(self._L.orbx_extract_batch(self._h, ptrs, B, w, h, imgs[0].strides[0], _p(kps), _p(desc), cap, _p(n)))
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(B)]

    def extract_batch_device(self, d_imgs, img_stride, pitch, batch, w, h, d_kps, d_desc, cap, d_n, stream=None):
        """device pointers (ints); asynchronous on `stream` (None = the handle's own)"""
        _check(self._L.orbx_extract_batch_device(self._h, d_imgs, img_stride, pitch, batch, w, h, d_kps, d_desc, cap, d_n, stream))

    def sync(self, stream=None):
        _check(self._L.orbx_sync(self._h, stream))

    # -- mvImagePyramid (include/ORBextractor.h:100)
    def pyramid_level(self, level, image_index=0):
        w, h = C.c_int(), C.c_int()
        _check(self._L.orbx_pyramid_level(self._h, image_index, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(self._L.orbx_pyramid_level(self._h, image_index, level, _p(out), out.strides[0], C.byref(w), C.byref(h)))
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # -- inspection / measurement
    def debug_candidates(self, level, image_index=0):
        n = C.c_int()
        _check(self._L.orbx_debug_candidates(self._h, image_index, level, None, None, None, 0, C.byref(n)))
        x = np.zeros(n.value, np.int32); y = np.zeros(n.value, np.int32); r = np.zeros(n.value, np.int32)
        _check(self._L.orbx_debug_candidates(self._h, image_index, level, _p(x), _p(y), _p(r), n.value, C.byref(n)))
        return x, y, r

    def debug_level_counts(self, image_index=0):
        c = np.zeros(self.nlevels, np.int32)
        _check(self._L.orbx_debug_level_counts(self._h, image_index, _p(c)))
        return c

    def profile_enable(self, on=True):
        _check(self._L.orbx_profile_enable(self._h, int(on)))

    def profile_stages(self, mask=0xFFFFFFFF):
        """which stages record events while profiling is on (bit = ORBX_STAGE_*; default all)"""
        _check(self._L.orbx_profile_stages(self._h, int(mask) & 0xFFFFFFFF))

    def profile_read(self, reset=True):
        ms = np.zeros(len(STAGES), np.float32); n = np.zeros(len(STAGES), np.int32)
        _check(self._L.orbx_profile_read(self._h, _p(ms), _p(n), int(reset)))
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(STAGES)}


def ComputeStereoMatches(extractorLeft, extractorRight, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mbf, mb):
    """Mirror of Frame::ComputeStereoMatches (reference src/Frame.cc:577-751) -> (mvuRight, mvDepth).
    `mb` is the stereo baseline in metres (minZ); the reference reads it uninitialised."""
    kL = np.ascontiguousarray(mvKeys, dtype=KP_DTYPE); kR = np.ascontiguousarray(mvKeysRight, dtype=KP_DTYPE)
    dL = np.ascontiguousarray(mDescriptors, dtype=np.uint8); dR = np.ascontiguousarray(mDescriptorsRight, dtype=np.uint8)
    ur = np.full(len(kL), -1, np.float32); dp = np.full(len(kL), -1, np.float32)
    _check(lib().orbx_stereo_match(extractorLeft._h, extractorRight._h, _p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR),
                                   mbf, mb, _p(ur), _p(dp)))
    return ur, dp


ROWTAB_FROM_KEYPOINTS, ROWTAB_OF_EXTRACTION = 0, 1


def stereo_match_batch_device(L, imgL0, R, imgR0, batch, d_kL, d_dL, d_nL, d_kR, d_dR, d_nR, cap, bf, min_z, d_ur, d_depth, stream=None,
                              row_table=ROWTAB_FROM_KEYPOINTS):
    """row_table: ROWTAB_FROM_KEYPOINTS (always correct: the table is built from d_kR) or ROWTAB_OF_EXTRACTION (the caller asserts that
    d_kR still holds what R's last extract_batch_device wrote: that launch's by-product table is used; refused if it cannot be)"""
    _check(lib().orbx_stereo_match_batch_device(L._h, imgL0, R._h, imgR0, batch, d_kL, d_dL, d_nL, d_kR, d_dR, d_nR, cap,
                                                bf, min_z, d_ur, d_depth, row_table, stream))


def stereo_row_table_available(R, d_kR, imgR0, batch, cap):
    return bool(lib().orbx_stereo_row_table_available(R._h, d_kR, imgR0, batch, cap))


class ORBVocabulary:
    """Mirror of ORB_SLAM2::ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, reference
    include/ORBVocabulary.h) for the part on the hot path: loadFromTextFile and transform()."""

    def __init__(self, k=None, L=None, parent=None, is_leaf=None, desc=None, weight=None, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        if k is not None:
            parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
            _check(self._L.orbx_vocab_create(device, k, L, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight), C.byref(self._h)))

    @classmethod
    def loadFromTextFile(cls, filename, device=0):
        v = cls(device=device)
        _check(v._L.orbx_vocab_load_text(device, str(filename).encode(), C.byref(v._h)))
        return v

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbx_vocab_destroy(h)
            self._h = None

    def info(self):
        k, L_, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(self._L.orbx_vocab_info(self._h, C.byref(k), C.byref(L_), C.byref(n), C.byref(w)))
        return dict(k=k.value, L=L_.value, nodes=n.value, words=w.value)

    def transform(self, features, levelsup=4):
        """-> dict(word_id, word_weight, node_id, bow_id, bow_val, fv_node_id, fv_node_off, fv_feat)"""
        desc = np.ascontiguousarray(features, np.uint8).reshape(-1, 32); n = len(desc)
        wid = np.zeros(n, np.uint32); ww = np.zeros(n, np.float64); nid = np.zeros(n, np.uint32)
        bid = np.zeros(n, np.uint32); bval = np.zeros(n, np.float64); nb = C.c_int()
        fid = np.zeros(n, np.uint32); foff = np.zeros(n + 1, np.int32); ffeat = np.zeros(n, np.uint32); fn = C.c_int()
        _check(self._L.orbx_bow_transform(self._h, _p(desc), n, levelsup, _p(wid), _p(ww), _p(nid), _p(bid), _p(bval), C.byref(nb),
                                          _p(fid), _p(foff), _p(ffeat), C.byref(fn)))
        return dict(word_id=wid, word_weight=ww, node_id=nid, bow_id=bid[:nb.value].copy(), bow_val=bval[:nb.value].copy(),
                    fv_node_id=fid[:fn.value].copy(), fv_node_off=foff[:fn.value + 1].copy(), fv_feat=ffeat[:foff[fn.value]].copy())


class DeviceKeyFrame:
    """orbx_kf: a keyframe's (or frame's) immutable matching data resident in HBM in FeatureVector order: descriptors,
    FeatureVector, angles and, when given, positions / octaves / u_right.  Map-point flags are passed per search."""

    def __init__(self, fs, device=0):
        s, keep = make_featset(fs)
        h = C.c_void_p()
        _check(lib().orbx_kf_create(device, C.byref(s), C.byref(h)))
        self._h, self.n, self.device = h, s.n, device

    def __del__(self):
        # at interpreter shutdown the module globals (and ctypes itself) may already be torn down: never raise from a finaliser
        try:
            h, L = getattr(self, "_h", None), _lib
            if h and L is not None:
                L.orbx_kf_destroy(h)
            self._h = None
        except Exception:
            pass


def _flags_for(flag, kf, who):
    """per-feature flag array of a resident keyframe: one byte per feature of THAT keyframe (the library reads kf.n bytes)"""
    fl = np.ascontiguousarray(flag, np.uint8)
    if fl.ndim != 1 or len(fl) != kf.n:
        raise OrbxError(-1, f"{who}: flag array has {fl.size} entries, the keyframe has {kf.n} features")
    return fl


class BowDatabase:
    """Device-resident keyframe set (include/orbx.h: orbx_bowdb_*): upload once, search many frames."""

    def __init__(self, keyframes, device=0):
        self._L = lib()
        sets = [make_featset(k) for k in keyframes]
        arr = (FeatSet * len(sets))(*[s_[0] for s_ in sets])
        self._h = C.c_void_p()
        _check(self._L.orbx_bowdb_create(device, arr, len(sets), C.byref(self._h)))
        self.nkf = len(sets)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbx_bowdb_destroy(h)
            self._h = None

    def search(self, F, nnratio=0.75, checkOri=True):
        b, kb = make_featset(F)
        out = np.full((self.nkf, b.n), -1, np.int32); n = np.zeros(self.nkf, np.int32)
        _check(self._L.orbx_bowdb_search(self._h, C.byref(b), nnratio, int(checkOri), _p(out), _p(n)))
        return out, n


class BowFrames:
    """Device-resident Frame::ComputeBoW results of a batch of frames (include/orbx.h: orbx_bow_frames_*)."""

    def __init__(self, max_batch, cap, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        _check(self._L.orbx_bow_frames_create(device, max_batch, cap, C.byref(self._h)))
        self.batch, self.cap = max_batch, cap

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbx_bow_frames_destroy(h)
            self._h = None

    def transform(self, voc, d_kps, d_desc, d_n, batch, levelsup=4, stream=None):
        """device pointers of orbx_extract_batch_device's outputs -> BowVector / FeatureVector in HBM (asynchronous)"""
        _check(self._L.orbx_bow_transform_batch_device(voc._h, self._h, d_kps, d_desc, d_n, batch, levelsup, stream))

    def read(self, index, stream=None):
        cap = self.cap
        bid = np.zeros(cap, np.uint32); bval = np.zeros(cap, np.float64); nb = C.c_int()
        fid = np.zeros(cap, np.uint32); foff = np.zeros(cap + 1, np.int32); ffeat = np.zeros(cap, np.uint32); fn = C.c_int()
        _check(self._L.orbx_bow_frames_read(self._h, index, stream, _p(bid), _p(bval), C.byref(nb), _p(fid), _p(foff), _p(ffeat), C.byref(fn)))
        return dict(bow_id=bid[:nb.value].copy(), bow_val=bval[:nb.value].copy(), fv_node_id=fid[:fn.value].copy(),
                    fv_node_off=foff[:fn.value + 1].copy(), fv_feat=ffeat[:foff[fn.value]].copy())

    def search(self, db, batch, d_match, d_nmatches, nnratio=0.75, checkOri=True, stream=None):
        """every keyframe of a BowDatabase against frames 0..batch-1: d_match[batch][nkf][cap], d_nmatches[batch][nkf] (device)"""
        _check(self._L.orbx_bowdb_search_batch_device(db._h, self._h, batch, nnratio, int(checkOri), d_match, d_nmatches, stream))

    def search_compact(self, db, batch, d_pairs, cap_pairs, d_nmatches, nnratio=0.75, checkOri=True, stream=None):
        """the same search with compact results: d_pairs[batch][nkf][cap_pairs][2] int32 = (frame feature, keyframe feature) in frame-feature order,
        d_nmatches[batch][nkf] = counts (orbx_bowdb_search_batch_device_compact)"""
        _check(self._L.orbx_bowdb_search_batch_device_compact(db._h, self._h, batch, nnratio, int(checkOri), d_pairs, int(cap_pairs), d_nmatches, stream))


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:41-103) for the searches on the
    north-star path.  Feature sets are dicts as accepted by make_featset()."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30  # src/ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio, self.mbCheckOrientation, self.device = float(nnratio), bool(checkOri), device

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return lib().orbx_hamming(_p(a), _p(b))

    def SearchByBoW(self, pKF, other, frame=None):
        """SearchByBoW(pKF, F) when `other` is a frame feature set (kind='frame', default) ->
        (match_f, nmatches); SearchByBoW(pKF1, pKF2) when other['kind']=='keyframe' -> (match12, nmatches)."""
        a, ka = make_featset(pKF); b, kb = make_featset(other)
        n = C.c_int()
        if other.get("kind", "frame") == "keyframe":
            out = np.full(a.n, -1, np.int32)
            _check(lib().orbx_search_by_bow_kf_kf(self.device, C.byref(a), C.byref(b), self.mfNNratio, int(self.mbCheckOrientation), _p(out), C.byref(n)))
        else:
            out = np.full(b.n, -1, np.int32)
            _check(lib().orbx_search_by_bow_kf_f(self.device, C.byref(a), C.byref(b), self.mfNNratio, int(self.mbCheckOrientation), _p(out), C.byref(n)))
        return out, n.value

    def SearchByBoWBatch(self, keyframes, F):
        """Relocalization loop (src/Tracking.cc:1661-1682): every keyframe against one frame."""
        sets = [make_featset(k) for k in keyframes]
        arr = (FeatSet * len(sets))(*[s[0] for s in sets])
        b, kb = make_featset(F)
        out = np.full((len(sets), b.n), -1, np.int32); n = np.zeros(len(sets), np.int32)
        _check(lib().orbx_search_by_bow_kf_f_batch(self.device, arr, len(sets), C.byref(b), self.mfNNratio, int(self.mbCheckOrientation), _p(out), _p(n)))
        return out, n

    def SearchByBoWKeyFrames(self, pKF1, keyframes2):
        """LoopClosing::ComputeSim3's loop (src/LoopClosing.cc:293-323): SearchByBoW(pKF1, pKF2) for every candidate in
        one call -> (match12[n2, n1], nmatches[n2])"""
        a, ka = make_featset(pKF1)
        sets = [make_featset(k) for k in keyframes2]
        arr = (FeatSet * len(sets))(*[s[0] for s in sets])
        out = np.full((len(sets), a.n), -1, np.int32); n = np.zeros(len(sets), np.int32)
        _check(lib().orbx_search_by_bow_kf_kf_batch(self.device, C.byref(a), arr, len(sets), self.mfNNratio, int(self.mbCheckOrientation), _p(out), _p(n)))
        return out, n

    def SearchForTriangulationBatch(self, pKF1, keyframes2, F12s, epipoles, scaleFactors2, levelSigma2_2, bOnlyStereo=False):
        """LocalMapping::CreateNewMapPoints' loop (src/LocalMapping.cc:241-309): the current keyframe against every neighbour in
        one call -> list of (npairs_i, 2) arrays"""
        a, ka = make_featset(pKF1)
        sets = [make_featset(k) for k in keyframes2]
        arr = (FeatSet * len(sets))(*[s[0] for s in sets])
        F = np.ascontiguousarray(np.asarray(F12s, np.float32).reshape(len(sets), 9))
        ep = np.ascontiguousarray(np.asarray(epipoles, np.float32).reshape(len(sets), 2))
        sf = np.ascontiguousarray(scaleFactors2, np.float32); sg = np.ascontiguousarray(levelSigma2_2, np.float32)
        cap = max(a.n, 1)
        pairs = np.zeros((len(sets), cap, 2), np.int32); n = np.zeros(len(sets), np.int32)
        _check(lib().orbx_search_for_triangulation_batch(self.device, C.byref(a), arr, len(sets), _p(F), _p(ep), _p(sf), _p(sg), len(sf),
                                                         int(bOnlyStereo), int(self.mbCheckOrientation), _p(pairs), cap, _p(n)))
        return [pairs[i, :n[i]].copy() for i in range(len(sets))]

    # ---- the same searches on resident keyframes (DeviceKeyFrame): flags travel per call
    def SearchByBoWResident(self, kf, kf_flag, frame):
        """SearchByBoW(pKF, F) with both sides resident -> (match_f, nmatches)"""
        fl = _flags_for(kf_flag, kf, "SearchByBoWResident")
        out = np.full(frame.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_kf_search_by_bow_kf_f(kf._h, _p(fl), frame._h, self.mfNNratio, int(self.mbCheckOrientation), _p(out), C.byref(n)))
        return out, n.value

    def SearchByBoWKeyFramesFrameResident(self, kfs, kf_flags, frame):
        """Tracking::Relocalization's loop (src/Tracking.cc:1661-1682) as one call: SearchByBoW(pKF, F) for every resident keyframe of `kfs`
        against one frame given as a host feature set (dict) -> (match_f[nkf][nF], nmatches[nkf])"""
        if len(kf_flags) != len(kfs):
            raise OrbxError(-1, "SearchByBoWKeyFramesFrameResident: one flag array per keyframe")
        fl = [_flags_for(x, k, "SearchByBoWKeyFramesFrameResident") for x, k in zip(kf_flags, kfs)]
        hs = (C.c_void_p * len(kfs))(*[k._h for k in kfs])
        fp = (C.c_void_p * len(kfs))(*[x.ctypes.data for x in fl])
        fs, keep = make_featset(frame)
        out = np.full((len(kfs), fs.n), -1, np.int32); n = np.zeros(len(kfs), np.int32)
        _check(lib().orbx_kf_search_by_bow_kfs_f(hs, fp, len(kfs), C.byref(fs), self.mfNNratio, int(self.mbCheckOrientation), _p(out), _p(n)))
        return out, n

    def SearchByBoWKeyFramesResident(self, kf1, flag1, kfs2, flags2):
        f1 = _flags_for(flag1, kf1, "SearchByBoWKeyFramesResident")
        if len(flags2) != len(kfs2):
            raise OrbxError(-1, "SearchByBoWKeyFramesResident: one flag array per second keyframe")
        f2 = [_flags_for(x, k, "SearchByBoWKeyFramesResident") for x, k in zip(flags2, kfs2)]
        hs = (C.c_void_p * len(kfs2))(*[k._h for k in kfs2])
        fp = (C.c_void_p * len(kfs2))(*[x.ctypes.data for x in f2])
        out = np.full((len(kfs2), kf1.n), -1, np.int32); n = np.zeros(len(kfs2), np.int32)
        _check(lib().orbx_kf_search_by_bow_kf_kf(kf1._h, _p(f1), hs, fp, len(kfs2), self.mfNNratio, int(self.mbCheckOrientation), _p(out), _p(n)))
        return out, n

    def SearchForTriangulationResident(self, kf1, flag1, kfs2, flags2, F12s, epipoles, scaleFactors2, levelSigma2_2, bOnlyStereo=False):
        f1 = _flags_for(flag1, kf1, "SearchForTriangulationResident") if flag1 is not None else None
        if flags2 is not None and len(flags2) != len(kfs2):
            raise OrbxError(-1, "SearchForTriangulationResident: one flag array per second keyframe")
        f2 = [_flags_for(x, k, "SearchForTriangulationResident") for x, k in zip(flags2, kfs2)] if flags2 is not None else None
        hs = (C.c_void_p * len(kfs2))(*[k._h for k in kfs2])
        fp = (C.c_void_p * len(kfs2))(*[x.ctypes.data for x in f2]) if f2 is not None else None
        F = np.ascontiguousarray(np.asarray(F12s, np.float32).reshape(len(kfs2), 9))
        ep = np.ascontiguousarray(np.asarray(epipoles, np.float32).reshape(len(kfs2), 2))
        sf = np.ascontiguousarray(scaleFactors2, np.float32); sg = np.ascontiguousarray(levelSigma2_2, np.float32)
        cap = max(kf1.n, 1)
        pairs = np.zeros((len(kfs2), cap, 2), np.int32); n = np.zeros(len(kfs2), np.int32)
        _check(lib().orbx_kf_search_for_triangulation(kf1._h, _p(f1), hs, fp, len(kfs2), _p(F), _p(ep), _p(sf), _p(sg), len(sf),
                                                      int(bOnlyStereo), int(self.mbCheckOrientation), _p(pairs), cap, _p(n)))
        return [pairs[i, :n[i]].copy() for i in range(len(kfs2))]

    @staticmethod
    def _frame(d):
        keep = {}
        def arr(k, dt):
            keep[k] = np.ascontiguousarray(d[k], dtype=dt); return keep[k].ctypes.data
        s_ = FrameFeats()
        s_.x = arr("x", np.float32); s_.y = arr("y", np.float32); s_.octave = arr("octave", np.int32); s_.angle = arr("angle", np.float32)
        s_.u_right = arr("u_right", np.float32); s_.desc = arr("desc", np.uint8)
        s_.occupied = arr("occupied", np.uint8) if d.get("occupied") is not None else None
        s_.n = len(keep["x"])
        s_.min_x, s_.min_y, s_.max_x, s_.max_y = [float(v) for v in d["bounds"]]
        return s_, keep

    @staticmethod
    def _points(d):
        keep = {}
        def arr(k, dt):
            if d.get(k) is None:
                return None
            keep[k] = np.ascontiguousarray(d[k], dtype=dt); return keep[k].ctypes.data
        s_ = ProjPoints()
        s_.u = arr("u", np.float32); s_.v = arr("v", np.float32); s_.aux = arr("aux", np.float32); s_.level = arr("level", np.int32)
        s_.angle = arr("angle", np.float32); s_.view_cos = arr("view_cos", np.float32); s_.desc = arr("desc", np.uint8)
        s_.valid = arr("valid", np.uint8); s_.has_obs = arr("has_obs", np.uint8)
        s_.n = len(keep["u"])
        return s_, keep

    def SearchByProjectionLastFrame(self, CurrentFrame, LastFramePoints, scaleFactors, th, direction=0, mbf=0.0):
        """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), src/ORBmatcher.cc:1396-1553
        -> (match_cur, nmatches)"""
        a, ka = self._frame(CurrentFrame); b, kb = self._points(LastFramePoints)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_by_projection_last_frame(self.device, C.byref(a), C.byref(b), _p(sf), len(sf), th, direction, mbf,
                                                          int(self.mbCheckOrientation), _p(out), C.byref(n)))
        return out, n.value

    def SearchByProjectionMapPoints(self, F, vpMapPoints, scaleFactors, th=3.0):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th), src/ORBmatcher.cc:48-129 -> (match_cur, nmatches)"""
        a, ka = self._frame(F); b, kb = self._points(vpMapPoints)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_by_projection_map_points(self.device, C.byref(a), C.byref(b), _p(sf), len(sf), th, self.mfNNratio,
                                                          _p(out), C.byref(n)))
        return out, n.value

    def SearchByProjectionKeyFrame(self, CurrentFrame, KFPoints, scaleFactors, th, ORBdist):
        """SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1555-1685
        -> (match_cur, nmatches)"""
        a, ka = self._frame(CurrentFrame); b, kb = self._points(KFPoints)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_by_projection_keyframe(self.device, C.byref(a), C.byref(b), _p(sf), len(sf), th, int(ORBdist),
                                                        int(self.mbCheckOrientation), _p(out), C.byref(n)))
        return out, n.value

    def SearchByProjectionSim3(self, pKF, vpPoints, scaleFactors, th):
        """SearchByProjection(KeyFrame *pKF, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:305-415 -> (match_kf, nmatches)"""
        a, ka = self._frame(pKF); b, kb = self._points(vpPoints)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_by_projection_sim3(self.device, C.byref(a), C.byref(b), _p(sf), len(sf), th, _p(out), C.byref(n)))
        return out, n.value

    def Fuse(self, pKF, vpMapPoints, scaleFactors, invLevelSigma2=None, th=3.0, max_dist=50):
        """search half of both Fuse overloads (src/ORBmatcher.cc:873-1164): invLevelSigma2 given -> the chi2-gated
        variant of Fuse(pKF, vpMapPoints, th).  -> (best_idx, best_dist, nfound)"""
        a, ka = self._frame(pKF); b, kb = self._points(vpMapPoints)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        sg = None if invLevelSigma2 is None else np.ascontiguousarray(invLevelSigma2, np.float32)
        bi = np.full(b.n, -1, np.int32); bd = np.full(b.n, 256, np.int32); n = C.c_int()
        _check(lib().orbx_window_best(self.device, C.byref(a), C.byref(b), _p(sf), None if sg is None else _p(sg), len(sf), th,
                                      0 if sg is None else 1, int(max_dist), _p(bi), _p(bd), C.byref(n)))
        return bi, bd, n.value

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=100):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:430-556
        -> (vnMatches12, nmatches, updated vbPrevMatched)"""
        a, ka = self._frame(F1); b, kb = self._frame(F2)
        xy = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_for_initialization(self.device, C.byref(a), C.byref(b), _p(xy), int(windowSize), self.mfNNratio,
                                                    int(self.mbCheckOrientation), _p(out), C.byref(n)))
        new_xy = xy.copy(); m = out >= 0   # :544-546
        new_xy[m, 0] = kb["x"][out[m]]; new_xy[m, 1] = kb["y"][out[m]]
        return out, n.value, new_xy

    def SearchBySim3(self, pKF1, pKF2, pts12, pts21, scaleFactors1, scaleFactors2, th):
        """SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1166-1394 -> (match12, nFound)"""
        a, ka = self._frame(pKF1); b, kb = self._frame(pKF2); p, kp = self._points(pts12); q, kq = self._points(pts21)
        s1 = np.ascontiguousarray(scaleFactors1, np.float32); s2 = np.ascontiguousarray(scaleFactors2, np.float32)
        out = np.full(a.n, -1, np.int32); n = C.c_int()
        _check(lib().orbx_search_by_sim3(self.device, C.byref(a), C.byref(b), C.byref(p), C.byref(q), _p(s1), _p(s2), len(s1), th,
                                         _p(out), C.byref(n)))
        return out, n.value

    def SearchForTriangulation(self, pKF1, pKF2, F12, ex, ey, scaleFactors2, levelSigma2_2, bOnlyStereo=False):
        a, ka = make_featset(pKF1); b, kb = make_featset(pKF2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scaleFactors2, np.float32); sg = np.ascontiguousarray(levelSigma2_2, np.float32)
        cap = max(a.n, 1)
        pairs = np.zeros((cap, 2), np.int32); n = C.c_int()
        _check(lib().orbx_search_for_triangulation(self.device, C.byref(a), C.byref(b), _p(F), ex, ey, _p(sf), _p(sg), len(sf),
                                                   int(bOnlyStereo), int(self.mbCheckOrientation), _p(pairs), cap, C.byref(n)))
        return pairs[:n.value].copy()


def UndistortKeyPoints(xy, fx, fy, cx, cy, distCoef, device=0):
    """Frame::UndistortKeyPoints (reference src/Frame.cc:470-515): cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK)"""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); d = np.ascontiguousarray(distCoef, np.float32)
    out = np.zeros_like(xy)
    _check(lib().orbx_undistort_keypoints(device, _p(xy), len(xy), fx, fy, cx, cy, _p(d), len(d), _p(out)))
    return out


class Rectifier:
    """cv::remap(src, dst, M1, M2, INTER_LINEAR) with the CV_32FC1 maps of cv::initUndistortRectifyMap
    (reference Examples/Stereo/stereo_euroc.cc:103-104, :136-137), held on the device in OpenCV's fixed-point form."""

    def __init__(self, src_size, map_x, map_y, device=0):
        self._L = lib()
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        if mx.ndim != 2 or mx.shape != my.shape:
            raise OrbxError(-1, "map_x / map_y must be equal 2-D float32 arrays")
        h = C.c_void_p()
        _check(self._L.orbx_rectifier_create(device, int(src_size[0]), int(src_size[1]), mx.shape[1], mx.shape[0], _p(mx), _p(my), C.byref(h)))
        self._h = h; self.size = (mx.shape[1], mx.shape[0]); self.src_size = (int(src_size[0]), int(src_size[1]))

    def remap_batch_device(self, d_src, src_stride, src_pitch, batch, d_dst, dst_stride, dst_pitch, stream=None):
        _check(self._L.orbx_remap_batch_device(self._h, d_src, src_stride, src_pitch, batch, d_dst, dst_stride, dst_pitch, stream))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.orbx_rectifier_destroy(self._h); self._h = None
        except Exception:
            pass


def ComputeDistinctiveDescriptors(descriptor_sets, device=0):
    """Batched MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:266-340): one [n_i,32] uint8
    array per map point -> BestIdx per map point (-1 for an empty one)."""
    sets = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descriptor_sets]
    off = np.zeros(len(sets) + 1, np.int32)
    off[1:] = np.cumsum([len(d) for d in sets])
    flat = np.concatenate(sets) if len(sets) and off[-1] else np.zeros((0, 32), np.uint8)
    out = np.zeros(len(sets), np.int32)
    _check(lib().orbx_distinctive_descriptors(device, _p(flat), _p(off), len(sets), _p(out)))
    return out
