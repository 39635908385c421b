"""Per-call latency of the three north-star matchers, called the way ORB-SLAM2 calls them (one keyframe pair per call,
host pointers in, host pointers out), with the CPU oracle timed beside each on the same inputs: bench.py's
`config.matchers` block.

Workload: 1000 x 1000 features, FeatureVectors over the 100 level-2 nodes of a k = 10 vocabulary (what DBoW2 gives
ORB-SLAM2 at levelsup = 4), second side = first side with 4 % of the descriptor bits flipped, permuted; 60 % of the
keyframe features hold map points (SearchByBoW), 40 % already triangulated (SearchForTriangulation).
  * SearchByBoW(KF, F)        src/ORBmatcher.cc:171-303, called per frame by Tracking::TrackReferenceKeyFrame
  * SearchByBoW(KF, KF)       :568-702, per loop candidate by LoopClosing::ComputeSim3 (src/LoopClosing.cc:293-323)
  * SearchForTriangulation    :704-871, per neighbour keyframe by LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:241-309)
Timed region = the C entry point only (ctypes structs prebuilt) for both sides; median of `reps` calls.
Run alone on the GPU box:  python tools/matcher_bench.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402


def make_sets(seed=5, n=1000, nkf=20, flip=0.04):
    """-> (current keyframe / frame side, [nkf neighbour keyframes], per-pair F12 / epipoles, scale tables)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    voc = synth.Vocab2(seed + 1, k=10)
    voc.seed_from(d, rng)
    x = rng.uniform(20, 1220, n).astype(np.float32); y = rng.uniform(20, 356, n).astype(np.float32)
    octave = rng.integers(0, 8, n).astype(np.int32)
    angle = rng.uniform(0, 360, n).astype(np.float32)

    def mk(desc, xx, yy, oc, an, flag):
        ids, off, feat = voc.feature_vector(desc)
        return dict(desc=np.ascontiguousarray(desc), node_id=ids, node_off=off, feat=feat, flag=flag, angle=an, x=xx, y=yy, octave=oc,
                    u_right=np.where(rng.random(len(desc)) < 0.3, 5.0, -1.0).astype(np.float32))
    cur = mk(d, x, y, octave, angle, (rng.random(n) < 0.6).astype(np.uint8))
    kfs, Fs, eps = [], [], []
    for i in range(nkf):
        perm = rng.permutation(n)
        dk = synth.flip_bits(rng, d, flip)[perm]
        shift = np.float32(rng.uniform(-40, 40))
        kfs.append(mk(dk, (x[perm] + shift).astype(np.float32), (y[perm] + rng.normal(0, 0.4, n)).astype(np.float32), octave[perm],
                      ((angle[perm] + rng.normal(0, 5, n)) % 360).astype(np.float32), (rng.random(n) < 0.6).astype(np.uint8)))
        F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + np.float32(2e-7) * rng.normal(0, 1, (3, 3)).astype(np.float32)
        Fs.append(F.astype(np.float32)); eps.append((float(rng.uniform(-500, 1700)), float(rng.uniform(0, 376))))
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    return cur, kfs, Fs, eps, sf.astype(np.float32), (sf * sf).astype(np.float32)


def adapter_timing(reps=60):
    """the batched / resident matcher forms through the COMPILED C++ adaptor (adapter/ORBmatcher_batch.cc, ORBmatcher_bow.cc), timed by the
    driver's own std::chrono clock around each adaptor call: map-point flags under the keyframe's accessors, FeatureVector flatten (host-pointer
    forms), epipoles, the ABI call and the mapping of the indices back to MapPoint* -- what ORB-SLAM2's threads would see.  Scene: the 1005
    left-eye features of a synthetic KITTI-shape frame and 20 derived keyframes over a 100-node vocabulary rule (tests/adapter_driver.cc, mode
    `batch`; tests/test_adapter.py::test_adapter_batched_resident_matchers compares every result of that mode with the oracle)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "examples", "adapter_bench")
    if not os.path.exists(exe):
        return {"error": "examples/adapter_bench not built (python -c 'import __graft_entry__ as g; g.build()')"}
    w, h = 1241, 376
    left, right, _ = synth.stereo_pair(616, w, h)
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.bin"), os.path.join(td, "out.txt")
        with open(inp, "wb") as f:
            f.write(np.array([w, h], np.int32).tobytes()); f.write(left.tobytes()); f.write(right.tobytes())
        run = subprocess.run([exe, "batch", inp, outp, str(reps)], capture_output=True, text=True, timeout=300)
        if run.returncode != 0:
            return {"error": (run.stderr or run.stdout)[-300:]}
        r = {}
        for line in open(outp):
            t = line.split(None, 2)
            if t[0] in ("time_ns", "time_reps", "features", "single_equals_batch", "after_drop_equal"):
                r[t[0]] = [int(v) for v in t[2].split()]
    t = r["time_ns"]
    nk = 20
    return {"unit": "us, median of %d adaptor calls, std::chrono inside the C++ driver" % r["time_reps"][0], "features_per_keyframe": r["features"][0], "pairs_per_batch": nk,
            "batch_us_per_pair": {"SearchForTriangulationBatch (LocalMapping::CreateNewMapPoints)": round(t[0] / 1e3 / nk, 2),
                                  "SearchByBoWBatch(KF, KFs) (LoopClosing::ComputeSim3)": round(t[1] / 1e3 / nk, 2),
                                  "SearchByBoWBatch(KFs, F) (Tracking::Relocalization)": round(t[2] / 1e3 / nk, 2)},
            "single_resident_us": {"SearchForTriangulation": round(t[3] / 1e3, 1), "SearchByBoW(KF, KF)": round(t[4] / 1e3, 1)},
            "single_host_pointers_us": {"SearchForTriangulation": round(t[5] / 1e3, 1), "SearchByBoW(KF, KF)": round(t[6] / 1e3, 1), "SearchByBoW(KF, F)": round(t[7] / 1e3, 1)},
            "single_pair_adaptors_equal_batch": bool(r["single_equals_batch"][0] and r["after_drop_equal"][0])}


def _median_us(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts) * 1e6)


def measure(pkg, oracle_py=None, reps=200, cpu_reps=40, nbatch=20, device=0):
    """-> dict for bench.py.  pkg = the loaded orb-slam2_amd package; oracle_py = oracle module (None: no CPU figures)."""
    L = pkg.lib()
    ox = pkg.orbx
    cur, kfs, Fs, eps, sf, sg = make_sets(nkf=nbatch)
    tri_cur = dict(cur); tri_cur["flag"] = (np.arange(len(cur["desc"])) % 5 < 2).astype(np.uint8)     # 40 % already have a MapPoint
    tri_kfs = []
    for k in kfs:
        t = dict(k); t["flag"] = (np.arange(len(k["desc"])) % 5 >= 3).astype(np.uint8); tri_kfs.append(t)
    A, ka = ox.make_featset(kfs[0]); Fr, kf_ = ox.make_featset(cur)
    TA, kta = ox.make_featset(tri_cur); TB, ktb = ox.make_featset(tri_kfs[0])
    n = C.c_int()
    out_f = np.full(Fr.n, -1, np.int32); out_a = np.full(A.n, -1, np.int32)
    cap = TA.n
    pairs = np.zeros((cap, 2), np.int32)
    F0 = np.ascontiguousarray(Fs[0].reshape(9)); ex0, ey0 = eps[0]
    p = ox._p
    res = {"workload": "1000 x 1000 features, 100 vocabulary nodes, one pair per call, host pointers in and out; median of %d calls" % reps,
           "unit": "us per call"}

    def chk(rc):
        if rc != 0:
            raise RuntimeError(L.orbx_last_error().decode())

    gpu = {
        "search_by_bow_kf_f": lambda: chk(L.orbx_search_by_bow_kf_f(device, C.byref(A), C.byref(Fr), 0.7, 1, p(out_f), C.byref(n))),
        "search_by_bow_kf_kf": lambda: chk(L.orbx_search_by_bow_kf_kf(device, C.byref(Fr), C.byref(A), 0.75, 1, p(out_a), C.byref(n))),
        "search_for_triangulation": lambda: chk(L.orbx_search_for_triangulation(device, C.byref(TA), C.byref(TB), p(F0), ex0, ey0, p(sf), p(sg), 8, 0, 0,
                                                                                 p(pairs), cap, C.byref(n))),
    }
    res["gpu"] = {k: round(_median_us(f, reps), 1) for k, f in gpu.items()}
    if hasattr(L, "orbx_debug_match_timing"):
        ph = np.zeros(4)
        res["gpu_host_phases_us"] = {"fields": "prepare, launch, wait, copy-out"}
        for k, f in gpu.items():
            acc = []
            for _ in range(50):
                f(); L.orbx_debug_match_timing(p(ph)); acc.append(ph.copy())
            res["gpu_host_phases_us"][k] = [round(float(v), 1) for v in np.median(np.array(acc), axis=0)]
    nm = {}
    gpu["search_by_bow_kf_f"](); nm["search_by_bow_kf_f"] = n.value
    gpu["search_by_bow_kf_kf"](); nm["search_by_bow_kf_kf"] = n.value
    gpu["search_for_triangulation"](); nm["search_for_triangulation"] = n.value
    res["matches_per_call"] = nm

    # ---- batch forms: the current keyframe against `nbatch` neighbours / candidates in one call
    if hasattr(L, "orbx_search_for_triangulation_batch"):
        sets_t = [ox.make_featset(k) for k in tri_kfs]
        arr_t = (ox.FeatSet * nbatch)(*[s[0] for s in sets_t])
        sets_k = [ox.make_featset(k) for k in kfs]
        arr_k = (ox.FeatSet * nbatch)(*[s[0] for s in sets_k])
        Fall = np.ascontiguousarray(np.stack([f.reshape(9) for f in Fs]).astype(np.float32))
        exy = np.ascontiguousarray(np.array(eps, np.float32))
        bpairs = np.zeros((nbatch, cap, 2), np.int32); bn = np.zeros(nbatch, np.int32)
        bm = np.full((nbatch, Fr.n), -1, np.int32)
        tb = _median_us(lambda: chk(L.orbx_search_for_triangulation_batch(device, C.byref(TA), arr_t, nbatch, p(Fall), p(exy), p(sf), p(sg), 8, 0, 0,
                                                                         p(bpairs), cap, p(bn))), max(reps // 4, 10))
        kb = _median_us(lambda: chk(L.orbx_search_by_bow_kf_kf_batch(device, C.byref(Fr), arr_k, nbatch, 0.75, 1, p(bm), p(bn))), max(reps // 4, 10))
        fb = _median_us(lambda: chk(L.orbx_search_by_bow_kf_f_batch(device, arr_k, nbatch, C.byref(Fr), 0.7, 1, p(bm), p(bn))), max(reps // 4, 10))
        res["gpu_batch"] = {"pairs_per_call": nbatch, "unit": "us per pair",
                            "search_for_triangulation_batch": round(tb / nbatch, 2), "search_by_bow_kf_kf_batch": round(kb / nbatch, 2),
                            "search_by_bow_kf_f_batch": round(fb / nbatch, 2)}

    # ---- resident keyframes (orbx_kf_*): descriptors / FeatureVectors live in HBM, only flags, node intersection and results move
    if hasattr(L, "orbx_kf_create"):
        dcur = pkg.DeviceKeyFrame(tri_cur); dks = [pkg.DeviceKeyFrame(k) for k in tri_kfs]
        hs1 = (C.c_void_p * 1)(dks[0]._h); hsN = (C.c_void_p * nbatch)(*[k._h for k in dks])
        bflags = [np.ascontiguousarray(k["flag"]) for k in kfs]; tflags = [np.ascontiguousarray(k["flag"]) for k in tri_kfs]
        bf1 = (C.c_void_p * 1)(bflags[0].ctypes.data); bfN = (C.c_void_p * nbatch)(*[x.ctypes.data for x in bflags])
        tf1 = (C.c_void_p * 1)(tflags[0].ctypes.data); tfN = (C.c_void_p * nbatch)(*[x.ctypes.data for x in tflags])
        cflag_b = np.ascontiguousarray(cur["flag"]); cflag_t = np.ascontiguousarray(tri_cur["flag"])
        Fall = np.ascontiguousarray(np.stack([f.reshape(9) for f in Fs]).astype(np.float32))
        exy = np.ascontiguousarray(np.array(eps, np.float32))
        bpairs = np.zeros((nbatch, cap, 2), np.int32); bn = np.zeros(nbatch, np.int32); bm = np.full((nbatch, Fr.n), -1, np.int32)
        r1 = {
            "search_by_bow_kf_f": lambda: chk(L.orbx_kf_search_by_bow_kf_f(dks[0]._h, p(bflags[0]), dcur._h, 0.7, 1, p(out_f), C.byref(n))),
            "search_by_bow_kf_kf": lambda: chk(L.orbx_kf_search_by_bow_kf_kf(dcur._h, p(cflag_b), hs1, bf1, 1, 0.75, 1, p(out_a), p(bn))),
            "search_for_triangulation": lambda: chk(L.orbx_kf_search_for_triangulation(dcur._h, p(cflag_t), hs1, tf1, 1, p(F0), p(exy), p(sf), p(sg), 8, 0, 0,
                                                                                       p(pairs), cap, p(bn))),
        }
        res["gpu_resident"] = {k: round(_median_us(f, reps), 1) for k, f in r1.items()}
        if hasattr(L, "orbx_debug_match_timing"):
            ph = np.zeros(4)
            res["gpu_resident_host_phases_us"] = {"fields": "prepare, launch, wait, copy-out"}
            for k, f in r1.items():
                acc = []
                for _ in range(50):
                    f(); L.orbx_debug_match_timing(p(ph)); acc.append(ph.copy())
                res["gpu_resident_host_phases_us"][k] = [round(float(v), 1) for v in np.median(np.array(acc), axis=0)]
        rb = {
            "search_by_bow_kf_kf": lambda: chk(L.orbx_kf_search_by_bow_kf_kf(dcur._h, p(cflag_b), hsN, bfN, nbatch, 0.75, 1, p(bm), p(bn))),
            "search_for_triangulation": lambda: chk(L.orbx_kf_search_for_triangulation(dcur._h, p(cflag_t), hsN, tfN, nbatch, p(Fall), p(exy), p(sf), p(sg), 8, 0, 0,
                                                                                       p(bpairs), cap, p(bn))),
        }
        res["gpu_resident_batch"] = {"pairs_per_call": nbatch, "unit": "us per pair"}
        for k, f in rb.items():
            res["gpu_resident_batch"][k] = round(_median_us(f, max(reps // 4, 10)) / nbatch, 2)
            if hasattr(L, "orbx_debug_match_timing"):
                ph = np.zeros(4); acc = []
                for _ in range(20):
                    f(); L.orbx_debug_match_timing(p(ph)); acc.append(ph.copy())
                res["gpu_resident_batch"][k + "_host_phases_us_per_call"] = [round(float(v), 1) for v in np.median(np.array(acc), axis=0)]

    # ---- the CPU oracle on the same inputs (one thread, as the reference runs each search)
    if oracle_py is not None:
        OL = oracle_py.lib()
        oa, _k1 = oracle_py.make_featset(kfs[0]); of, _k2 = oracle_py.make_featset(cur)
        ota, _k3 = oracle_py.make_featset(tri_cur); otb, _k4 = oracle_py.make_featset(tri_kfs[0])
        o_f = np.full(of.n, -1, np.int32); o_a = np.full(oa.n, -1, np.int32); o_pairs = np.zeros((cap, 2), np.int32)
        cpu = {
            "search_by_bow_kf_f": lambda: OL.oracle_search_by_bow_kf_f(C.byref(oa), C.byref(of), 0.7, 1, p(o_f)),
            "search_by_bow_kf_kf": lambda: OL.oracle_search_by_bow_kf_kf(C.byref(of), C.byref(oa), 0.75, 1, p(o_a)),
            "search_for_triangulation": lambda: OL.oracle_search_for_triangulation(C.byref(ota), C.byref(otb), p(F0), ex0, ey0, p(sf), p(sg), 0.6, 0, 0,
                                                                                   p(o_pairs), cap),
        }
        res["cpu_oracle"] = {k: round(_median_us(f, cpu_reps), 1) for k, f in cpu.items()}
        res["cpu_oracle"]["threads"] = 1
        # the results of the timed calls are the oracle's
        gpu["search_by_bow_kf_f"](); cpu["search_by_bow_kf_f"]()
        same = bool((out_f == o_f).all())
        gpu["search_by_bow_kf_kf"](); cpu["search_by_bow_kf_kf"]()
        same = same and bool((out_a == o_a).all())
        gpu["search_for_triangulation"](); np_g = n.value
        np_o = cpu["search_for_triangulation"]()
        same = same and np_g == np_o and bool((pairs[:np_g] == o_pairs[:np_o]).all())
        res["verified"] = same
        res["gpu_over_cpu"] = {k: round(res["gpu"][k] / res["cpu_oracle"][k], 2) for k in gpu}
    try:
        res["adapter_us"] = adapter_timing()
    except Exception as exc:      # measurement leg only
        res["adapter_us"] = {"error": str(exc)[:300]}
    return res


if __name__ == "__main__":
    import json
    import __graft_entry__ as ge
    from oracle import oracle_py
    print(json.dumps(measure(ge.load_pkg(), oracle_py), indent=1))
