#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/liborb_oracle.so).

The reference holds no golden vectors (it has no tests) and cannot be built here (OpenCV absent),
so these fixtures pin the ORACLE's behaviour ("parity unpinned" against upstream):
they catch regressions of the oracle and give the GPU tests a second, oracle-independent-at-run-time
comparison target.  Inputs are stored with the outputs.  Run: python tools/gen_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tools import synth            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def extract_case(name, img, nf, nlevels=8):
    o = O.Oracle(nf, 1.2, nlevels, 20, 7)
    k, d = o.extract(img)
    crc = np.array([zlib.crc32(o.level(l).tobytes()) for l in range(nlevels)], np.uint32)
    dims = np.array([o.level(l).shape[::-1] for l in range(nlevels)], np.int32)
    ncand = np.array([len(o.candidates(l)[0]) for l in range(nlevels)], np.int32)
    cand_crc = np.array([zlib.crc32(np.stack(o.candidates(l)).astype(np.int32).tobytes()) for l in range(nlevels)], np.uint32)
    nkp = np.array([o.nkeypoints(l) for l in range(nlevels)], np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, nfeatures=nf, nlevels=nlevels, kps=k, desc=d,
                        level_dims=dims, level_crc=crc, ncand=ncand, cand_crc=cand_crc, nkp=nkp)
    return o, k, d


def next_rows():
    """the projection-type searches (f1), vocabulary transform (f2), distinctive descriptor (f3),
    colour / rectification / undistortion front end (f4, f2) on small seeded inputs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_projection as TP
    import test_remap as TR
    f32 = np.float32
    out = {}
    cur, pts, sf = TP._scene(201, 600, 500, dense=True)
    inv_s2 = (1.0 / (sf * sf)).astype(f32)
    for k_, v in cur.items():
        out["cur_" + k_] = np.asarray(v)
    for k_, v in pts.items():
        out["pts_" + k_] = np.asarray(v)
    out["sf"] = sf; out["inv_s2"] = inv_s2
    for d in (0, 1, 2):
        m, n = O.search_by_projection_last(cur, pts, sf, 7.0, d, 40.0, True)
        out[f"last_{d}_m"] = m; out[f"last_{d}_n"] = n
    p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
    m, n = O.search_by_projection_points(cur, p2, sf, 3.0, 0.8); out["points_m"] = m; out["points_n"] = n
    m, n = O.search_by_projection_keyframe(cur, pts, sf, 10.0, 100, True); out["kf_m"] = m; out["kf_n"] = n
    m, n = O.search_by_projection_sim3(cur, pts, sf, 10.0); out["sim3p_m"] = m; out["sim3p_n"] = n
    p3 = dict(pts); p3["aux"] = (pts["u"] - 8).astype(f32)
    for chi2 in (0, 1):
        bi, bd, n = O.window_best(cur, p3, sf, inv_s2, 4.0, chi2, 50)
        out[f"best_{chi2}_idx"] = bi; out[f"best_{chi2}_dist"] = bd; out[f"best_{chi2}_n"] = n
    c1, c2, p12, p21, sf2 = TP._sim3_scene(202, 400)
    for nm, dct in (("s1", c1), ("s2", c2), ("p12", p12), ("p21", p21)):
        for k_, v in dct.items():
            out[f"{nm}_{k_}"] = np.asarray(v)
    m, n = O.search_by_sim3(c1, c2, p12, p21, sf2, sf2, 7.5); out["sim3_m"] = m; out["sim3_n"] = n
    f1, f2, prev = TP._init_scene(203, 500)
    for nm, dct in (("i1", f1), ("i2", f2)):
        for k_, v in dct.items():
            out[f"{nm}_{k_}"] = np.asarray(v)
    out["init_prev"] = prev
    m, n = O.search_for_initialization(f1, f2, prev, 100, 0.9, True); out["init_m"] = m; out["init_n"] = n
    np.savez_compressed(os.path.join(OUT, "projection_s201.npz"), **out)
    # f2 / f3 / f4
    rng = np.random.Generator(np.random.PCG64(204))
    desc = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    par, leaf, nd, w = synth.vocab_tree(205, 10, 3, stop_frac=0.05, data=desc)
    t = O.Vocabulary(10, 3, par, leaf, nd, w).transform(desc, 1)
    obs = [synth.flip_bits(rng, np.repeat(desc[i:i + 1], 3 + i % 9, axis=0), 0.08) for i in range(40)]
    img = synth.image(206, 200, 150, nshapes=200)
    rgb = np.stack([img, np.roll(img, 3, 1), np.roll(img, 5, 0)], 2)
    mx, my = TR._maps(207, 200, 150, 192, 144)
    xy = np.stack([rng.uniform(0, 640, 200), rng.uniform(0, 480, 200)], 1).astype(f32)
    cam = (517.306408, 516.469215, 318.643040, 255.313989, np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], f32))
    np.savez_compressed(os.path.join(OUT, "frontend_s204.npz"), voc_parent=par, voc_leaf=leaf, voc_desc=nd, voc_weight=w, desc=desc,
                        **{"bow_" + k_: v for k_, v in t.items()},
                        obs_flat=np.concatenate(obs), obs_off=np.cumsum([0] + [len(o) for o in obs]).astype(np.int32),
                        obs_best=np.array([O.distinctive_descriptor(o) for o in obs], np.int32),
                        rgb=rgb, gray_rgb=O.cvt_gray(rgb, 1), gray_bgr=O.cvt_gray(rgb, 0),
                        img=img, map_x=mx, map_y=my, remapped=O.remap_bilinear(img, mx, my),
                        und_xy=xy, und_cam=np.array(cam[:4], f32), und_dist=cam[4], und_out=O.undistort_points(xy, *cam[:4], cam[4]))


def main():
    os.makedirs(OUT, exist_ok=True)
    extract_case("extract_320x240_s101", synth.image(101, 320, 240, nshapes=500), 500)
    extract_case("extract_640x480_s102", synth.image(102, 640, 480), 1000)
    extract_case("extract_160x120_l4_s103", synth.image(103, 160, 120, nshapes=300), 300, nlevels=4)
    # stereo
    left, right, disp = synth.stereo_pair(104, 640, 480)
    oL, oR = O.Oracle(1000, 1.2, 8, 20, 7), O.Oracle(1000, 1.2, 8, 20, 7)
    kL, dL = oL.extract(left); kR, dR = oR.extract(right)
    bf, b = 386.1448, 386.1448 / 718.856
    ur, dp = O.stereo_match(oL, oR, kL, dL, kR, dR, bf, b)
    np.savez_compressed(os.path.join(OUT, "stereo_640x480_s104.npz"), left=left, right=right, disp=disp, bf=bf, b=b,
                        kL=kL, dL=dL, kR=kR, dR=dR, u_right=ur, depth=dp)
    # BoW searches on the stereo left features
    rng = np.random.Generator(np.random.PCG64(105))
    voc = synth.Vocab2(106); voc.seed_from(dL, rng)
    perm = rng.permutation(len(dL))
    d2 = synth.flip_bits(rng, dL, 0.05)[perm]; k2 = kL[perm]
    def mk(desc, kp, flag):
        ids, off, feat = voc.feature_vector(desc)
        return dict(desc=desc, node_id=ids, node_off=off, feat=feat, flag=flag, angle=kp["angle"], x=kp["x"], y=kp["y"],
                    octave=kp["octave"], u_right=np.full(len(kp), -1, np.float32))
    kf = mk(d2, k2, (rng.random(len(d2)) < 0.6).astype(np.uint8))
    fr = mk(dL, kL, (rng.random(len(dL)) < 0.6).astype(np.uint8))
    m_kf_f, n1 = O.search_by_bow_kf_f(kf, fr, 0.75, True)
    m_kf_kf, n2 = O.search_by_bow_kf_kf(kf, fr, 0.75, True)
    F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + np.float32(2e-7) * np.arange(9, dtype=np.float32).reshape(3, 3))
    kf_t = dict(kf); kf_t["flag"] = (np.arange(len(d2)) % 3 == 0).astype(np.uint8)
    fr_t = dict(fr); fr_t["flag"] = (np.arange(len(dL)) % 5 == 0).astype(np.uint8)
    pairs = O.search_for_triangulation(kf_t, fr_t, F12, 300.0, 200.0, oL.scale_factors(), oL.level_sigma2(), 0.6, False, False)
    flat = {}
    for nm, s in (("kf", kf), ("fr", fr)):
        for k_, v in s.items():
            flat[f"{nm}_{k_}"] = v
    np.savez_compressed(os.path.join(OUT, "bow_s105.npz"), **flat, tri_flag_kf=kf_t["flag"], tri_flag_fr=fr_t["flag"], F12=F12,
                        sf=oL.scale_factors(), sig2=oL.level_sigma2(), m_kf_f=m_kf_f, n_kf_f=n1, m_kf_kf=m_kf_kf, n_kf_kf=n2, tri_pairs=pairs)
    next_rows()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
