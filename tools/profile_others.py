"""Workload for the rocprofv3 kernel-trace of the kernels OUTSIDE the headline step: the eight projection
searches (k_grid_build / k_proj_lists / k_proj_resolve / k_init_resolve), Frame::ComputeBoW (k_vocab_descend / k_bow_build),
MapPoint::ComputeDistinctiveDescriptors (k_distinctive), EuRoC rectification (k_remap), colour ingest (k_gray) and the three
vocabulary-guided searches in their host-pointer form, each at a representative size and repeated REP times.
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/profile_others.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import test_projection as TP
from tools import synth
pkg = ge.load_pkg()
REP = 20
f32 = np.float32
# ---- projection searches: 1000 points x 1000 features (a tracking frame)
cur, pts, sf = TP._scene(3, 1000, 1000, dense=False)
inv = (1.0 / (sf * sf)).astype(f32)
m = pkg.ORBmatcher(0.9, True)
p2 = dict(pts); p2["aux"] = (pts["u"] - 5).astype(f32)
c1, c2, p12, p21, s2 = TP._sim3_scene(5, 1000)
f1, f2, prev = TP._init_scene(6, 1000)
for _ in range(REP):
    m.SearchByProjectionLastFrame(cur, pts, sf, 15.0, 0, 40.0)
    m.SearchByProjectionMapPoints(cur, p2, sf, 3.0)
    m.SearchByProjectionKeyFrame(cur, pts, sf, 10.0, 100)
    m.SearchByProjectionSim3(cur, pts, sf, 10.0)
    m.Fuse(cur, p2, sf, inv, 3.0, 50)
    m.SearchBySim3(c1, c2, p12, p21, s2, s2, 7.5)
    m.SearchForInitialization(f1, f2, prev, 100)
# ---- ComputeBoW: 1000 descriptors through a k=10, L=6 vocabulary; SearchByBoW / triangulation host forms
img = synth.image(11, 752, 480)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(752, 480))
kp, d = ex(img)
par, leaf, nd, w = synth.vocab_tree(12, 10, 6, stop_frac=0.0, data=None)
rng = np.random.Generator(np.random.PCG64(13))
nd[:110] = synth.flip_bits(rng, d[rng.integers(0, len(d), 110)], 0.1)
voc = pkg.ORBVocabulary(10, 6, par, leaf, nd, w)
t = None
for _ in range(REP):
    t = voc.transform(d, 4)
fs = dict(desc=d, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"], flag=np.ones(len(d), np.uint8), angle=kp["angle"].copy(),
          x=kp["x"].copy(), y=kp["y"].copy(), octave=kp["octave"].copy(), u_right=np.full(len(d), -1, f32))
mm = pkg.ORBmatcher(0.75, True)
F12 = np.array([0, 0, 0, 0, 0, -1 / 500.0, 0, 1 / 500.0, 0], f32)
fs0 = dict(fs); fs0["flag"] = np.zeros(len(d), np.uint8)
for _ in range(REP):
    mm.SearchByBoW(fs, fs)
    kk = dict(fs); kk["kind"] = "keyframe"
    mm.SearchByBoW(fs, kk)
    mm.SearchForTriangulation(fs0, fs0, F12, 400.0, 240.0, np.asarray(ex.GetScaleFactors(), f32), np.asarray(ex.GetScaleSigmaSquares(), f32))
# ---- ComputeDistinctiveDescriptors: 2000 map points x 8 observations
sets = [synth.flip_bits(rng, d[rng.integers(0, len(d), 1)].repeat(8, 0), 0.05) for _ in range(2000)]
for _ in range(REP):
    pkg.ComputeDistinctiveDescriptors(sets)
# ---- remap (EuRoC 752x480 rectification) and colour ingest
yy, xx = np.mgrid[0:480, 0:752].astype(f32)
rect = pkg.Rectifier((752, 480), (xx * 0.98 + 5).astype(f32), (yy * 0.99 + 2).astype(f32))
col = np.repeat(img[:, :, None], 3, 2).copy()
for _ in range(REP):
    ex.extract_rectified(rect, img)
    ex.extract_color(col, rgb=True)
print("profile workload done")
