"""Latency of the host-pointer entry points (orbx_extract, orbx_stereo_match) on one 1241x376 frame.
Run on the GPU box: python tools/bench_host_latency.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from tools import synth
pkg = ge.load_pkg(); L = pkg.lib()
W,H = 1241,376
img = synth.image(5,W,H)
ex = pkg.ORBextractor(1000,1.2,8,20,7,device=0,max_size=(W,H))
cap = ex.max_keypoints(W,H)
kps = np.zeros(cap, pkg.KP_DTYPE); desc = np.zeros((cap,32),np.uint8); n = C.c_int()
def call():
    rc = L.orbx_extract(ex._h, img.ctypes.data, W, H, img.strides[0], kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)); assert rc == 0
for _ in range(20): call()
ts=[]
for _ in range(300):
    t=time.perf_counter(); call(); ts.append(time.perf_counter()-t)
ts=np.array(ts)*1e6
print("orbx_extract host API: median %.1f us  mean %.1f us  p95 %.1f us  n=%d" % (np.median(ts), ts.mean(), np.percentile(ts,95), n.value))
# stereo host path: extract L, extract R (sequential) + ComputeStereoMatches
L_,R_,_ = synth.stereo_pair(9,W,H)
exL = pkg.ORBextractor(1000,1.2,8,20,7,device=0,max_size=(W,H)); exR = pkg.ORBextractor(1000,1.2,8,20,7,device=0,max_size=(W,H))
kL,dL = exL(L_); kR,dR = exR(R_)
ur = np.zeros(len(kL),np.float32); dp = np.zeros(len(kL),np.float32)
def st():
    rc = L.orbx_stereo_match(exL._h, exR._h, kL.ctypes.data, dL.ctypes.data, len(kL), kR.ctypes.data, dR.ctypes.data, len(kR), C.c_float(386.1448), C.c_float(0.5372), ur.ctypes.data, dp.ctypes.data); assert rc==0
for _ in range(10): st()
ts=[]
for _ in range(200):
    t=time.perf_counter(); st(); ts.append(time.perf_counter()-t)
ts=np.array(ts)*1e6
print("orbx_stereo_match host API: median %.1f us  p95 %.1f us  matches %d" % (np.median(ts), np.percentile(ts,95), (ur>=0).sum()))
# one-call stereo frame
ex2 = pkg.ORBextractor(1000,1.2,8,20,7,device=0,max_size=(W,H),max_batch=2)
cap2 = ex2.max_keypoints(W,H)
k2 = np.zeros((2,cap2), pkg.KP_DTYPE); d2 = np.zeros((2,cap2,32),np.uint8); n2 = np.zeros(2,np.int32); u2 = np.zeros(cap2,np.float32); z2 = np.zeros(cap2,np.float32)
def st1():
    rc = L.orbx_extract_stereo(ex2._h, L_.ctypes.data, R_.ctypes.data, W, H, L_.strides[0], C.c_float(386.1448), C.c_float(0.5372), k2.ctypes.data, d2.ctypes.data, cap2, n2.ctypes.data, u2.ctypes.data, z2.ctypes.data); assert rc == 0
for _ in range(20): st1()
ts=[]
for _ in range(300):
    t=time.perf_counter(); st1(); ts.append(time.perf_counter()-t)
ts=np.array(ts)*1e6
print("orbx_extract_stereo (one call per stereo frame): median %.1f us  p95 %.1f us  kpL %d kpR %d matches %d" % (np.median(ts), np.percentile(ts,95), n2[0], n2[1], (u2[:n2[0]]>=0).sum()))
