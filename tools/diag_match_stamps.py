"""Phase stamps (s_memrealtime, 100 MHz) of the wave that finalises a per-call matcher search: where the kernel's ~20 us go.
Needs the diagnostic build:  python -c "import __graft_entry__ as g; g.build_diag()"  then  python tools/diag_match_stamps.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBX_SO"] = os.path.join(ROOT, "diag", "liborbx_diag.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from tools import matcher_bench as mb  # noqa: E402

pkg = ge.load_pkg()
L = pkg.lib()
ox = pkg.orbx
cur, kfs, Fs, eps, sf, sg = mb.make_sets(nkf=20)
names = ["item+pair read", "tile staged", "node done", "arrival atomic", "histogram pass", "results issued", "system fence", "ticket"]
m = pkg.ORBmatcher(0.7, True)
dcur = pkg.DeviceKeyFrame(cur); dk = pkg.DeviceKeyFrame(kfs[0])
F = [Fs[0]]; E = [eps[0]]
calls = {
    "bow_kf_f host pointers": lambda: m.SearchByBoW(kfs[0], cur),
    "bow_kf_f resident": lambda: m.SearchByBoWResident(dk, kfs[0]["flag"], dcur),
    "triangulation host pointers": lambda: m.SearchForTriangulation(cur, kfs[0], Fs[0], eps[0][0], eps[0][1], sf, sg),
    "triangulation resident": lambda: m.SearchForTriangulationResident(dcur, cur["flag"], [dk], [kfs[0]["flag"]], F, E, sf, sg),
}
dks = [pkg.DeviceKeyFrame(k) for k in kfs]
calls["triangulation resident batch of 20 (pair 0's finaliser)"] = lambda: m.SearchForTriangulationResident(dcur, cur["flag"], dks, [k["flag"] for k in kfs], Fs, eps, sf, sg)
calls["bow kf_kf resident batch of 20 (pair 0's finaliser)"] = lambda: m.SearchByBoWKeyFramesResident(dcur, cur["flag"], dks, [k["flag"] for k in kfs])
st = (C.c_ulonglong * 16)()
for name, fn in calls.items():
    acc = []
    for it in range(60):
        fn()
        L.orbx_diag_match_stamps(st)
        v = np.array(st[:9], np.float64)
        if it >= 10:
            acc.append(np.diff(v) * 0.01)      # 100 MHz ticks -> microseconds
    med = np.median(np.array(acc), axis=0)
    print("%-58s total %5.1f us | " % (name, med.sum()) + "  ".join("%s %.1f" % (n, x) for n, x in zip(names, med)))
