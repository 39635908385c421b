"""Latency of orbx_search_by_projection_last_frame (1000 points x 1000 features, uniform and clustered) next to
the CPU oracle.  Run on the GPU box: python tools/bench_projection_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
from test_projection import _scene
from oracle import oracle_py as O
pkg = ge.load_pkg()
for dense in (False, True):
    cur, pts, sf = _scene(3, 1000, 1000, dense=dense)
    mt = pkg.ORBmatcher(0.9, True)
    for _ in range(5): mt.SearchByProjectionLastFrame(cur, pts, sf, 15.0, 0, 40.0)
    ts=[]
    for _ in range(100):
        t=time.perf_counter(); got,n=mt.SearchByProjectionLastFrame(cur, pts, sf, 15.0, 0, 40.0); ts.append(time.perf_counter()-t)
    t=time.perf_counter()
    for _ in range(20): O.search_by_projection_last(cur, pts, sf, 15.0, 0, 40.0, True)
    tc=(time.perf_counter()-t)/20
    print("dense" if dense else "uniform", "GPU host-inclusive median %.1f us (incl. python marshalling); CPU oracle %.1f us; matches %d" % (np.median(ts)*1e6, tc*1e6, n))
