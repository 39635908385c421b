import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as ge
from tools import synth
from oracle import oracle_py
pkg = ge.load_pkg()
for (w, h, nf, lv) in [(2560, 1440, 4000, 8), (4096, 3072, 8000, 8), (4096, 3072, 4000, 8), (1241, 376, 1000, 12), (640, 480, 500, 4), (1241, 376, 6000, 8)]:
    img = synth.image(123, w, h, nshapes=max(400, w * h // 311))
    ex = pkg.ORBextractor(nf, 1.2, lv, 20, 7, device=0, max_size=(w, h))
    try:
        t = time.time(); k, d = ex(img); tg = time.time() - t
    except pkg.OrbxError as e:
        print(w, h, nf, lv, "rejected:", e, flush=True); continue
    t = time.time(); ok, od = oracle_py.Oracle(nf, 1.2, lv, 20, 7).extract(img); tc = time.time() - t
    print(w, h, nf, lv, len(k), len(ok), k.tobytes() == ok.tobytes(), d.tobytes() == od.tobytes(), "gpu %.3fs cpu %.3fs" % (tg, tc), flush=True)
