"""A/B of kernel variants on ONE box: per-stage milliseconds of the B-frame stereo step for the library named by ORBX_SO (each stage timed
by itself: events only around that stage, tools/trace_gaps finding), and the whole step without events.
python tools/ab_stages.py [B] [steps]      (run once per variant: ORBX_SO=diag/liborbx_ab_<name>.so)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
from tools import synth
pkg = ge.load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
pairs = [synth.stereo_pair(1000 + i, 1241, 376)[:2] for i in range(8)]
rig = bench.StereoRig(pkg, torch, dev, 0, 1241, 376, 1000, B, pairs)
ex = rig.ex
for _ in range(5): rig.step()
rig.stream.synchronize()
out = {}
for st in ("resize", "fast", "tree", "desc", "stereo"):
    ex.profile_read(reset=True)
    ex.profile_stages(1 << pkg.orbx.STAGES.index(st))
    ex.profile_enable(True)
    for _ in range(steps): rig.step()
    rig.stream.synchronize()
    ex.profile_enable(False)
    p = ex.profile_read(reset=True)
    out[st] = p[st][0] / steps
ex.profile_stages(0xFFFFFFFF)
best = 1e9
for rep in range(3):
    rig.stream.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): rig.step()
    rig.stream.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
print("%-34s step %.4f ms (%.0f frames/s)  " % (os.path.basename(os.environ.get("ORBX_SO", "in-tree")), best, B / best * 1e3) +
      "  ".join("%s %.4f" % (k, v) for k, v in out.items()), flush=True)
