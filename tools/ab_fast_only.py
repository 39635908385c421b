"""k_fast's time per B-frame launch for the library named by ORBX_SO (events around that stage only).  python tools/ab_fast_only.py [B] [steps]"""
import os, sys
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
res = []
for rep in range(2):
    ex.profile_read(reset=True)
    ex.profile_stages(1 << pkg.orbx.STAGES.index("fast"))
    ex.profile_enable(True)
    for _ in range(steps): rig.step()
    rig.stream.synchronize()
    ex.profile_enable(False)
    res.append(ex.profile_read(reset=True)["fast"][0] / steps)
print("%-28s k_fast %.4f %.4f ms" % (os.path.basename(os.environ.get("ORBX_SO", "in-tree")), res[0], res[1]), flush=True)
