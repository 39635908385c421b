"""Do the launch chains of independent frames (or batches) overlap on the chip?  N device-resident rigs of B frames (own handle, own
stream), steps submitted round-robin from one thread: aggregate frames/s against one rig alone.   python tools/bench_b1_lanes.py [N] [steps] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
from tools import synth
pkg = ge.load_pkg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # frames per rig and step
dev = torch.device("cuda", 0)
pairs = [synth.stereo_pair(1000 + i, 1241, 376)[:2] for i in range(8)]
rigs = [bench.StereoRig(pkg, torch, dev, 0, 1241, 376, 1000, B, pairs) for _ in range(N)]
for r in rigs:
    for _ in range(10): r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for r in rigs: r.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("%d lanes x %d frames: %.1f frames/s in total, %.1f us per step of a lane" % (N, B, N * B * steps / el, el / steps * 1e6))
