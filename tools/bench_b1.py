"""One stereo frame per launch set (B = 1), device-resident: the single-camera operating point of ORB-SLAM2
(reference Examples/Stereo/stereo_kitti.cc:68-117).  Prints frames/s; under `rocprofv3 --kernel-trace` the trace shows the
per-frame launch chain (tools/trace_gaps.py).   python tools/bench_b1.py [frames_per_step] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
from tools import synth
pkg = ge.load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
pairs = [synth.stereo_pair(1000 + i, 1241, 376)[:2] for i in range(4)]
rig = bench.StereoRig(pkg, torch, dev, 0, 1241, 376, 1000, B, pairs)
for _ in range(10):
    rig.step()
rig.stream.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    rig.step()
rig.stream.synchronize()
el = time.perf_counter() - t0
print("B=%d: %.1f frames/s, %.1f us per step" % (B, B * steps / el, el / steps * 1e6))
