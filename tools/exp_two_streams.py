"""experiment: one handle x 256 frames against two handles x 128 frames on two streams (same total work per step)"""
import sys, time, importlib
sys.path.insert(0, ".")
import numpy as np, torch
import bench
import __graft_entry__ as ge
pkg = ge.load_pkg()
from tools import synth
dev = torch.device("cuda:0")
pairs = [synth.stereo_pair(2 + i, bench.W, bench.H)[:2] for i in range(8)]
def run(rigs, steps=20):
    for _ in range(3):
        for r in rigs: r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for r in rigs: r.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return sum(r.B for r in rigs) * steps / el
for split in (1, 2, 4):
    rigs = [bench.StereoRig(pkg, torch, dev, 0, bench.W, bench.H, bench.NFEAT, 256 // split, pairs) for _ in range(split)]
    print(split, "handles:", round(run(rigs), 1), "frames/s", flush=True)
    del rigs
