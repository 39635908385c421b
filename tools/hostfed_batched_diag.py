"""Where the batched host-fed rate goes: the same 64-frame batches (a) computed from device-resident images, (b) uploaded only, (c) downloaded only,
(d) the full pipeline of bench.host_fed_batched.   python tools/hostfed_batched_diag.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
from tools import synth
pkg = ge.load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
pairs = [synth.stereo_pair(1000 + i, 1241, 376)[:2] for i in range(8)]
rigs = [bench.StereoRig(pkg, torch, dev, 0, 1241, 376, 1000, B, pairs) for _ in range(2)]
h_in = [torch.from_numpy(r.imgs.cpu().numpy()).pin_memory() for r in rigs]
d_in = [torch.empty_like(r.imgs) for r in rigs]
def rate(fn, n=steps):
    for _ in range(4): fn(0); fn(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i % 2)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return B * n / el
print("compute only (images resident), two handles / streams: %.0f frames/s" % rate(lambda j: rigs[j].step()))
cin = torch.cuda.Stream()
def up(j):
    with torch.cuda.stream(cin): d_in[j].copy_(h_in[j], non_blocking=True)
r = rate(up); print("H2D only: %.0f frames/s = %.1f GB/s" % (r, r * h_in[0].numel() / B / 1e9))
h_out = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in (r_.nout, r_.kps, r_.desc, r_.ur, r_.dp)] for r_ in rigs]
cout = torch.cuda.Stream()
def down(j):
    with torch.cuda.stream(cout):
        for ht, dt in zip(h_out[j], (rigs[j].nout, rigs[j].kps, rigs[j].desc, rigs[j].ur, rigs[j].dp)): ht.copy_(dt, non_blocking=True)
r = rate(down); ob = sum(t.numel() * t.element_size() for t in h_out[0]); print("D2H only: %.0f frames/s = %.1f GB/s (%d B per frame)" % (r, r * ob / B / 1e9, ob // B))
def updown(j): up(j); down(j)
print("H2D + D2H together: %.0f frames/s" % rate(updown))
def upcompute(j):
    up(j); rigs[j].step()
print("H2D + compute (no dependencies): %.0f frames/s" % rate(upcompute))
print("full pipeline:", bench.host_fed_batched(pkg, torch, dev, 0, pairs, B, steps))
# does anything in a step block the host?  enqueue 20 steps on each rig and time the enqueue itself
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): rigs[i % 2].step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue of 20 steps: %.2f ms on the host, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
t0 = time.perf_counter()
for i in range(20): up(i % 2)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue of 20 uploads: %.2f ms on the host, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
t0 = time.perf_counter()
for i in range(20): down(i % 2)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue of 20 downloads: %.2f ms on the host, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
def all_nodeps(j): up(j); rigs[j].step(); down(j)
print("H2D + compute + D2H, no dependencies: %.0f frames/s" % rate(all_nodeps))
# dependencies only where the data flows: upload -> compute -> download per rig (events), nothing across rigs
ev_up = [torch.cuda.Event() for _ in rigs]; ev_cp = [torch.cuda.Event() for _ in rigs]
def chained(j):
    up(j); ev_up[j].record(cin)
    rigs[j].stream.wait_event(ev_up[j]); rigs[j].step(d_in[j]); ev_cp[j].record(rigs[j].stream)
    cout.wait_event(ev_cp[j]); down(j)
print("upload -> compute -> download chained by events (no back-pressure waits): %.0f frames/s" % rate(chained))
