"""Wall-clock span of the waves of ONE k_fast / k_desc launch (diagnostic build, s_memrealtime 100 MHz): first wave start -> last wave
end, the longest wave, the mean wave.  Compare with the kernel's duration in a rocprofv3 trace: the difference is launch overhead,
the longest wave is the floor of the launch.   python tools/diag_spans.py [frames per launch]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ORBX_SO"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diag", "liborbx_spans.so")   # g.build_diag(("-DORBX_DIAG", "-DORBX_DIAG_SPANS_ONLY"), "liborbx_spans.so")
import numpy as np, torch
import __graft_entry__ as ge
from tools import synth
pkg = ge.load_pkg(); L = pkg.lib()
W, H, B = 1241, 376, (int(sys.argv[1]) if len(sys.argv) > 1 else 1)
pairs = [synth.stereo_pair(1000 + i, W, H)[:2] for i in range(4)]
pitch = 1280; host = np.zeros((2 * B, H, pitch), np.uint8)
for i in range(B): host[i, :, :W] = pairs[i % 4][0]; host[B + i, :, :W] = pairs[i % 4][1]
imgs = torch.from_numpy(host).cuda()
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(W, H), max_batch=2 * B)
cap = ex.max_keypoints(W, H)
kps = torch.zeros((2 * B, cap, 7), device='cuda'); desc = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(2 * B, dtype=torch.int32, device='cuda')
def step(): ex.extract_batch_device(imgs.data_ptr(), H * pitch, pitch, 2 * B, W, H, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), None)
for _ in range(3): step()
ex.sync()
SL = 16384
out = (C.c_uint32 * (2 * SL * 2))()
L.orbx_diag_spans(out, 1)
N = 20
res = {0: [], 1: []}
worst = {0: [], 1: []}
for _ in range(N):
    step(); ex.sync()
    L.orbx_diag_spans(out, 1)
    o = np.frombuffer(out, dtype=np.uint32).reshape(2, SL, 2).astype(np.int64)
    for k in range(2):
        v = o[k][o[k, :, 1] != 0]
        life = (v[:, 1] - v[:, 0]) / 100.0
        res[k].append([(v[:, 1].max() - v[:, 0].min()) / 100.0, life.max(), life.mean(), np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99), len(v),
                       (v[:, 0].max() - v[:, 0].min()) / 100.0])
        ids = np.nonzero(o[k, :, 1] != 0)[0]
        worst[k].append(ids[np.argsort(-life)[:5]].tolist())
for k, nm in enumerate(("k_fast", "k_desc")):
    r = np.array(res[k]).mean(axis=0)
    print(f"{nm}: first start -> last end {r[0]:.1f} us (last start {r[7]:.1f} us after the first); wave life: longest {r[1]:.1f}, mean {r[2]:.1f}, p50 {r[3]:.1f}, p90 {r[4]:.1f}, p99 {r[5]:.1f} us; waves {r[6]:.0f}")
    print("   slowest workgroup ids of the last launches:", worst[k][-3:])

# phases of the slowest k_fast waves of the last launch (end of: load+zero, pretest, fullscore, nms, emit), and their cells' candidate counts
ph = (C.c_uint32 * (16384 * 8))()
L.orbx_diag_fast_phases(ph)
ph = np.frombuffer(ph, dtype=np.uint32).reshape(16384, 8).astype(np.int64)
o = np.frombuffer(out, dtype=np.uint32).reshape(2, SL, 2).astype(np.int64)
ids = np.nonzero(o[0, :, 1] != 0)[0]
life = (o[0, ids, 1] - o[0, ids, 0]) / 100.0
order = ids[np.argsort(-life)]
def show(i):
    t0 = o[0, i, 0]
    d = [(ph[i, k] - (ph[i, k - 1] if k else t0)) / 100.0 for k in range(5)]
    extra = ""
    if ph[i, 5] and ph[i, 6]:    # one-round cells: list built at [5], wave 0's score loop done at [6], everybody's at [2]
        extra = " (score: list %.1f, loop of wave 0 %.1f, wait %.1f)" % ((ph[i, 5] - ph[i, 1]) / 100.0, (ph[i, 6] - ph[i, 5]) / 100.0, (ph[i, 2] - ph[i, 6]) / 100.0)
    return "id %5d life %5.1f us: load %.1f pretest %.1f score %.1f nms %.1f emit %.1f" % (i, (o[0, i, 1] - t0) / 100.0, *d) + extra
print("slowest k_fast waves:")
for i in order[:8]: print("  ", show(i))
print("median-ish waves:")
for i in order[len(order) // 2: len(order) // 2 + 3]: print("  ", show(i))
# mean phase durations over the logged waves of the last launch (those that ran all phases)
full = [i for i in ids if ph[i, 4] and ph[i, 0] and ph[i, 5] and ph[i, 6]]
if full:
    f = np.array(full)
    t0 = o[0, f, 0]
    seq = [("load+zero", ph[f, 0] - t0), ("pretest", ph[f, 1] - ph[f, 0]), ("list", ph[f, 5] - ph[f, 1]), ("score", ph[f, 6] - ph[f, 5]), ("barrier", ph[f, 2] - ph[f, 6]),
           ("nms", ph[f, 3] - ph[f, 2]), ("emit", ph[f, 4] - ph[f, 3])]
    print("k_fast form", ex.debug_fast_form(), "mean phase durations over", len(f), "waves (us):", {k: round(float(v.mean()) / 100.0, 2) for k, v in seq},
          "life", round(float((o[0, f, 1] - t0).mean()) / 100.0, 2))
