"""Randomized parity soak of the BATCH sizes between a single frame and the big batches: the host picks each kernel's form by launch size
(grouped / mid / per-level pyramid, 1-4 waves per FAST cell, 256- / 1024-thread quadtree, one / four keypoints per wave and folded / separate
median cut in the stereo matcher), so batches of 1 .. 40 images of random geometry are extracted (and matched as stereo pairs) in one call
and every image compared with the CPU oracle.   Run on the GPU box: python tools/soak_batch.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from tools import synth
from oracle import oracle_py as O
pkg = ge.load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 4711))
t0 = time.time(); trial = n_img = n_pair = 0
bf, min_z = 386.1448, 0.5372
while time.time() - t0 < budget:
    trial += 1
    w = int(rng.integers(200, 1300)); h = int(rng.integers(150, 700))
    nf = int(rng.integers(100, 2500)); nlevels = int(rng.choice([4, 6, 8, 8, 8])); sf = float(rng.choice([1.2, 1.2, 1.2, 1.3, 1.5]))
    B = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 13, 16, 24, 25, 32, 40]))      # pairs
    try:
        oL, oR = O.Oracle(nf, sf, nlevels, 20, 7), O.Oracle(nf, sf, nlevels, 20, 7)
        pairs = [synth.stereo_pair(9000 + 50 * trial + i, w, h, nshapes=int(w * h / int(rng.integers(200, 700))) + 30)[:2] for i in range(min(B, 4))]
        exp = []
        for l, r in pairs:
            kL, dL = oL.extract(l); kR, dR = oR.extract(r)
            exp.append((kL, dL, kR, dR) + tuple(O.stereo_match(oL, oR, kL, dL, kR, dR, bf, min_z)))
    except RuntimeError:
        continue
    tag = f"trial {trial}: {w}x{h} nf={nf} L={nlevels} sf={sf} pairs={B}"
    ex = pkg.ORBextractor(nf, sf, nlevels, 20, 7, device=0, max_size=(w, h), max_batch=2 * B)
    pitch = (w + 63) // 64 * 64
    host = np.zeros((2 * B, h, pitch), np.uint8)
    for i in range(B):
        host[i, :, :w] = pairs[i % len(pairs)][0]; host[B + i, :, :w] = pairs[i % len(pairs)][1]
    imgs = torch.from_numpy(host).cuda()
    cap = ex.max_keypoints(w, h)
    kps = torch.zeros((2 * B, cap, 7), device="cuda"); desc = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(2 * B, dtype=torch.int32, device="cuda"); ur = torch.zeros((B, cap), device="cuda"); dp = torch.zeros((B, cap), device="cuda")
    ex.extract_batch_device(imgs.data_ptr(), h * pitch, pitch, 2 * B, w, h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), None)
    pkg.orbx.stereo_match_batch_device(ex, 0, ex, B, B, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), kps[B:].data_ptr(), desc[B:].data_ptr(), n[B:].data_ptr(),
                                       cap, bf, min_z, ur.data_ptr(), dp.data_ptr(), None,
                                       row_table=pkg.orbx.ROWTAB_OF_EXTRACTION if pkg.orbx.stereo_row_table_available(ex, kps[B:].data_ptr(), B, B, cap) and B % 2 == 0
                                       else pkg.orbx.ROWTAB_FROM_KEYPOINTS)
    ex.sync()
    nn = n.cpu().numpy(); kk = kps.cpu().numpy().view(np.uint8).reshape(2 * B, cap, 28); dd = desc.cpu().numpy(); uu = ur.cpu().numpy(); zz = dp.cpu().numpy()
    for i in range(B):
        e = exp[i % len(pairs)]
        for eye, (ek, ed) in enumerate(((e[0], e[1]), (e[2], e[3]))):
            j = i + eye * B
            assert nn[j] == len(ek), f"{tag}: image {j}: {nn[j]} keypoints vs oracle {len(ek)}"
            assert kk[j, :nn[j]].tobytes() == ek.tobytes() and dd[j, :nn[j]].tobytes() == ed.tobytes(), f"{tag}: image {j} differs"
        assert uu[i, :nn[i]].tobytes() == e[4].tobytes() and zz[i, :nn[i]].tobytes() == e[5].tobytes(), f"{tag}: stereo pair {i} differs"
    n_img += 2 * B; n_pair += B
    del ex, imgs, kps, desc
    if trial % 10 == 0:
        print(f"{time.time() - t0:6.1f}s trials {trial} images {n_img} pairs {n_pair}", flush=True)
print(f"batch soak done: {trial} trials, {n_img} images and {n_pair} stereo pairs in batches of 1..40 pairs, all equal to the oracle")
