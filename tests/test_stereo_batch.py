"""GPU parity of the BATCHED stereo path exactly as bench.py drives it (the path the headline metric times):
one handle, orbx_extract_batch_device on [L0..L(B-1), R0..R(B-1)], then
orbx_stereo_match_batch_device(ex, 0, ex, B, B, ...) -- per-pair indexing of keypoint / descriptor / row-table /
entry / cut buffers at batch > 1 -- against the CPU oracle pair by pair, bit for bit
(reference src/Frame.cc:82-97 and :577-751).  Also the two-handle form with non-zero first images."""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu

W, H = 1241, 376
BF, MIN_Z = 386.1448, 386.1448 / 718.856     # Examples/Stereo/KITTI00-02.yaml:8,25


def _pairs(n, seed0):
    """n distinct KITTI-shape pairs; pair 3 is featureless (both eyes flat), pair 5 has a right eye and pair 6 a left
    eye that is flat outside a 120x100 window (about 640 keypoints instead of the quota: unequal counts inside one batch)"""
    out = []
    for i in range(n):
        l, r, _ = synth.stereo_pair(seed0 + i, W, H)
        if i == 3:
            l = np.full((H, W), 77, np.uint8); r = np.full((H, W), 77, np.uint8)
        if i == 5:
            r2 = np.full_like(r, 60); r2[100:200, 500:620] = r[100:200, 500:620]; r = r2
        if i == 6:
            l2 = np.full_like(l, 200); l2[100:200, 500:620] = l[100:200, 500:620]; l = l2
        out.append((l, r))
    return out


def _oracle_results(oracle, pairs, nf):
    res = []
    for l, r in pairs:
        oL, oR = oracle.Oracle(nf, 1.2, 8, 20, 7), oracle.Oracle(nf, 1.2, 8, 20, 7)
        kL, dL = oL.extract(l); kR, dR = oR.extract(r)
        ur, dp = oracle.stereo_match(oL, oR, kL, dL, kR, dR, BF, MIN_Z)
        res.append((kL, dL, kR, dR, ur, dp))
    return res


def _compare(tag, p, exp, n_l, n_r, k_l, d_l, k_r, d_r, ur, dp):
    kL, dL, kR, dR, our, odp = exp
    assert n_l == len(kL) and n_r == len(kR), f"{tag} pair {p}: keypoint counts ({n_l},{n_r}) vs oracle ({len(kL)},{len(kR)})"
    assert k_l[:n_l].tobytes() == kL.tobytes() and d_l[:n_l].tobytes() == dL.tobytes(), f"{tag} pair {p}: left eye differs"
    assert k_r[:n_r].tobytes() == kR.tobytes() and d_r[:n_r].tobytes() == dR.tobytes(), f"{tag} pair {p}: right eye differs"
    bad = np.nonzero(ur[:n_l].view(np.uint32) != our.view(np.uint32))[0]
    assert len(bad) == 0, f"{tag} pair {p}: uRight differs at {bad[:5].tolist()}: {ur[bad[:5]]} vs {our[bad[:5]]}"
    assert dp[:n_l].tobytes() == odp.tobytes(), f"{tag} pair {p}: depth differs"


@pytest.mark.parametrize("nf,B", [(1000, 16), (2000, 16)])
def test_stereo_match_batch_one_handle(pkg, oracle, nf, B):
    """BASELINE configs M / 2 / 5 at batch 16, the call sequence of bench.py:step()"""
    import torch
    pairs = _pairs(B, 300 + nf)
    exp = _oracle_results(oracle, pairs, nf)
    pitch = (W + 63) // 64 * 64
    host = np.zeros((2 * B, H, pitch), np.uint8)
    for i, (l, r) in enumerate(pairs):
        host[i, :, :W] = l
        host[B + i, :, :W] = r
    dev = torch.device("cuda", 0)
    imgs = torch.from_numpy(host).to(dev)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, device=0, max_size=(W, H), max_batch=2 * B)
    cap = ex.max_keypoints(W, H)
    kps = torch.zeros((2 * B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device=dev)
    nout = torch.zeros(2 * B, dtype=torch.int32, device=dev)
    ur = torch.full((B, cap), 123.0, dtype=torch.float32, device=dev)   # poisoned: every entry < n must be rewritten
    dp = torch.full((B, cap), 123.0, dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    for rep in range(2):   # twice: the second run starts from the first run's workspaces (entries, cut state, row tables)
        ex.extract_batch_device(imgs.data_ptr(), H * pitch, pitch, 2 * B, W, H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), sp)
        pkg.orbx.stereo_match_batch_device(ex, 0, ex, B, B, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(),
                                           kps[B:].data_ptr(), desc[B:].data_ptr(), nout[B:].data_ptr(), cap,
                                           BF, MIN_Z, ur.data_ptr(), dp.data_ptr(), sp, row_table=pkg.orbx.ROWTAB_OF_EXTRACTION)
        ex.sync(sp)
        n_h = nout.cpu().numpy()
        k_h = kps.cpu().numpy().view(np.uint8).reshape(2 * B, cap, 28)
        d_h = desc.cpu().numpy(); ur_h = ur.cpu().numpy(); dp_h = dp.cpu().numpy()
        for p in range(B):
            _compare(f"nf={nf} rep={rep}", p, exp[p], int(n_h[p]), int(n_h[B + p]), k_h[p], d_h[p], k_h[B + p], d_h[B + p], ur_h[p], dp_h[p])
    matched = [(ur_h[p, :n_h[p]] >= 0).sum() for p in range(B)]
    assert matched[3] == 0 and n_h[3] == 0                       # the featureless pair
    assert n_h[B + 5] < 0.8 * n_h[5] and n_h[6] < 0.8 * n_h[B + 6]   # unequal counts really occur
    assert sum(m > 50 for m in matched) >= B - 4                 # the test is not vacuous


def test_stereo_match_batch_two_handles_offset(pkg, oracle):
    """left and right eyes in different handles, first images != 0 (imgL0 = 2, imgR0 = 1)"""
    import torch
    nf, B = 1000, 5
    pairs = _pairs(B, 900)
    exp = _oracle_results(oracle, pairs, nf)
    dev = torch.device("cuda", 0)
    NL, NR = B + 2, B + 1
    hl = np.zeros((NL, H, W), np.uint8); hr = np.zeros((NR, H, W), np.uint8)
    hl[:2] = synth.image(5, W, H); hr[:1] = synth.image(6, W, H)        # unrelated images in front
    for i, (l, r) in enumerate(pairs):
        hl[2 + i] = l; hr[1 + i] = r
    dl, dr = torch.from_numpy(hl).to(dev), torch.from_numpy(hr).to(dev)
    exL = pkg.ORBextractor(nf, 1.2, 8, 20, 7, device=0, max_size=(W, H), max_batch=NL)
    exR = pkg.ORBextractor(nf, 1.2, 8, 20, 7, device=0, max_size=(W, H), max_batch=NR)
    cap = exL.max_keypoints(W, H)

    def bufs(n):
        return (torch.zeros((n, cap, 7), dtype=torch.float32, device=dev), torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev),
                torch.zeros(n, dtype=torch.int32, device=dev))
    kl, dsl, nl = bufs(NL); kr, dsr, nr = bufs(NR)
    ur = torch.zeros((B, cap), dtype=torch.float32, device=dev); dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    exL.extract_batch_device(dl.data_ptr(), H * W, W, NL, W, H, kl.data_ptr(), dsl.data_ptr(), cap, nl.data_ptr(), None)
    exL.sync()
    exR.extract_batch_device(dr.data_ptr(), H * W, W, NR, W, H, kr.data_ptr(), dsr.data_ptr(), cap, nr.data_ptr(), None)
    exR.sync()
    pkg.orbx.stereo_match_batch_device(exL, 2, exR, 1, B, kl[2:].data_ptr(), dsl[2:].data_ptr(), nl[2:].data_ptr(),
                                       kr[1:].data_ptr(), dsr[1:].data_ptr(), nr[1:].data_ptr(), cap,
                                       BF, MIN_Z, ur.data_ptr(), dp.data_ptr(), None, row_table=pkg.orbx.ROWTAB_OF_EXTRACTION)
    exL.sync()
    nl_h, nr_h = nl.cpu().numpy(), nr.cpu().numpy()
    kl_h = kl.cpu().numpy().view(np.uint8).reshape(NL, cap, 28); kr_h = kr.cpu().numpy().view(np.uint8).reshape(NR, cap, 28)
    dl_h, dr_h, ur_h, dp_h = dsl.cpu().numpy(), dsr.cpu().numpy(), ur.cpu().numpy(), dp.cpu().numpy()
    for p in range(B):
        _compare("two handles", p, exp[p], int(nl_h[2 + p]), int(nr_h[1 + p]), kl_h[2 + p], dl_h[2 + p], kr_h[1 + p], dr_h[1 + p], ur_h[p], dp_h[p])


def test_host_fed_batches_equal_oracle(pkg, oracle):
    """bench.py's batched host-fed path (HostFedPipeline: pinned host images -> H2D on a copy stream under the previous batch's kernels -> extraction +
    stereo match -> D2H of all results, four buffer sets going round): every frame of several consecutive batches with DIFFERENT contents, read from
    the pinned result blocks, equals the oracle byte for byte -- a batch is neither computed from a half-uploaded image block nor overwritten
    before it has been downloaded"""
    import torch
    import bench
    W_, H_, NF, B = 640, 480, 800, 6
    dev = torch.device("cuda", 0)
    pipe = bench.HostFedPipeline(pkg, torch, dev, 0, W_, H_, NF, B, depth=4)
    nb = 9                                              # more batches than buffer sets: every set is reused at least once
    pairs = [synth.stereo_pair(700 + i, W_, H_)[:2] for i in range(5)]
    oL, oR = oracle.Oracle(NF, 1.2, 8, 20, 7), oracle.Oracle(NF, 1.2, 8, 20, 7)
    exp = []
    for l, r in pairs:
        kL, dL = oL.extract(l); kR, dR = oR.extract(r)
        ur, dp = oracle.stereo_match(oL, oR, kL, dL, kR, dR, BF, MIN_Z)
        exp.append((kL, dL, kR, dR, ur, dp))
    pending = []
    checked = 0

    def check(j, which):
        n, k, d, ur, dp = pipe.results(j)
        for i, pi in enumerate(which):
            kL, dL, kR, dR, our, odp = exp[pi]
            nl, nr = int(n[i]), int(n[B + i])
            assert nl == len(kL) and nr == len(kR), (which, i)
            assert k[i, :nl].tobytes() == kL.tobytes() and d[i, :nl].tobytes() == dL.tobytes()
            assert k[B + i, :nr].tobytes() == kR.tobytes() and d[B + i, :nr].tobytes() == dR.tobytes()
            assert ur[i, :nl].tobytes() == our.tobytes() and dp[i, :nl].tobytes() == odp.tobytes()
    for b in range(nb):
        if len(pending) == pipe.depth:                  # the buffer set about to be refilled: collect its results first
            j, which = pending.pop(0)
            check(j, which); checked += 1
        which = [(3 * b + i) % len(pairs) for i in range(B)]          # every batch a different mix of the pairs
        hi = pipe.host_images(pipe.k % pipe.depth).numpy()
        for i, pi in enumerate(which):
            hi[i, :, :W_] = pairs[pi][0]; hi[B + i, :, :W_] = pairs[pi][1]
        pending.append((pipe.submit(), which))
    for j, which in pending:
        check(j, which); checked += 1
    assert checked == nb
