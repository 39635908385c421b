"""SURVEY.md 8f row f4 (second half): cv::remap INTER_LINEAR / CV_32FC1 maps / 8UC1 / BORDER_CONSTANT 0, the EuRoC
rectification of reference Examples/Stereo/stereo_euroc.cc:136-137.  CPU: the oracle (OpenCV's literal weight table,
restated from memory -> parity unpinned) against an independent numpy restatement with arithmetic weights.
GPU: the device kernel against the oracle, and rectify+extract against oracle-rectify + oracle-extract."""
import numpy as np
import pytest

from tools import synth

f32 = np.float32


def _maps(seed, sw, sh, dw, dh, kind="euroc"):
    """a plausible undistort+rectify map pair: small rotation, radial term, shift; some of it leaves the source"""
    rng = np.random.Generator(np.random.PCG64(seed))
    x, y = np.meshgrid(np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64))
    if kind == "identity":
        return x.astype(f32), y.astype(f32)
    if kind == "half":   # every fraction index on a regular lattice, integer and half positions included
        return (x * (sw / dw) - 0.5).astype(f32), (y * (sh / dh) + 0.25).astype(f32)
    cx, cy = dw / 2 + rng.uniform(-5, 5), dh / 2 + rng.uniform(-5, 5)
    th = rng.uniform(-0.02, 0.02)
    xn, yn = (x - cx) / 400.0, (y - cy) / 400.0
    r2 = xn * xn + yn * yn
    k = 1 + rng.uniform(-0.25, -0.05) * r2 + 0.07 * r2 * r2
    xr = np.cos(th) * xn - np.sin(th) * yn; yr = np.sin(th) * xn + np.cos(th) * yn
    z = 400 * 1.12 * max(sw / dw, sh / dh)   # zoomed out a little: the corners and one edge leave the source
    return (xr * k * z + sw / 2 + rng.uniform(-8, 8)).astype(f32), (yr * k * z + sh / 2 + rng.uniform(-8, 8)).astype(f32)


def _np_remap(img, mx, my):
    sh, sw = img.shape
    fsx = np.rint((mx * f32(32)).astype(f32)).astype(np.int64); fsy = np.rint((my * f32(32)).astype(f32)).astype(np.int64)
    sx = np.clip(fsx >> 5, -32768, 32767); sy = np.clip(fsy >> 5, -32768, 32767)
    fx = fsx & 31; fy = fsy & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    pad = np.zeros((sh + 2, sw + 2), np.int64); pad[1:-1, 1:-1] = img
    def tap(dx, dy):
        xx = sx + dx; yy = sy + dy
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok, pad[np.clip(yy, -1, sh) + 1, np.clip(xx, -1, sw) + 1], 0)
    val = tap(0, 0) * w[0] + tap(1, 0) * w[1] + tap(0, 1) * w[2] + tap(1, 1) * w[3]
    return np.clip((val + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind,seed", [("euroc", 1), ("euroc", 2), ("identity", 3), ("half", 4)])
def test_oracle_remap_vs_numpy(oracle, kind, seed):
    img = synth.image(40 + seed, 376, 240)
    mx, my = _maps(seed, 376, 240, 360, 232, kind)
    got = oracle.remap_bilinear(img, mx, my)
    exp = _np_remap(img, mx, my)
    assert (got == exp).all(), np.argwhere(got != exp)[:5]
    if kind == "identity":
        assert (got == img[:232, :360]).all()
    if kind == "euroc":
        assert (got == 0).any() and (got != 0).mean() > 0.8   # part of the map leaves the source: border value


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed,size", [("euroc", 5, (752, 480, 752, 480)), ("euroc", 6, (752, 480, 700, 433)), ("half", 7, (333, 217, 301, 199)),
                                            ("identity", 8, (752, 480, 752, 480))])
def test_hip_remap_parity(pkg, oracle, kind, seed, size):
    import torch
    sw, sh, dw, dh = size
    mx, my = _maps(seed, sw, sh, dw, dh, kind)
    rect = pkg.Rectifier((sw, sh), mx, my)
    B = 3
    imgs = np.stack([synth.image(60 + seed + i, sw, sh) for i in range(B)])
    spitch = (sw + 63) // 64 * 64; dpitch = (dw + 63) // 64 * 64
    host = np.zeros((B, sh, spitch), np.uint8); host[:, :, :sw] = imgs
    d_src = torch.from_numpy(host).cuda(); d_dst = torch.full((B, dh, dpitch), 77, dtype=torch.uint8, device="cuda")
    rect.remap_batch_device(d_src.data_ptr(), sh * spitch, spitch, B, d_dst.data_ptr(), dh * dpitch, dpitch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = d_dst.cpu().numpy()
    for i in range(B):
        exp = oracle.remap_bilinear(imgs[i], mx, my)
        assert (out[i, :, :dw] == exp).all(), (i, np.argwhere(out[i, :, :dw] != exp)[:5])


@pytest.mark.gpu
def test_hip_extract_rectified(pkg, oracle):
    sw, sh = 752, 480
    mx, my = _maps(9, sw, sh, sw, sh)
    rect = pkg.Rectifier((sw, sh), mx, my)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(sw, sh))
    img = synth.image(70, sw, sh)
    kps, desc, r = ex.extract_rectified(rect, img, want_rect=True)
    er = oracle.remap_bilinear(img, mx, my)
    assert (r == er).all()
    okps, odesc = oracle.Oracle(1000, 1.2, 8, 20, 7).extract(er)
    assert len(kps) == len(okps) > 500 and kps.tobytes() == okps.tobytes() and desc.tobytes() == odesc.tobytes()
    with pytest.raises(pkg.OrbxError):
        pkg.Rectifier((sw, sh), mx, my[:-1])
