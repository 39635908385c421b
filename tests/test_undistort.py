"""Frame::UndistortKeyPoints (reference src/Frame.cc:470-515): cv::undistortPoints with P = K.  CPU: the oracle against a
numpy fp64 restatement and against the defining property (re-applying the forward distortion model returns the input).
GPU: bit-exact floats against the oracle."""
import numpy as np
import pytest

# TUM1.yaml-like camera: fx fy cx cy, k1 k2 p1 p2 k3
CAMS = [(517.306408, 516.469215, 318.643040, 255.313989, [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]),
        (535.4, 539.2, 320.1, 247.6, [-0.1, 0.05, 0.001, -0.002]),
        (458.654, 457.296, 367.215, 248.375, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])]


def _pts(seed, n, w=640, h=480):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1).astype(np.float32)


def _np_undistort(xy, fx, fy, cx, cy, dist):
    f = np.float32
    k = np.zeros(5); k[:len(dist)] = np.asarray(dist, f).astype(np.float64)
    fx, fy, cx, cy = [np.float64(f(v)) for v in (fx, fy, cx, cy)]
    x = (xy[:, 0].astype(np.float64) - cx) * (1. / fx); y = (xy[:, 1].astype(np.float64) - cy) * (1. / fy)
    x0, y0 = x.copy(), y.copy()
    for _ in range(5):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x); dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (x0 - dx) * ic; y = (y0 - dy) * ic
    return np.stack([(fx * x + cx).astype(f), (fy * y + cy).astype(f)], 1), x, y


@pytest.mark.parametrize("cam", range(len(CAMS)))
def test_oracle_undistort(oracle, cam):
    fx, fy, cx, cy, d = CAMS[cam]
    xy = _pts(cam, 2000)
    got = oracle.undistort_points(xy, fx, fy, cx, cy, d)
    exp, xn, yn = _np_undistort(xy, fx, fy, cx, cy, d)
    assert got.tobytes() == exp.tobytes()
    # forward model on the undistorted normalised point gives back the distorted pixel (5 iterations: loose near the corners)
    k = np.zeros(5); k[:len(d)] = d
    r2 = xn * xn + yn * yn
    rad = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
    xd = xn * rad + 2 * k[2] * xn * yn + k[3] * (r2 + 2 * xn * xn); yd = yn * rad + k[2] * (r2 + 2 * yn * yn) + 2 * k[3] * xn * yn
    back = np.stack([xd * fx + cx, yd * fy + cy], 1)
    centre = np.hypot(xy[:, 0] - cx, xy[:, 1] - cy) < 200
    assert np.abs(back - xy)[centre].max() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("cam", range(len(CAMS)))
def test_hip_undistort(pkg, oracle, cam):
    fx, fy, cx, cy, d = CAMS[cam]
    xy = _pts(10 + cam, 5000)
    got = pkg.UndistortKeyPoints(xy, fx, fy, cx, cy, d)
    exp = oracle.undistort_points(xy, fx, fy, cx, cy, d)
    assert got.tobytes() == exp.tobytes()
    corners = np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32)   # Frame::ComputeImageBounds (:517-552)
    assert pkg.UndistortKeyPoints(corners, fx, fy, cx, cy, d).tobytes() == oracle.undistort_points(corners, fx, fy, cx, cy, d).tobytes()


@pytest.mark.gpu
def test_hip_undistort_identity_and_errors(pkg):
    xy = _pts(20, 100)
    assert (pkg.UndistortKeyPoints(xy, 500.0, 500.0, 320.0, 240.0, [0, 0, 0, 0]) == xy).all()   # mDistCoef[0] == 0: copy (:472-476)
    assert len(pkg.UndistortKeyPoints(np.zeros((0, 2), np.float32), 500.0, 500.0, 320.0, 240.0, [0.1, 0, 0, 0])) == 0
    with pytest.raises(pkg.OrbxError):
        pkg.UndistortKeyPoints(xy, 500.0, 500.0, 320.0, 240.0, [0.1, 0.0])
