// orbx_frame.hip — Frame::UndistortKeyPoints (reference src/Frame.cc:470-515; SURVEY.md 8f row f2, "plus" part).
// cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on the N keypoint positions: per point, in double,
//   x = (u - cx)/fx, y = (v - cy)/fy (as multiplications by ifx = 1./fx, ify = 1./fy), five fixed-point iterations of
//   the inverse Brown-Conrady model (k1 k2 p1 p2 k3), then back through P = mK, result rounded to float
// [OpenCV 3.2 cvUndistortPoints restated from memory -- parity unpinned; 2.4.11's loop is the same arithmetic for the
// 4/5-coefficient models ORB-SLAM2's settings files carry: the rational, thin-prism and tilt terms are exact no-ops at
// zero].  Points are independent: one thread each, fp64 VALU (-ffp-contract=off keeps the oracle's rounding).
#include "orbx_device.h"
#include <string.h>
#include <vector>

struct UndistortParams { double fx, fy, ifx, ify, cx, cy, k[5]; };

__global__ __launch_bounds__(256) void k_undistort(const float2 *__restrict__ in, float2 *__restrict__ out, int n, UndistortParams p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 s = in[i];
    double x = ((double)s.x - p.cx) * p.ifx, y = ((double)s.y - p.cy) * p.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2);
        const double dx = 2 * p.k[2] * x * y + p.k[3] * (r2 + 2 * x * x);
        const double dy = p.k[2] * (r2 + 2 * y * y) + 2 * p.k[3] * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    float2 o;
    o.x = (float)(p.fx * x + p.cx);
    o.y = (float)(p.fy * y + p.cy);
    out[i] = o;
}

struct FrameCtx { hipStream_t stream = nullptr; float2 *d_in = nullptr, *d_out = nullptr; float2 *h = nullptr; size_t cap = 0; };
static thread_local FrameCtx g_frame[16];

extern "C" int orbx_undistort_keypoints(int device, const float *xy, int n, float fx, float fy, float cx, float cy,
                                        const float *dist_coef, int ndist, float *xy_out)
{
    if (n < 0 || (n && (!xy || !xy_out)) || !dist_coef || (ndist != 4 && ndist != 5) || fx == 0.f || fy == 0.f) {
        orbx_set_error("orbx_undistort_keypoints: invalid argument (4 or 5 distortion coefficients)");
        return ORBX_E_INVALID;
    }
    if (n == 0) return ORBX_OK;
    if (dist_coef[0] == 0.0f) { // src/Frame.cc:472-476: mvKeysUn = mvKeys
        if (xy_out != xy) memcpy(xy_out, xy, sizeof(float) * 2 * (size_t)n);
        return ORBX_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 16) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    FrameCtx *c = &g_frame[device];
    if (!c->stream) ORBX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if ((size_t)n > c->cap) {
        if (c->d_in) ORBX_HIP(hipFree(c->d_in));
        if (c->d_out) ORBX_HIP(hipFree(c->d_out));
        if (c->h) ORBX_HIP(hipHostFree(c->h));
        c->d_in = c->d_out = c->h = nullptr;
        const size_t cap = (size_t)n * 2;
        ORBX_HIP(hipMalloc((void **)&c->d_in, sizeof(float2) * cap));
        ORBX_HIP(hipMalloc((void **)&c->d_out, sizeof(float2) * cap));
        ORBX_HIP(hipHostMalloc((void **)&c->h, sizeof(float2) * cap, hipHostMallocDefault));
        c->cap = cap;
    }
    UndistortParams p;
    p.fx = fx; p.fy = fy; p.ifx = 1. / p.fx; p.ify = 1. / p.fy; p.cx = cx; p.cy = cy;
    for (int i = 0; i < 5; i++) p.k[i] = i < ndist ? (double)dist_coef[i] : 0.0;
    memcpy(c->h, xy, sizeof(float2) * (size_t)n);
    ORBX_HIP(hipMemcpyAsync(c->d_in, c->h, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_in, c->d_out, n, p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(c->h, c->d_out, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    ORBX_HIP(hipStreamSynchronize(c->stream));
    memcpy(xy_out, c->h, sizeof(float2) * (size_t)n);
    return ORBX_OK;
}
