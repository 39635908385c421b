// orbx_remap.hip — EuRoC stereo rectification on device (second half):
//   cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR)   reference Examples/Stereo/stereo_euroc.cc:136-137
// with the CV_32FC1 map pair of cv::initUndistortRectifyMap (:103-104; computed once per camera by the reference's
// own host code and handed over as-is).  OpenCV's generic 8-bit path, restated [from memory, parity unpinned]:
//   * the float maps become fixed point once: sx = cvRound(mapx*32), sy = cvRound(mapy*32); integer part
//     saturate_cast<short>(s >> 5), fraction index (sy & 31)*32 + (sx & 31)        (INTER_BITS 5)
//   * weights w = (32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy, each * 32  (BilinearTab_i, INTER_REMAP_COEF_BITS 15;
//     the products are exact so the table's sum fix-up never fires; entry (0,0) saturates to 32767 and the fix-up
//     moves the missing 1 to the fourth tap, which cannot change an 8-bit result -- tests/test_remap.py checks the
//     arithmetic weights used here against the oracle's literal table)
//   * D = saturate_cast<uchar>((S00*w0 + S01*w1 + S10*w2 + S11*w3 + (1<<14)) >> 15), BORDER_CONSTANT 0: a tap outside
//     the source reads 0 (remapBilinear's border branch).
// HBM-bound: per output pixel 6 B of fixed-point map + 1 B written + the gathered source bytes (read once through L2).
#include "orbx_device.h"

struct orbx_rectifier {
    int device;
    int src_w, src_h, dst_w, dst_h, map_pitch; // map_pitch in pixels (multiple of 4)
    uint32_t *d_xy;   // [dst_h][map_pitch] (short x | short y << 16)
    uint16_t *d_a;    // [dst_h][map_pitch] fraction index
};

__global__ __launch_bounds__(256) void k_map_convert(const float *__restrict__ mx, const float *__restrict__ my, int w, int h, int mp,
                                                     uint32_t *__restrict__ xy, uint16_t *__restrict__ a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h || x >= mp) return;
    uint32_t o = 0x80008000u; // padding columns point far outside: they produce the border value
    uint16_t f = 0;
    if (x < w) {
        const int sx = dev_cv_round(mx[(long long)y * w + x] * 32.0f), sy = dev_cv_round(my[(long long)y * w + x] * 32.0f);
        int ix = sx >> 5, iy = sy >> 5;
        ix = ix < -32768 ? -32768 : ix > 32767 ? 32767 : ix; // saturate_cast<short>
        iy = iy < -32768 ? -32768 : iy > 32767 ? 32767 : iy;
        o = ((uint32_t)ix & 0xFFFFu) | ((uint32_t)iy << 16);
        f = (uint16_t)((sy & 31) * 32 + (sx & 31));
    }
    xy[(long long)y * mp + x] = o;
    a[(long long)y * mp + x] = f;
}

// 4 output pixels per thread (one dword store); grid z = image of the batch
__global__ __launch_bounds__(256) void k_remap(const uint8_t *__restrict__ src, long long src_stride, int spitch, int sw, int sh,
                                               const uint32_t *__restrict__ xy, const uint16_t *__restrict__ a, int mp, int dw, int dh,
                                               uint8_t *__restrict__ dst, long long dst_stride, int dpitch)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= dh || x4 >= dpitch || x4 >= mp) return;
    const uint8_t *S = src + (long long)blockIdx.z * src_stride;
    const uint4 q = *reinterpret_cast<const uint4 *>(xy + (long long)y * mp + x4);
    const uint2 fa = *reinterpret_cast<const uint2 *>(a + (long long)y * mp + x4);
    const uint32_t qq[4] = { q.x, q.y, q.z, q.w };
    const uint32_t ff[4] = { fa.x & 0xFFFFu, fa.x >> 16, fa.y & 0xFFFFu, fa.y >> 16 };
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sx = (int)(short)(qq[i] & 0xFFFFu), sy = (int)(short)(qq[i] >> 16);
        const int fx = ff[i] & 31, fy = ff[i] >> 5;
        const int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
        int v = 0;
        if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
            const uint8_t *p = S + (long long)sy * spitch + sx;
            v = (p[0] * w0 + p[1] * w1 + p[spitch] * w2 + p[spitch + 1] * w3 + (1 << 14)) >> 15;
        } else if (!(sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
            const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw;
            const bool y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
            const int v0 = (x0 && y0) ? S[(long long)sy * spitch + sx] : 0, v1 = (x1 && y0) ? S[(long long)sy * spitch + sx + 1] : 0;
            const int v2 = (x0 && y1) ? S[(long long)(sy + 1) * spitch + sx] : 0, v3 = (x1 && y1) ? S[(long long)(sy + 1) * spitch + sx + 1] : 0;
            v = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
        }
        out |= (uint32_t)(v > 255 ? 255 : v) << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(dst + (long long)blockIdx.z * dst_stride + (long long)y * dpitch + x4) = out;
}

extern "C" int orbx_rectifier_create(int device, int src_w, int src_h, int dst_w, int dst_h, const float *map_x, const float *map_y,
                                     orbx_rectifier **out)
{
    if (!out || !map_x || !map_y || src_w < 1 || src_h < 1 || dst_w < 1 || dst_h < 1 || src_w > 32767 || src_h > 32767) {
        orbx_set_error("orbx_rectifier_create: invalid argument");
        return ORBX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_rectifier *r = new orbx_rectifier();
    r->device = device; r->src_w = src_w; r->src_h = src_h; r->dst_w = dst_w; r->dst_h = dst_h;
    r->map_pitch = (dst_w + 63) & ~63;
    const size_t npx = (size_t)r->map_pitch * dst_h, nmap = (size_t)dst_w * dst_h;
    float *d_mx = nullptr, *d_my = nullptr;
    hipError_t he = hipMalloc((void **)&r->d_xy, npx * 4);
    if (he == hipSuccess) he = hipMalloc((void **)&r->d_a, npx * 2);
    if (he == hipSuccess) he = hipMalloc((void **)&d_mx, nmap * 4);
    if (he == hipSuccess) he = hipMalloc((void **)&d_my, nmap * 4);
    if (he == hipSuccess) he = hipMemcpy(d_mx, map_x, nmap * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_my, map_y, nmap * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(k_map_convert, dim3(r->map_pitch / 64, (dst_h + 3) / 4), dim3(256), 0, 0, d_mx, d_my, dst_w, dst_h, r->map_pitch,
                           r->d_xy, r->d_a);
        he = hipDeviceSynchronize();
    }
    if (d_mx) hipFree(d_mx);
    if (d_my) hipFree(d_my);
    if (he != hipSuccess) {
        orbx_set_error("orbx_rectifier_create: %s", hipGetErrorString(he));
        if (r->d_xy) hipFree(r->d_xy);
        if (r->d_a) hipFree(r->d_a);
        delete r;
        return ORBX_E_HIP;
    }
    *out = r;
    return ORBX_OK;
}

extern "C" void orbx_rectifier_destroy(orbx_rectifier *r)
{
    if (!r) return;
    hipSetDevice(r->device);
    if (r->d_xy) hipFree(r->d_xy);
    if (r->d_a) hipFree(r->d_a);
    delete r;
}

extern "C" int orbx_rectifier_size(const orbx_rectifier *r, int *dst_w, int *dst_h)
{
    if (!r) { orbx_set_error("null rectifier"); return ORBX_E_INVALID; }
    if (dst_w) *dst_w = r->dst_w;
    if (dst_h) *dst_h = r->dst_h;
    return ORBX_OK;
}

extern "C" int orbx_remap_batch_device(const orbx_rectifier *r, const void *d_src, size_t src_stride, size_t src_pitch, int batch,
                                       void *d_dst, size_t dst_stride, size_t dst_pitch, void *stream)
{
    if (!r || !d_src || !d_dst || batch < 1 || src_pitch < (size_t)r->src_w || dst_pitch < (size_t)r->dst_w || (dst_pitch & 3) ||
        ((uintptr_t)d_dst & 3) || (dst_stride & 3) || (batch > 1 && (src_stride < src_pitch * r->src_h || dst_stride < dst_pitch * r->dst_h))) {
        orbx_set_error("orbx_remap_batch_device: invalid argument (dst pointer, pitch and stride must be multiples of 4)");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(hipSetDevice(r->device));
    const int cols4 = ((r->dst_w + 3) / 4 + 63) / 64;
    hipLaunchKernelGGL(k_remap, dim3(cols4, (r->dst_h + 3) / 4, batch), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_src,
                       (long long)src_stride, (int)src_pitch, r->src_w, r->src_h, r->d_xy, r->d_a, r->map_pitch, r->dst_w, r->dst_h,
                       (uint8_t *)d_dst, (long long)dst_stride, (int)dst_pitch);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}
