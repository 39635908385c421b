// orbx_device.h — wave64 / workgroup helpers and the exact-arithmetic device primitives.
#pragma once
#include "orbx_internal.h"

#define WAVE 64

// ---- arrays that take part in a RELAXED PUBLISH PROTOCOL (k_stereo's folded median cut, k_match's last-arriver finalisation).
// Producers on any XCD write their results as relaxed agent-scope atomic stores (gfx942 / gfx950: `global_store ... sc1`, a write-through
// that leaves no dirty line in the producer's L2), wait for the acknowledgements (`publish_drain`: s_waitcnt vmcnt(0)) and only then bump a
// relaxed agent-scope arrival counter; the workgroup whose bump completes the count reads the array with relaxed agent-scope atomic loads
// (sc1: served from the memory side, never from a stale line of its own XCD's L2).  That replaces a release fence per producer
// (`__threadfence()` = buffer_wbl2: 17-27 us of a 2000-item launch, profiles/r03_match_stamps.txt) and is correct ONLY while EVERY access to
// such an array inside the launch is an agent-scope atomic: one plain load may hit a stale L2 line, one plain store may sit dirty in an L2 the
// consumer never sees.  `Published<T>` therefore has no operator[] and hands out no pointer: a plain access of these arrays does not
// compile.  A launch whose consumer is a LATER launch (kernel boundaries write back and invalidate) may use `plain_across_launches()`.
// The reasoning is about this memory system, so the build is refused for any other target.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liborbx relies on gfx950 cache semantics (sc1 write-through stores in the relaxed publish protocols): build with --offload-arch=gfx950 only"
#endif
template <typename T>
class Published {
    T *p_;
public:
    Published() = default;
    __host__ __device__ explicit Published(T *p) : p_(p) {}
    __device__ __forceinline__ void put(long long i, T v) const { __hip_atomic_store(p_ + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ T get(long long i) const { return __hip_atomic_load(p_ + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __host__ __device__ __forceinline__ Published at(long long off) const { return Published(p_ + off); }
    // only for a launch that does not run the protocol (its results are consumed by a later launch)
    __device__ __forceinline__ T *plain_across_launches() const { return p_; }
};
// a producer's stores have been acknowledged by the memory side: what it publishes next (the arrival bump) is ordered behind them
__device__ __forceinline__ void publish_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the arrival bump itself; returns the number of producers that had arrived before this one
__device__ __forceinline__ int publish_arrive(int *counter) { return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// forward declaration of the DPP helper macro used below
#define ORBX_DPP(v, identity, ctrl, row_mask, bank_mask) \
    __builtin_amdgcn_update_dpp((int)(identity), (int)(v), ctrl, row_mask, bank_mask, false)

// inclusive wave64 prefix sum on the DPP path: Kogge-Stone inside each 16-lane row (row_shr 1, 2, 4, 8), then the
// row totals ripple with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += ORBX_DPP(v, 0, 0x111, 0xf, 0xf);
    v += ORBX_DPP(v, 0, 0x112, 0xf, 0xf);
    v += ORBX_DPP(v, 0, 0x114, 0xf, 0xf);
    v += ORBX_DPP(v, 0, 0x118, 0xf, 0xf);
    v += ORBX_DPP(v, 0, 0x142, 0xa, 0xf);
    v += ORBX_DPP(v, 0, 0x143, 0xc, 0xf);
    return v;
}

// inclusive wave64 prefix OR, same network
__device__ __forceinline__ unsigned wave_incl_scan_or(unsigned v)
{
    v |= (unsigned)ORBX_DPP(v, 0, 0x111, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, 0, 0x112, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, 0, 0x114, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, 0, 0x118, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, 0, 0x142, 0xa, 0xf);
    v |= (unsigned)ORBX_DPP(v, 0, 0x143, 0xc, 0xf);
    return v;
}

// Wave64 reductions on the DPP data path (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast 15 / 31
// across rows; the total lands in lane 63 and is broadcast with v_readlane): 6 VALU + DPP steps instead of the six
// dependent ds_bpermute round trips that __shfl_xor lowers to.  Lanes without a source read `identity`.

__device__ __forceinline__ int wave_sum(int v)
{
    v += ORBX_DPP(v, 0, 0x111, 0xf, 0xf); // row_shr:1
    v += ORBX_DPP(v, 0, 0x112, 0xf, 0xf); // row_shr:2
    v += ORBX_DPP(v, 0, 0x114, 0xf, 0xe); // row_shr:4
    v += ORBX_DPP(v, 0, 0x118, 0xf, 0xc); // row_shr:8
    v += ORBX_DPP(v, 0, 0x142, 0xa, 0xf); // row_bcast:15
    v += ORBX_DPP(v, 0, 0x143, 0xc, 0xf); // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    const unsigned id = 0xFFFFFFFFu;
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x111, 0xf, 0xf));
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x112, 0xf, 0xf));
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x114, 0xf, 0xe));
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x118, 0xf, 0xc));
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x142, 0xa, 0xf));
    v = min(v, (unsigned)ORBX_DPP(v, id, 0x143, 0xc, 0xf));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// reductions inside each 16-lane row (a quarter wave), result in every lane of the row: butterfly of row rotations
__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
    v = min(v, (unsigned)ORBX_DPP(v, v, 0x128, 0xf, 0xf)); // row_ror:8
    v = min(v, (unsigned)ORBX_DPP(v, v, 0x124, 0xf, 0xf)); // row_ror:4
    v = min(v, (unsigned)ORBX_DPP(v, v, 0x122, 0xf, 0xf)); // row_ror:2
    v = min(v, (unsigned)ORBX_DPP(v, v, 0x121, 0xf, 0xf)); // row_ror:1
    return v;
}
__device__ __forceinline__ unsigned row16_or_u32(unsigned v)
{
    v |= (unsigned)ORBX_DPP(v, v, 0x128, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, v, 0x124, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, v, 0x122, 0xf, 0xf);
    v |= (unsigned)ORBX_DPP(v, v, 0x121, 0xf, 0xf);
    return v;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x111, 0xf, 0xf));
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x112, 0xf, 0xf));
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x114, 0xf, 0xe));
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x118, 0xf, 0xc));
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x142, 0xa, 0xf));
    v = max(v, (unsigned)ORBX_DPP(v, 0, 0x143, 0xc, 0xf));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// exclusive scan of one value per thread over an NT-thread workgroup (NT = 256 or 1024); s_w = NT / 64 ints of LDS
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *s_w)
{
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));        // (opaque: &s_w[wv] is not kept in a register between calls -- in k_bow2 it went to scratch)
    const int lane = t_ & (WAVE - 1), wv = t_ >> 6;
    const int inc = wave_incl_scan(v);
    if (lane == WAVE - 1) s_w[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / WAVE; i++) { const int t = s_w[i]; if (i < wv) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
__device__ __forceinline__ int block_excl_scan256(int v, int *total, int *s_w) { return block_excl_scan<256>(v, total, s_w); }

// in-place exclusive scan of an LDS array a[0..m) by an NT-thread workgroup; returns the total.
// Callers must __syncthreads() before (inputs complete); outputs are visible on return.
template <int NT>
__device__ __forceinline__ int lds_excl_scan_nt(int *a, int m, int *s_w)
{
    const int per = (m + NT - 1) / NT;
    const int beg = threadIdx.x * per;
    const int end = beg + per < m ? beg + per : m;
    int s = 0;
    for (int i = beg; i < end; i++) s += a[i];
    int tot;
    int run = block_excl_scan<NT>(s, &tot, s_w);
    for (int i = beg; i < end; i++) { int t = a[i]; a[i] = run; run += t; }
    __syncthreads();
    return tot;
}
__device__ __forceinline__ int lds_excl_scan(int *a, int m, int *s_w) { return lds_excl_scan_nt<256>(a, m, s_w); }

// floor(i / d) for 0 <= i < 2^16, 1 <= d < 2^16 by multiply-high (d == 1 would overflow the magic)
struct FastDiv {
    unsigned magic; bool one;
    __device__ __forceinline__ explicit FastDiv(int d) : magic(0xFFFFFFFFu / (unsigned)d + 1u), one(d == 1) {}
    __device__ __forceinline__ int div(int i) const { return one ? i : (int)__umulhi((unsigned)i, magic); }
};

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// cv::fastAtan2 (OpenCV 3.x polynomial): fp32, one rounding per operation
// (the translation unit is compiled with -ffp-contract=off).
__device__ __forceinline__ float dev_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    // one division and one polynomial for both octant cases: min / (max + eps) is the quotient either branch of the
    // reference forms (ax >= ay: ay / (ax + eps), else ax / (ay + eps)); same operations on the same values
    const float ax = fabsf(x), ay = fabsf(y);
    const bool steep = !(ax >= ay);
    const float c = (steep ? ax : ay) / ((steep ? ay : ax) + eps), c2 = c * c;
    const float pl = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = steep ? 90.f - pl : pl;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Deterministic sin/cos: fp64 Cody-Waite reduction by pi/2 + fixed minimax polynomials on
// [-pi/4, pi/4], explicit operation order, rounded once to fp32 (SURVEY.md A.6 build rule: the
// reference's libm cosf/sinf, src/ORBextractor.cc:123, is not reproducible across hosts).
__device__ __forceinline__ void dev_sincos(float angle_rad, float *s_out, float *c_out)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)angle_rad;
    double k = rint(x * INV_PIO2);
    double r = (x - k * PIO2_HI) - k * PIO2_LO;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    int q = (int)((long long)k & 3);
    double s, c;
    if (q == 0) { s = sn; c = cs; }
    else if (q == 1) { s = cs; c = -sn; }
    else if (q == 2) { s = -sn; c = -cs; }
    else { s = -cs; c = sn; }
    *s_out = (float)s;
    *c_out = (float)c;
}

// cvRound on a float that is known to be small: round-half-to-even
__device__ __forceinline__ int dev_cv_round(float v) { return (int)rintf(v); }

__device__ __forceinline__ int hamming256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
    return d;
}
