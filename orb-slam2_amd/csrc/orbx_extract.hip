// orbx_extract.hip — ORB extractor for gfx950 (MI355X): pyramid, per-cell FAST-9/16 + NMS,
// quadtree cull, intensity-centroid orientation, on-patch 7x7 blur and steered rBRIEF-256.
//
// Replaces ORBextractor::operator() (reference src/ORBextractor.cc:1261-1339) and everything it
// calls; bit-exact contract in SURVEY.md Appendix A/B.  Written for wave64 / LDS staging; all
// arithmetic is integer except fastAtan2 / the pattern rotation (fp32, contraction off) and the
// shared fp64 sincos.  One launch covers a whole batch of images (grid.y = image).
#include "orbx_device.h"
#include <atomic>
#include "orb_pattern.inc"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

__constant__ uint32_t c_pat4[256];        // x0 | y0<<8 | x1<<16 | y1<<24, signed bytes (src/ORBextractor.cc:160-418, data)
__constant__ uint4 c_omask[64];           // IC_Angle: per lane (row, half) the byte mask of its 16-pixel window inside the circular patch

// ================================================================ K1: pyramid level (E2)
// cv::resize INTER_LINEAR 8UC1 from level l-1 to level l.  Coefficient tables are
// computed on the host with the reference's float/double arithmetic; the kernel is pure integer.
// A one-wave workgroup produces a 256 x RS_TH (8) output tile, lane = four adjacent output columns, all rows:
//  * the source rectangle (<= 311 x 12) goes to LDS with direct loads (global_load_lds_dwordx4: three whole source rows per
//    instruction, any byte alignment, no VGPR round trip, no address arithmetic per element); 3840 bytes of LDS per wave =
//    the CU's maximum of 32 waves (the kernel is latency bound: 16-row tiles, 21 waves per CU, were 3 % slower);
//  * the loop runs over SOURCE rows (fully unrolled: every LDS offset is an immediate): the horizontal interpolation
//    of a source row is computed once and serves the (up to two) output rows it belongs to -- at scale 1.2 that is
//    1.33 instead of 2 horizontal passes per output row; an output row is emitted as soon as its lower source row
//    is done.  Which output row that is comes from the tile row's host record and depends only on the tile (every lane has the
//    same rows): scalar control flow -- a few hundred scalar instructions per tile whatever its width, which is why a lane
//    takes four columns (with two the kernel was bound by the scalar unit, not by the vector ALUs).
// One-wave workgroups need no barrier partners and drift apart in time, so loads of one tile overlap arithmetic of
// another on the same CU (a 16-wave workgroup walking all levels of an image with barriers between them was no faster).
#define RS_PX 4      // output columns per lane
#define RS_TW (64 * RS_PX)
#ifndef RS_TH_LOG2
#define RS_TH_LOG2 3
#endif
#define RS_TH (1 << RS_TH_LOG2)
#define RS_NT 64     // threads per workgroup: one wave
#define RS_PITCH 320 // LDS bytes per staged source row (>= 1.2 * RS_TW + 2 + 3): twenty 16-byte pieces, three rows per direct load
#define RS_ROWS (RS_TH == 16 ? 22 : 12)   // source rows of a tile at scale 1.2: ceil(1.2 * RS_TH) + 2

// Launch constants by value, and everything a tile needs to start its loads in ONE record per tile row / tile column (host
// tables): the wave's first dependent fetch is already the last one before the direct loads (it used to walk kernel arguments
// -> geometry -> coefficient tables -> emit table, holding its LDS all the while).
struct ResizeArgs {
    int d_w, d_pitch, s_w, s_pitch, s_level0;
    int tab_x, tab_tx, tab_ty;          // int16 units into the table buffer
    long long d_off, s_off;             // byte offsets of the two levels inside one image's pyramid block
};
#define RS_TY_REC (4 + 4 * RS_ROWS)     // tile-row record, int16 units: (first source row, source rows, 0, 0), then RS_ROWS x (e, b0, b1, 0)

__global__ __launch_bounds__(RS_NT) void k_resize(const ResizeArgs A, PyrRef pr,
                                                uint8_t *__restrict__ pyr_w, const int16_t *__restrict__ tabs)
{
    __shared__ __align__(16) uint8_t src_t[RS_ROWS * RS_PITCH];
    // image-fastest grid: consecutive workgroups (dealt round-robin over the XCDs) take the same tile of different images, so what is in
    // flight on the chip at any time is spread over every image of the batch (1.4 MB apart) instead of packed into a few -- measured 2-3 %
    // faster than tile-fastest, like every attempt to keep neighbouring tiles on one XCD was slower
    const int b = blockIdx.x, lane = threadIdx.x, tile_x = blockIdx.y, tile_y = blockIdx.z;
    const int x_t = tile_x * RS_TW;
    const int spitch = A.s_level0 ? pr.img0_pitch : A.s_pitch;
    const uint8_t *src = A.s_level0 ? pr.img0 + (long long)b * pr.img0_stride : pr.pyr + (long long)b * pr.pyr_stride + A.s_off;
    uint8_t *dst = pyr_w + (long long)b * pr.pyr_stride + A.d_off;
    const int16_t *tx = tabs + A.tab_x;
    const int16_t *ry = tabs + A.tab_ty + tile_y * RS_TY_REC, *rx = tabs + A.tab_tx + 4 * tile_x;
    const int sy_min = __builtin_amdgcn_readfirstlane((int)ry[0]), nrows = __builtin_amdgcn_readfirstlane((int)ry[1]);
    const int sx_min = __builtin_amdgcn_readfirstlane((int)rx[0]), nfull = __builtin_amdgcn_readfirstlane((int)rx[1]),
              tail = __builtin_amdgcn_readfirstlane((int)rx[2]);
    // Per SOURCE row of the tile (wave-uniform scalars): which output row is complete once this source row has been
    // interpolated, and its vertical weights: (y | skip << 12 | same << 13 | two << 14, b0, b1, 0), or e = -1 for none;
    // rows of other tiles are already filtered out by the host
    short4 qs[RS_ROWS];
#pragma unroll
    for (int k = 0; k < RS_ROWS; k++) qs[k] = *reinterpret_cast<const short4 *>(ry + 4 + 4 * k);
    // the lane's output columns: source offsets and the 11-bit weights (requested before the tile loads: independent of them)
    const int x4 = x_t + RS_PX * lane;
    short4 qx[RS_PX];
#pragma unroll
    for (int i = 0; i < RS_PX; i++) qx[i] = *reinterpret_cast<const short4 *>(tx + 4 * min(x4 + i, A.d_w - 1)); // (ofs, a0, a1, 0)
    {
        // 16-byte pieces that lie wholly inside the source row are fetched by direct loads (global_load_lds_dwordx4: 1 KB per wave
        // instruction, any byte alignment): lane = (row lane / 20, piece lane % 20) of three whole rows per load -- the LDS pitch
        // of 320 bytes is exactly twenty pieces.  The < 16 bytes a right-edge tile still needs behind the last whole piece are
        // fetched as bytes (a piece there could reach past the caller's last image row).
        const uint8_t *s0 = src + (long long)sy_min * spitch + sx_min;
        const int lr = lane / 20, lc = lane - lr * 20;
        if (lr < 3 && lc < nfull)
            for (int r = 0; r < nrows; r += 3)
                if (r + lr < nrows)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(s0 + (long long)(r + lr) * spitch + 16 * lc),
                                                     reinterpret_cast<uint32_t *>(src_t + r * RS_PITCH), 16, 0, 0);
        if (tail)
            for (int i = lane; i < nrows * tail; i += RS_NT) {
                const int r = i / tail, c = 16 * nfull + (i - r * tail);
                src_t[r * RS_PITCH + c] = s0[(long long)r * spitch + c];
            }
    }
    int o0[RS_PX], o1[RS_PX], a0[RS_PX], a1[RS_PX];
#pragma unroll
    for (int i = 0; i < RS_PX; i++) {
        const int sx0 = qx[i].x;
        o0[i] = sx0 - sx_min;
        o1[i] = min(sx0 + 1, A.s_w - 1) - sx_min;
        a0[i] = qx[i].y;
        a1[i] = qx[i].z;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the direct loads have landed in LDS
    __syncthreads();
    const int d_w = A.d_w, d_pitch = A.d_pitch;
    int hp[RS_PX], hc[RS_PX];                   // (t >> 4) of source rows k - 1 and k
#pragma unroll
    for (int i = 0; i < RS_PX; i++) hp[i] = hc[i] = 0;
    uint8_t *dcol = dst + x4;
    // the weights are non-negative and each pair sums to 2048 (+-1 by rounding), so v stays inside [0, 255]:
    // ((2049 * (255 * 2049 >> 4)) >> 16) + 2 >> 2 == 255 -- cv::resize's saturate_cast never fires for INTER_LINEAR
    // (b * h) >> 16 as the high half of (b << 16) * h: one multiply, no shift (b <= 2049, h < 2^15: the product stays below 2^42)
#define EMIT(Y, B0, B1, HA) do { \
        const unsigned w0_ = (unsigned)(B0) << 16, w1_ = (unsigned)(B1) << 16; \
        uint32_t out_ = 0; \
        _Pragma("unroll") for (int i = 0; i < RS_PX; i++) out_ |= ((__umulhi(w0_, (unsigned)(HA)[i]) + __umulhi(w1_, (unsigned)hc[i]) + 2u) >> 2) << (8 * i); \
        if (x4 < d_w) *reinterpret_cast<uint32_t *>(dcol + (long long)(Y) * d_pitch) = out_; } while (0)
#pragma unroll
    for (int k = 0; k < RS_ROWS; k++) {         // fully unrolled: every LDS offset below is an immediate
        if (k < nrows) {                        // wave-uniform
#pragma unroll
            for (int i = 0; i < RS_PX; i++) {
                hp[i] = hc[i];
                hc[i] = (src_t[k * RS_PITCH + o0[i]] * a0[i] + src_t[k * RS_PITCH + o1[i]] * a1[i]) >> 4;
            }
            const int e = __builtin_amdgcn_readfirstlane((int)qs[k].x);
            if (e >= 0) {
                const int y = e & 0xFFF;
                if (!(e & 0x1000)) {
                    const int b0 = __builtin_amdgcn_readfirstlane((int)qs[k].y), b1 = __builtin_amdgcn_readfirstlane((int)qs[k].z);
                    if (e & 0x2000) EMIT(y, b0, b1, hc);                // bottom clamp: both source rows are this one
                    else EMIT(y, b0, b1, hp);
                }
                // two output rows end on the clamped last source row when consecutive levels have equal heights: the second one
                // is y + 1 with both rows = this one and the clamp weights (2048, 0)
                if (e & 0x4000) EMIT(y + 1, 2048, 0, hc);
            }
        }
    }
#undef EMIT
}

// Fallback for scale factors whose source rectangle does not fit the LDS tile of k_resize
// (ORB-SLAM2 always uses 1.2): same arithmetic straight from global memory, 4 pixels per thread.
__global__ __launch_bounds__(256) void k_resize_direct(const Geom *__restrict__ g, int l, PyrRef pr,
                                                       uint8_t *__restrict__ pyr_w, const int16_t *__restrict__ tabs)
{
    const LevelGeom &D = g->lv[l];
    const LevelGeom &S = g->lv[l - 1];
    const int b = blockIdx.z;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= D.h || x4 >= D.pitch) return;
    int spitch;
    const uint8_t *src = orbx_level_ptr(pr, S, l - 1, b, &spitch);
    uint8_t *dst = pyr_w + (long long)b * pr.pyr_stride + D.pyr_off;
    if (D.resize_lds == 2) { // exact 2x in both directions: cv::resize switches INTER_LINEAR to the 2x2 area average
        const uint8_t *r0 = src + (long long)(2 * y) * spitch, *r1 = r0 + spitch;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = x4 + i;
            if (x < D.w) out |= (uint32_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2) << (8 * i);
        }
        *reinterpret_cast<uint32_t *>(dst + (long long)y * D.pitch + x4) = out;
        return;
    }
    const int16_t *tx = tabs + D.tab_x, *ty = tabs + D.tab_y;
    const int sy0 = ty[4 * y], b0 = ty[4 * y + 1], b1 = ty[4 * y + 2];
    const int sy1 = sy0 + 1 < S.h ? sy0 + 1 : S.h - 1;
    const uint8_t *r0 = src + (long long)sy0 * spitch, *r1 = src + (long long)sy1 * spitch;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 + i;
        if (x < D.w) {
            const int sx0 = tx[4 * x], a0 = tx[4 * x + 1], a1 = tx[4 * x + 2];
            const int sx1 = sx0 + 1 < S.w ? sx0 + 1 : S.w - 1;
            const int t0 = r0[sx0] * a0 + r0[sx1] * a1;
            const int t1 = r1[sx0] * a0 + r1[sx1] * a1;
            int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : v > 255 ? 255 : v;
            out |= (uint32_t)v << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t *>(dst + (long long)y * D.pitch + x4) = out;
}

// ---------------------------------------------------------------- K1g: several pyramid levels per launch (small launches)
// A single frame's pyramid is seven dependent launches of a few microseconds of work each: its time is the launch chain (7 x 5 us), not
// the pixels.  k_pyr_group shortens the chain: a workgroup owns one tile of the LAST level of a group of consecutive levels and computes,
// level by level through two LDS buffers, every pixel of the earlier levels that tile depends on (host tables: per tile column / row and
// level the region [lo, hi) and the part [lo, own_hi) it also writes to the pyramid -- the regions of neighbouring tiles overlap by the
// interpolation halo, the owned parts tile each level exactly).  The halo pixels are computed twice (1.4 x the pixels for five levels on
// 32 x 16 tiles), which is why batches keep k_resize; the arithmetic per pixel is k_resize's, from the same coefficient tables.
#define ORBX_PYR_GROUP_MAX 7
#define PG_TW 32
#define PG_TH 16
#define PG_NT 512
#define PG_LDS_LIMIT (64 * 1024)
struct PyrGroupLevel { int w, h, pitch, tab_x, tab_y, pad; long long pyr_off; };
struct PyrGroupArgs {
    int n, s_level0, s_w, s_h, s_pitch, tab_cx, tab_cy, lds_b;
    long long s_off;
    PyrGroupLevel lv[ORBX_PYR_GROUP_MAX];
};
#define PG_CX_REC 8     // int16 units per (tile column, step): lo, hi, own_hi, dwords per source row (step 0), magic of the dwords / 4-pixel groups per row (lo, hi), 0, 0
#define PG_CY_REC 4     // per (tile row, step): lo, hi, own_hi, 0

extern __shared__ __align__(16) uint8_t pg_smem[];

__global__ __launch_bounds__(PG_NT) void k_pyr_group(const PyrGroupArgs A, PyrRef pr, uint8_t *__restrict__ pyr_w, const int16_t *__restrict__ tabs)
{
    const int b = blockIdx.z, tid = threadIdx.x;
    const int16_t *cx = tabs + A.tab_cx + (int)blockIdx.x * PG_CX_REC * (A.n + 1);
    const int16_t *cy = tabs + A.tab_cy + (int)blockIdx.y * PG_CY_REC * (A.n + 1);
    const int spitch = A.s_level0 ? pr.img0_pitch : A.s_pitch;
    const uint8_t *src = A.s_level0 ? pr.img0 + (long long)b * pr.img0_stride : pr.pyr + (long long)b * pr.pyr_stride + A.s_off;
    uint8_t *cur = pg_smem, *nxt = pg_smem + A.lds_b;
    int ox = cx[0], oy = cy[0], cp;             // origin and pitch of the region in `cur`
    {
        const int rows = cy[1] - oy, ndw = cx[3];
        const unsigned magic = (unsigned)(uint16_t)cx[4] | ((unsigned)(uint16_t)cx[5] << 16);
        cp = 4 * ndw;
        if ((((uintptr_t)src | (unsigned)spitch) & 3) == 0) {       // ox is a multiple of 4: whole aligned dwords, never past the row's pitch
            for (int i = tid; i < rows * ndw; i += PG_NT) {
                const int r = ndw == 1 ? i : (int)__umulhi((unsigned)i, magic), c = i - r * ndw;   // (2^32 / 1 has no 32-bit magic)
                reinterpret_cast<uint32_t *>(cur)[i] = *reinterpret_cast<const uint32_t *>(src + (long long)(oy + r) * spitch + ox + 4 * c);
            }
        } else {                                                     // a caller's level-0 image at an odd address or pitch: bytes inside the row only
            const int wb = min(cp, A.s_w - ox);
            for (int i = tid; i < rows * cp; i += PG_NT) {
                const int r = i / cp, c = i - r * cp;
                if (c < wb) cur[i] = src[(long long)(oy + r) * spitch + ox + c];
            }
        }
    }
    __syncthreads();
    int s_w = A.s_w, s_h = A.s_h;
    for (int k = 1; k <= A.n; k++) {
        const PyrGroupLevel &L = A.lv[k - 1];
        const int16_t *qx_ = cx + PG_CX_REC * k, *qy_ = cy + PG_CY_REC * k;
        const int lx = qx_[0], hx = qx_[1], own_x = qx_[2], ly = qy_[0], hy = qy_[1], own_y = qy_[2];
        const unsigned magic = (unsigned)(uint16_t)qx_[4] | ((unsigned)(uint16_t)qx_[5] << 16);
        const int rw = hx - lx, rh = hy - ly, dp = (rw + 3) & ~3;
        const int16_t *tx = tabs + L.tab_x, *ty = tabs + L.tab_y;
        uint8_t *dst = pyr_w + (long long)b * pr.pyr_stride + L.pyr_off;
        const bool keep = k < A.n;               // the last level of the group is only written out
        // a thread takes four adjacent pixels of a row: one row record, the four column records as two 16-byte loads, one packed
        // LDS store (the region's rows are dword aligned in `nxt`) and one dword store to the pyramid
        const int ng = (rw + 3) >> 2;
        for (int i = tid; i < ng * rh; i += PG_NT) {
            const int yy = ng == 1 ? i : (int)__umulhi((unsigned)i, magic), xx = 4 * (i - yy * ng);
            const int x = lx + xx, y = ly + yy;
            const short4 qy = *reinterpret_cast<const short4 *>(ty + 4 * y);
            const int sy0 = qy.x, sy1 = min(sy0 + 1, s_h - 1);
            const uint8_t *r0 = cur + (sy0 - oy) * cp - ox, *r1 = cur + (sy1 - oy) * cp - ox;
            short4 qx[4];
            if (xx + 3 < rw) {           // the four records are contiguous (8 bytes each, 8-byte aligned)
                const uint2 *tp = reinterpret_cast<const uint2 *>(tx + 4 * x);
                const uint2 t0_ = tp[0], t1_ = tp[1], t2_ = tp[2], t3_ = tp[3];
                qx[0] = *reinterpret_cast<const short4 *>(&t0_); qx[1] = *reinterpret_cast<const short4 *>(&t1_);
                qx[2] = *reinterpret_cast<const short4 *>(&t2_); qx[3] = *reinterpret_cast<const short4 *>(&t3_);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) qx[j] = *reinterpret_cast<const short4 *>(tx + 4 * min(x + j, hx - 1));
            }
            uint32_t out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sx0 = qx[j].x, sx1 = min(sx0 + 1, s_w - 1);
                const int t0 = r0[sx0] * qx[j].y + r0[sx1] * qx[j].z;
                const int t1 = r1[sx0] * qx[j].y + r1[sx1] * qx[j].z;
                const int v = (((qy.y * (t0 >> 4)) >> 16) + ((qy.z * (t1 >> 4)) >> 16) + 2) >> 2;   // in [0, 255], see k_resize
                out |= (uint32_t)v << (8 * j);
            }
            if (keep) *reinterpret_cast<uint32_t *>(nxt + yy * dp + xx) = out;
            if (y < own_y) {
                uint8_t *d = dst + (long long)y * L.pitch + x;
                if (x + 3 < own_x) *reinterpret_cast<uint32_t *>(d) = out;      // (any byte alignment: global memory takes unaligned dwords)
                else
#pragma unroll
                    for (int j = 0; j < 4; j++) if (x + j < own_x) d[j] = (uint8_t)(out >> (8 * j));
            }
        }
        __syncthreads();
        { uint8_t *t = cur; cur = nxt; nxt = t; }
        ox = lx; oy = ly; cp = dp; s_w = L.w; s_h = L.h;
    }
}

// ================================================================ K2: FAST per cell (E3)
// cornerScore<16> without a threshold: with x_k the 16 ring pixels, A = max over the 16 arcs of 9
// contiguous ring pixels of min(v - x) = v - min_arcs(max_arc x) and B = max_arcs(min_arc x) - v.
// A pixel is a FAST-9 corner at threshold t iff max(A,B) > t and its OpenCV score is then
// max(A,B)-1 independent of t, so one score map at minThFAST serves both passes of
// src/ORBextractor.cc:988-995.  The sliding 9-window max/min over the circular ring is a doubling
// 3x3 composition of three-input min/max (v_min3_i32 / v_max3_i32).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 pk(unsigned lo, unsigned hi)
{
    const unsigned v = lo | (hi << 16);
    return __builtin_bit_cast(u16x2, v);
}

template <int P>
__device__ __forceinline__ int fast_score_full(const uint8_t *t, int th)
{
    const int v = t[0];
    int x[16];
    x[0] = t[3 * P];      x[1] = t[3 * P + 1];  x[2] = t[2 * P + 2];   x[3] = t[P + 3];
    x[4] = t[3];          x[5] = t[-P + 3];     x[6] = t[-2 * P + 2];  x[7] = t[-3 * P + 1];
    x[8] = t[-3 * P];     x[9] = t[-3 * P - 1]; x[10] = t[-2 * P - 2]; x[11] = t[-P - 3];
    x[12] = t[-3];        x[13] = t[P - 3];     x[14] = t[2 * P - 2];  x[15] = t[3 * P - 1];
    // window 9 = 3 x 3 with three-input min/max (v_min3_i32 / v_max3_i32)
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = min(min(x[k], x[(k + 1) & 15]), x[(k + 2) & 15]);
        hi3[k] = max(max(x[k], x[(k + 1) & 15]), x[(k + 2) & 15]);
    }
    int lo9[16], hi9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo9[k] = min(min(lo3[k], lo3[(k + 3) & 15]), lo3[(k + 6) & 15]);
        hi9[k] = max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]);
    }
    int a[6], bq[6];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a[k] = max(max(lo9[3 * k], lo9[3 * k + 1]), lo9[3 * k + 2]);
        bq[k] = min(min(hi9[3 * k], hi9[3 * k + 1]), hi9[3 * k + 2]);
    }
    const int max_of_min = max(max(max(a[0], a[1]), a[2]), max(max(a[3], a[4]), lo9[15]));
    const int min_of_max = min(min(min(bq[0], bq[1]), bq[2]), min(min(bq[3], bq[4]), hi9[15]));
    const int s = max(v - min_of_max, max_of_min - v);
    return s > th ? s - 1 : 0;
}

// The same score for TWO pixels per lane (round 4).  gfx950 has packed three-input f16 minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16)
// at the issue cost of v_min3_u32 (4.4 cycles per wave-instruction, tools/ubench/pk3_cost.hip) -- two three-input comparisons per instruction.
// An 8-bit pixel x travels as the half-precision bit pattern 0x4000 + x: a positive NORMAL number (2 + x / 512) whose order is the order of
// x, so the float minimum / maximum of patterns IS the integer minimum / maximum of pixels (all 2^24 triples x both halves checked on the
// device by the microbenchmark; no denormal mode, NaN or signed zero can be involved).  The low halves carry candidate a, the high halves
// candidate b of the lane: the arc network below is the one of fast_score_full, instruction for instruction, on 128 candidates at a time.
__device__ __forceinline__ unsigned pk_min3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pk_max3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// FAST_PK_BIAS: 0x40004000 makes every pattern a NORMAL half (0x4000 + x) at the price of one v_or_b32 per ring pixel; with 0 the patterns are the
// half-precision SUBNORMALS 0x0000 .. 0x00FF, whose order is also the order of x and which v_pk_minimum3_f16 / v_pk_maximum3_f16 compare exactly as
// long as the wave's float mode keeps f16 denormals (MODE.FP_DENORM[3:2] = 3: LLVM's default for every AMDGPU kernel; k_fast sets it itself on
// entry so that the result cannot depend on a build flag).  Both forms checked over all 2^24 triples by tools/ubench/pk3_cost.hip; 0.743 -> 0.718 ms.
#ifndef FAST_PK_BIAS
#define FAST_PK_BIAS 0u
#endif
template <int P>
__device__ __forceinline__ void fast_score_pair(const uint8_t *ta, const uint8_t *tb, int th, int *sa, int *sb)
{
    constexpr int off[16] = { 3 * P, 3 * P + 1, 2 * P + 2, P + 3, 3, -P + 3, -2 * P + 2, -3 * P + 1,
                              -3 * P, -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3, 2 * P - 2, 3 * P - 1 };
    // (the packing costs a v_perm_b32 + a v_or_b32 per ring pixel.  ds_read_u8_d16 / _d16_hi would pack in the load, but on an SRAM-ECC part
    // -- gfx950:sramecc+ -- a d16 load clobbers the other half of its register: tried from inline assembly, wrong results, and not faster)
    unsigned x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        u16x2 v2;
        v2.x = ta[off[k]]; v2.y = tb[off[k]];
        x[k] = __builtin_bit_cast(unsigned, v2) | FAST_PK_BIAS;
    }
    unsigned lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = pk_min3(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
        hi3[k] = pk_max3(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
    }
    unsigned lo9[16], hi9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo9[k] = pk_min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        hi9[k] = pk_max3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    }
    unsigned a[5], bq[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a[k] = pk_max3(lo9[3 * k], lo9[3 * k + 1], lo9[3 * k + 2]);
        bq[k] = pk_min3(hi9[3 * k], hi9[3 * k + 1], hi9[3 * k + 2]);
    }
    const unsigned mom = pk_max3(pk_max3(a[0], a[1], a[2]), pk_max3(a[3], a[4], lo9[15]), lo9[15]);     // max over arcs of the arc minimum, both halves
    const unsigned mox = pk_min3(pk_min3(bq[0], bq[1], bq[2]), pk_min3(bq[3], bq[4], hi9[15]), hi9[15]); // min over arcs of the arc maximum
    const int va = ta[0], vb = tb[0];
    const int s_a = max(va - (int)(mox & 0xFFu), (int)(mom & 0xFFu) - va);
    const int s_b = max(vb - (int)((mox >> 16) & 0xFFu), (int)((mom >> 16) & 0xFFu) - vb);
    *sa = s_a > th ? s_a - 1 : 0;
    *sb = s_b > th ? s_b - 1 : 0;
}

// One wave (64-thread workgroup) per (cell, image) -- no workgroup barriers, many independent cells in flight per CU.
// The kernel is VALU-issue bound, and on gfx950 only a few wave64 opcodes issue at the full rate (add / sub / and / or /
// xor / lshr / mov and the 16-bit VOP2 forms: ~2.5 cycles per wave-instruction; min / max / min3 / perm / alignbyte /
// packed-16 / mul / cmp / cndmask / mbcnt / DPP / SDWA: ~4.3, tools/ubench/op_cost.hip), so every phase is written for few
// instruction-cycles per pixel:
//  1. the cell (+3 px halo) goes to LDS with direct loads (global_load_lds_dword: no VGPR round trip, no ds_write; one
//     instruction = RPL whole tile rows), all in flight at once, while the wave zeroes its score tile and bitmaps;
//  2. pretest on 4 horizontally adjacent pixels per lane in byte-parallel (SWAR) form with full-rate ops only: on values
//     halved to 7 bits a borrow-free per-byte subtract leaves "x7 <= c7 - s7" in bit 7 of every byte.  Halving makes the test
//     CONSERVATIVE (never a false negative; floor((c-s)/2) <= c7 - s7), which is all a pretest needs: a FAST-9 corner has two
//     adjacent compass pixels (of N, E, S, W at distance 3) darker than c - t or brighter than c + t.  The four flag bits are
//     OR-ed into a per-row candidate bitmap in LDS (ds_or_b32): no ballots, no per-iteration prefix sums;
//  3. lane = row: the row bitmaps are unrolled into the dense ordered candidate list (one wave prefix sum per cell);
//  4. threshold-free cornerScore (v_min3 / v_max3 arc network) on dense lanes -> score tile;
//  5. strict in-cell 3x3 maximum per listed pixel -> survivor bitmap; lane = row again: ordered emission.
// Threshold schedule of src/ORBextractor.cc:988-995 as it stands: the whole sequence runs at iniThFAST; only a cell that
// keeps nothing (no corner, or only tied maxima) runs again at minThFAST.  Scores do not depend on the threshold and the
// iniThFAST pretest passes ~40 % fewer pixels to the score network than a minThFAST one, which nearly every textured cell
// used to pay for.  Candidate = x | y<<12 | score<<24, (x,y) relative to (16,16).
extern __shared__ __align__(16) unsigned char fast_smem[];
#ifndef FAST_XG
#define FAST_XG 2
#endif
// Launch constants of k_fast by value (kernel-argument segment): fetching them through the Geom pointer was one more level in
// the chain of dependent scalar loads every wave starts with (arguments -> geometry -> cell record -> tile).
struct FastArgs {
    int total_cells;
    int lds_sc, lds_list, lds_bm;   // LDS carve: score tile, candidate list, candidate bitmap (the survivor bitmap follows it)
    int bm_rows;                    // bitmap rows (u64 each): tallest detect area plus the row overrun of the last pretest iteration
    int ini_th, min_th;
    int list_cap;                   // entries of the candidate list (used by the several-waves-per-cell form; one wave: ORBX_FAST_LIST_CAP)
    long long cand_total;
};

#ifdef ORBX_DIAG
__device__ unsigned long long g_fast_stamp[4096 * 8]; // diagnostic build only: summed phase cycles of k_fast, 4096 slots
__device__ unsigned long long g_desc_stamp[4096 * 8]; // same for k_desc
__device__ unsigned long long g_tree_stamp[4096 * 8]; // same for the level-0 workgroups of k_tree (slot 6 = phase-2 sweeps, 7 = workgroups)
#ifdef ORBX_DIAG_SPANS_ONLY     // the summed phase stamps perturb the waves they measure (an atomic per phase): here every wave of k_fast logs
                                // the end of its phases in its own slot instead
__device__ unsigned g_fast_phase[16384][8];
#define STAMP_TO(arr, k) do { (void)_t_prev; if ((const void *)arr == (const void *)g_fast_stamp && threadIdx.x == 0) { \
    const unsigned _id = blockIdx.x + gridDim.x * blockIdx.y; if (_id < 16384) g_fast_phase[_id][k] = (unsigned)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP_TO(arr, k) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    if (threadIdx.x == 0) atomicAdd(&arr[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + (k)], _t - _t_prev); _t_prev = _t; } while (0)
#endif
// wall-clock life of every wave of a launch (s_memrealtime, 100 MHz): (start, end) in the wave's own slot -- no atomics, nothing shared
#define SPAN_SLOTS 16384
__device__ uint2 g_span[2][SPAN_SLOTS];
#define SPAN_BEGIN() const unsigned _sp0 = (unsigned)__builtin_amdgcn_s_memrealtime()
#define SPAN_END(K) do { if ((threadIdx.x & 63) == 0) { const unsigned _id = blockIdx.x + gridDim.x * blockIdx.y; \
    if (_id < SPAN_SLOTS) g_span[K][_id] = make_uint2(_sp0, (unsigned)__builtin_amdgcn_s_memrealtime()); } } while (0)
#define STAMP(k) STAMP_TO(g_fast_stamp, k)
#define DSTAMP(k) STAMP_TO(g_desc_stamp, k)
#define TSTAMP(k) do { if (blockIdx.y == 0) STAMP_TO(g_tree_stamp, k); } while (0)
// timeline of the level-0 tree of image 0: (tag, m, cycles since the previous entry)
__device__ unsigned g_tree_tl[256][4];
#define TLOG(tag, mval) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && _tl_n < 256) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    g_tree_tl[_tl_n][0] = (tag); g_tree_tl[_tl_n][1] = (unsigned)(mval); g_tree_tl[_tl_n][2] = (unsigned)(_t - _tl_prev); g_tree_tl[_tl_n][3] = 1; _tl_n++; _tl_prev = _t; } } while (0)
#else
#define TLOG(tag, mval) do { } while (0)
#define SPAN_BEGIN() do { } while (0)
#define SPAN_END(K) do { } while (0)
#define STAMP(k) do { } while (0)
#define DSTAMP(k) do { } while (0)
#define TSTAMP(k) do { } while (0)
#endif

// P / SP = LDS pitches of the pixel tile and the score tile: (48, 40) when every cell of the pyramid is at most 38 px wide
// (one direct load = 5 tile rows of 12 dwords, 60 lanes), else (80, 64) (3 rows of 20 dwords).  Both tile pitches put rows
// r and r + 8 (and no closer pair) on the same LDS banks: candidates line up along vertical image edges, and with a
// 64-byte pitch (rows r, r + 2 on the same banks) the byte reads of the score network ran 3.4x the bank-conflict cycles.
// NW = waves per cell: 1 for batches (above).  A launch of a frame or two leaves most of the chip idle and lasts as long as its
// fullest cell (a cell with 5x the candidates of the median one ran 15.6 us against 5.3 us: the score and maximum loops walk the
// candidate list 64 at a time): there NW waves share the cell -- tile rows, pretest rows and list entries are dealt round-robin to
// the waves, the bitmaps and tiles are the workgroup's, list and emission stay with wave 0.  Same results by construction: every
// phase writes disjoint bytes or ORs bits, and the phases are separated by the barriers the one-wave form already has.
template <int P, int SP, int NW, bool IMG_FAST = false>
__global__ __launch_bounds__(64 * NW) void k_fast(const FastArgs fa, const CellRec *__restrict__ cells, PyrRef pr,
                                             int *__restrict__ cell_cnt, uint32_t *__restrict__ cand, uint32_t *__restrict__ cand_prim)
{
    constexpr int DWR = P / 4;      // dwords per tile row = lanes per row of one direct load
    constexpr int RPL = 64 / DWR;   // whole tile rows per direct load (lanes >= RPL * DWR stay idle)
    // The kernel is VALU-issue bound and sensitive to where its code lies: shifted by an ODD number of dwords (its 8-byte instructions
    // then straddle 8-byte fetch units) it runs 2.5 % slower, any even shift is the same (tools/ab_fast_only.py on -DORBX_FAST_PAD=1..15
    // builds).  A one-instruction change near the top of the kernel had moved it by 4 bytes: when this kernel changes, compare both parities.
#ifdef ORBX_FAST_PAD    // experiment: shift the kernel's code by ORBX_FAST_PAD dwords (s_nop 0)
    asm volatile(".fill %0, 4, 0xBF800000" :: "n"(ORBX_FAST_PAD));
#endif
    if (FAST_PK_BIAS == 0u)
        __builtin_amdgcn_s_setreg((1 << 11) | (6 << 6) | 1, 3);      // hwreg(HW_REG_MODE, 6, 2) = 3: f16 / f64 denormals kept (fast_score_pair compares subnormal patterns)
    uint8_t *tile = fast_smem;
    uint8_t *sc = fast_smem + fa.lds_sc;
    uint16_t *list = reinterpret_cast<uint16_t *>(fast_smem + fa.lds_list);
    uint32_t *bm = reinterpret_cast<uint32_t *>(fast_smem + fa.lds_bm); // candidate bitmap, then survivor bitmap: u64 per row
    uint32_t *sv = bm + 2 * fa.bm_rows;
    const int ini_th = fa.ini_th, min_th = fa.min_th;
    // IMG_FAST (one wave per cell, batches): grid (image, cell) -- image-fastest, see k_resize: consecutive workgroups take the same cell of
    // different images (0.789 -> 0.778 ms per 512 images); else grid (cell, image): a frame or two, or more cells than grid.y can hold
    const int b = IMG_FAST ? blockIdx.x : blockIdx.y;
    const int lane = NW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63), wv = NW == 1 ? 0 : (int)(threadIdx.x >> 6), tid = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8, speed only): remap so that FAST_XG
    // horizontally adjacent cells land on the same XCD (their halos share cache lines in that XCD's L2) while each
    // XCD's work stays spread over the whole image (contiguous runs per XCD were measured slower).
    int cell;
    {
        const int bx = IMG_FAST ? blockIdx.y : blockIdx.x;
        const int grp = bx / (8 * FAST_XG), r = bx - grp * (8 * FAST_XG);
        cell = grp * (8 * FAST_XG) + (r & 7) * FAST_XG + (r >> 3);
        if (cell >= fa.total_cells) return;
    }
#ifdef ORBX_DIAG
    unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
#endif
    SPAN_BEGIN();
    // the 40-byte record as ten dwords (scalar loads; 16-bit fields fetched by themselves become vector loads on gfx950)
    CellRec rec;
    {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(cells + cell);
        uint32_t w[10];
#pragma unroll
        for (int i = 0; i < 10; i++) w[i] = cw[i];
        rec.level = (short)(w[0] & 0xFFFF); rec.skip = (short)(w[0] >> 16);
        rec.ini_x = (short)(w[1] & 0xFFFF); rec.ini_y = (short)(w[1] >> 16);
        rec.tw = (short)(w[2] & 0xFFFF); rec.th = (short)(w[2] >> 16);
        rec.pitch = (int)w[3]; rec.cand_cap = (int)w[4]; rec.gpr_magic = w[5];
        rec.pyr_off = (long long)(((unsigned long long)w[7] << 32) | w[6]);
        rec.cand_slot = (long long)(((unsigned long long)w[9] << 32) | w[8]);
    }
    int *my_cnt = cell_cnt + (long long)b * fa.total_cells + cell;
    if (rec.skip) { // src/ORBextractor.cc:961-976 skip rules, evaluated on the host
        if (lane == 0) *my_cnt = 0;
        return;
    }
    const int ini_x = rec.ini_x, ini_y = rec.ini_y, tw = rec.tw, th = rec.th, dw = tw - 6, dh = th - 6;
    const int pitch = rec.level == 0 ? pr.img0_pitch : rec.pitch;
    const uint8_t *img = rec.level == 0 ? pr.img0 + (long long)b * pr.img0_stride
                                        : pr.pyr + (long long)b * pr.pyr_stride + rec.pyr_off;
    // ---- 1. tile: the fetch starts one byte left of the cell (gfx950 global and LDS-direct loads need no alignment), so the
    // first detectable pixel always sits at tile column 4: every pretest group of four pixels is a whole LDS dword whatever
    // the cell's position or the caller's pitch.  Lane = (row lane / DWR, dword lane % DWR) of RPL whole rows per load; the data
    // lands at tile + RPL * P * k + 4 * lane, i.e. row-major with pitch P.
    constexpr int xo = 1; // tile column of image column ini_x
    {
        const int lr0 = lane / DWR, lc = lane - lr0 * DWR;
        const int ndw = (tw + xo + 3) >> 2;             // dwords per row that hold cell pixels (<= 17)
        // scalar row base + one 32-bit lane offset: the row groups advance on the scalar unit, no 64-bit vector adds per load
        const uint8_t *base = img + (long long)ini_y * pitch + (ini_x - xo);
        const unsigned voff = (unsigned)(lr0 * pitch + 4 * lc);
        const int full = th / RPL;
        if (lc < ndw && lr0 < RPL) {
            base += (long long)wv * RPL * pitch;
            for (int k = wv; k < full; k += NW, base += (long long)NW * RPL * pitch)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(base + voff), reinterpret_cast<uint32_t *>(tile + RPL * P * k), 4, 0, 0);
            // the last, partial group of rows never reads below the cell (after the loop `base` stands at this wave's next group:
            // the partial group is `full`, taken by the wave whose turn it is)
            if ((NW == 1 || full % NW == wv) && full * RPL + lr0 < th)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(base + voff), reinterpret_cast<uint32_t *>(tile + RPL * P * full), 4, 0, 0);
        }
    }
    {   // meanwhile: zero score tile (1-px zero rim included) and both bitmaps
        uint4 *z = reinterpret_cast<uint4 *>(sc);
        for (int i = tid; i < ((dh + 2) * SP + 15) / 16; i += 64 * NW) z[i] = make_uint4(0, 0, 0, 0);
        uint4 *zb = reinterpret_cast<uint4 *>(bm);
        for (int i = tid; i < fa.bm_rows; i += 64 * NW) zb[i] = make_uint4(0, 0, 0, 0);   // 2 bitmaps x 8 bytes per row
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the direct loads have landed in LDS
    __syncthreads();
    STAMP(0);
    const uint8_t *t0 = tile + 3 * P + xo + 3;           // detectable pixel (0, 0)
    // pretest geometry: tile dwords 1 .. gpr of a row hold the detectable pixels (tile columns 4 .. dw + 3); one iteration =
    // rpi whole rows, lane = (row lr, group gq); lanes beyond rpi * gpr idle with an empty pixel mask
    const int gpr = (dw + 3) >> 2;
    // floor(i / gpr) by multiply-high with the host's magic number (an integer division costs ~30 instructions here)
    const int lq = gpr == 1 ? lane : (int)__umulhi((unsigned)lane, rec.gpr_magic);
    const int rpi = min(gpr == 1 ? 64 : (int)__umulhi(64u, rec.gpr_magic), 8);   // (the last column's cells can be narrow: keep the row overrun <= 8)
    const int lr = min(lq, rpi), gq = lane - lq * gpr;
    const int nvalid = max(1, min(4, dw - 4 * gq));      // only the last group of a row can be partial
    const unsigned vmask = lr < rpi ? (0x80808080u >> (8 * (4 - nvalid))) : 0u;
    const int bm_sh = 4 * (gq & 7);
    uint32_t *slot = cand + (long long)b * fa.cand_total + rec.cand_slot;
    uint32_t *prim = cand_prim + ((long long)b * fa.total_cells + cell) * ORBX_CAND_PRIM;   // the first 16 candidates: dense, 64 B per cell
    // Bitmap rows are walked with one lane per SEGMENT: a whole row (64 bits, low word then high word), or -- when the detect
    // area is at most 32 x 32, most cells of a 30-px grid -- half a 32-bit row word: the bit loops below run as long as the
    // fullest segment, and half rows are half as full
    const bool half_mode = dh <= 32 && dw <= 32;
    const int brow = half_mode ? lane >> 1 : lane;
    const unsigned bsh = half_mode ? 16u * (lane & 1) : 0u;
    int th_cur = ini_th, nsurv = 0;
    int emitted = 0;            // one wave, one round: maxima written straight from the corner list (see below)
    bool direct = false;
    for (int pass = 0; pass < 2; pass++) {
        // ---- 2. SWAR pretest.  With s = t + 1 and x7 = x >> 1 per byte: x < c - t  ==>  x7 <= c7 - s7 (dark) and
        // x > c + t  ==>  (127 - x7) <= (127 - c7) - s7 (bright).  R = (c7 | 0x80) - s7 cannot borrow across bytes; its bit 7
        // says c7 >= s7 (else no x can pass) and its low 7 bits are c7 - s7; (R | 0x80) - x7 then has bit 7 set iff
        // x7 <= c7 - s7.  The bright side is the same on complemented values, folded into an add: KB + x7 with
        // KB = (RB | 0x80) - 0x7f.  Corner candidates: (N | S) & (E | W) on either side.
        {
            const unsigned s7 = (unsigned)((th_cur + 1) >> 1) * 0x01010101u;
            const uint8_t *pc = tile + (lr + 3 + wv * rpi) * P + 4 * (1 + gq);
            uint32_t *pb = bm + (lr + wv * rpi) * 2 + (gq >> 3);
            for (int r0 = wv * rpi; r0 < dh; r0 += NW * rpi, pc += NW * rpi * P, pb += NW * rpi * 2) {
                const uint32_t *rc = reinterpret_cast<const uint32_t *>(pc);
                const unsigned C = rc[0], Wd = rc[-1], Ed = rc[1], N = rc[3 * DWR], S = rc[-3 * DWR];
                const unsigned Wv = __builtin_amdgcn_alignbyte(C, Wd, 1), Ev = __builtin_amdgcn_alignbyte(Ed, C, 3);
                const unsigned c7 = (C >> 1) & 0x7f7f7f7fu, n7 = (N >> 1) & 0x7f7f7f7fu, u7 = (S >> 1) & 0x7f7f7f7fu,
                               e7 = (Ev >> 1) & 0x7f7f7f7fu, w7 = (Wv >> 1) & 0x7f7f7f7fu;
                const unsigned R = (c7 | 0x80808080u) - s7, RD = R | 0x80808080u;
                const unsigned RB = ((c7 ^ 0x7f7f7f7fu) | 0x80808080u) - s7, KB = (RB | 0x80808080u) - 0x7f7f7f7fu;
                const unsigned dark = ((RD - n7) | (RD - u7)) & ((RD - e7) | (RD - w7)) & R;
                const unsigned bright = ((KB + n7) | (KB + u7)) & ((KB + e7) | (KB + w7)) & RB;
                const unsigned any = (dark | bright) & vmask;
                // bits 7, 15, 23, 31 -> one nibble: the multiplier routes bit 8k of (any >> 7) to bit 24 + k, no carries
                const unsigned nib = (((any >> 7) * 0x01020408u) >> 24) << bm_sh;
                if (nib) atomicOr(pb, nib);   // rows >= dh of the last iteration land in bitmap rows that are never read
            }
        }
        __syncthreads();
        STAMP(1);
        // ---- 3. bitmap -> ordered list of (py << 6 | px): lane = segment, exclusive prefix of the segment populations.  The list holds
        // ORBX_FAST_LIST_CAP entries (LDS is what limits the waves per CU, and a textured cell lists ~130 of its ~1000 pixels);
        // a cell with more candidates takes them in rounds of that many: all scores first, then the maxima
        unsigned c_lo = 0, c_hi = 0;
        if (brow < dh) { const uint2 m = *reinterpret_cast<const uint2 *>(bm + 2 * brow); c_lo = half_mode ? (m.x >> bsh) & 0xFFFFu : m.x; c_hi = half_mode ? 0u : m.y; }
        const int c_cnt = __popc(c_lo) + __popc(c_hi);
        const int c_incl = wave_incl_scan(c_cnt);
        const int nlist = __builtin_amdgcn_readlane(c_incl, 63);
        const unsigned rowbits = ((unsigned)brow << 6) | bsh;
        // ---- 4. full score on the compacted pixels (dense lanes); entries ascend in (py, px)
        // With `compact` the entries that turned out to be corners (3 % of the pixels, against the 13 % the pretest lists) are
        // packed to the front of the list in place (a write never passes the reads of its own or a later iteration): the
        // maximum search below then takes one iteration where the full list took two or three.  Returns their number.
        auto score_entries = [&](int n, bool compact) -> int {
            int n2 = 0;
            if (NW == 1) {
                // one wave: 128 entries per iteration, two per lane (fast_score_pair); lane L takes entries i0 + L and i0 + 64 + L, so the
                // corners of the first 64 precede those of the second 64 in the compacted list as they did in the list
                for (int i0 = 0; i0 < n; i0 += 128) {
                    const int ia = i0 + lane, ib = ia + 64;
                    const bool in_a = ia < n, in_b = ib < n;
                    const int ea = list[in_a ? ia : i0], eb = list[in_b ? ib : i0];      // (entry i0 always exists: a lane without an entry recomputes it and drops the result)
                    const int pya = ea >> 6, pxa = ea & 63, pyb = eb >> 6, pxb = eb & 63;
                    int sa, sb;
                    fast_score_pair<P>(t0 + pya * P + pxa, t0 + pyb * P + pxb, th_cur, &sa, &sb);
                    if (!in_a) sa = 0;
                    if (!in_b) sb = 0;
                    if (in_a) sc[(pya + 1) * SP + pxa + 1] = (uint8_t)sa;
                    if (in_b) sc[(pyb + 1) * SP + pxb + 1] = (uint8_t)sb;
                    if (compact) {
                        const unsigned long long ma = __ballot(sa > 0), mb = __ballot(sb > 0);
                        const int na = __popcll(ma);
                        if (sa > 0) list[n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u))] = (uint16_t)ea;
                        if (sb > 0) list[n2 + na + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))] = (uint16_t)eb;
                        n2 += na + __popcll(mb);
                    }
                }
                return compact ? n2 : n;
            }
            // several waves per cell: wave w takes the entries [128 (w + NW k), 128 (w + NW k) + 128), two per lane as above; no compaction (the
            // maximum search walks the whole list: entries that are no corners have score 0)
            for (int i0 = 128 * wv; i0 < n; i0 += 128 * NW) {
                const int ia = i0 + lane, ib = ia + 64;
                const bool in_a = ia < n, in_b = ib < n;
                const int ea = list[in_a ? ia : i0], eb = list[in_b ? ib : i0];
                const int pya = ea >> 6, pxa = ea & 63, pyb = eb >> 6, pxb = eb & 63;
                int sa, sb;
                fast_score_pair<P>(t0 + pya * P + pxa, t0 + pyb * P + pxb, th_cur, &sa, &sb);
                if (in_a) sc[(pya + 1) * SP + pxa + 1] = (uint8_t)sa;
                if (in_b) sc[(pyb + 1) * SP + pxb + 1] = (uint8_t)sb;
            }
            return n;
        };
        // ---- 5. strict 3x3 maximum of the listed pixels (only they can score > 0) -> survivor bitmap
        auto mark_maxima = [&](int n) {
            for (int i0 = 64 * wv; i0 < n; i0 += 64 * NW) {
                const int i = i0 + lane;
                if (i < n) {          // lanes without an entry issue no LDS traffic at all (an LDS atomic costs per active lane, also one that ORs a zero)
                    const int e = list[i], py = e >> 6, px = e & 63;
                    const uint8_t *c = sc + (py + 1) * SP + px + 1;
                    const int s = c[0];
                    const int nb = max(max(max((int)c[-1], (int)c[1]), max((int)c[-SP - 1], (int)c[-SP])),
                                       max(max((int)c[-SP + 1], (int)c[SP - 1]), max((int)c[SP], (int)c[SP + 1])));
                    if (s > nb) atomicOr(sv + 2 * py + (px >> 5), 1u << (px & 31));   // s > nb >= 0 implies a corner at th_cur
                }
            }
        };
        const int list_cap = NW == 1 ? ORBX_FAST_LIST_CAP : fa.list_cap;     // several waves: a whole cell, always one round
        if (nlist <= list_cap) {
            if (NW == 1) {
                unsigned lo = c_lo, hi = c_hi;
                uint16_t *lp = list + (c_incl - c_cnt);
                while (lo) { *lp++ = (uint16_t)(rowbits | (unsigned)__builtin_ctz(lo)); lo &= lo - 1; }
                while (hi) { *lp++ = (uint16_t)(rowbits | 32u | (unsigned)__builtin_ctz(hi)); hi &= hi - 1; }
            } else {
                // every wave has the same segments and offsets: wave w unrolls bits [16 w / NW * ..) of each 16-bit quarter -- the bit
                // walk is as long as the fullest piece, and a piece is 1 / NW of what one wave walked
                constexpr int PIECE = 64 / 4;                        // a 64-bit row in four 16-bit quarters, each cut in NW pieces
                const unsigned long long rowm = (unsigned long long)c_lo | ((unsigned long long)c_hi << 32);
                uint16_t *lp0 = list + (c_incl - c_cnt);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int b0 = q * PIECE + (PIECE * wv) / NW, b1 = q * PIECE + (PIECE * (wv + 1)) / NW;
                    const unsigned long long below0 = (1ull << b0) - 1ull, below1 = b1 >= 64 ? ~0ull : (1ull << b1) - 1ull;   // b0 < 64 always
                    unsigned long long m = rowm & below1 & ~below0;
                    uint16_t *lp = lp0 + __popcll(rowm & below0);
                    while (m) { *lp++ = (uint16_t)(rowbits | (unsigned)__builtin_ctzll(m)); m &= m - 1; }
                }
            }
            __syncthreads();
            STAMP(5);
            const int ncorner = score_entries(nlist, true);
            STAMP(6);
            __syncthreads();
            STAMP(2);
            if (NW == 1) {
                // one wave, one round (nearly every cell): the compacted corner list is already in the cell's row-major order, so the strict
                // maxima among them are EMITTED as they are found -- a ballot and a rank per 64 corners -- instead of going through the
                // survivor bitmap, a second prefix sum over its segments and a bit walk per segment (~55 of a cell's 730 vector instructions)
                int run = 0;
                const int X0e = ini_x + 3 - ORBX_MIN_BORDER, Y0e = ini_y + 3 - ORBX_MIN_BORDER;
                for (int i0 = 0; i0 < ncorner; i0 += 64) {
                    const int i = i0 + lane;
                    bool is_max = false;
                    uint32_t recw = 0;
                    if (i < ncorner) {
                        const int e = list[i], py = e >> 6, px = e & 63;
                        const uint8_t *c = sc + (py + 1) * SP + px + 1;
                        const int s = c[0];
                        const int nb = max(max(max((int)c[-1], (int)c[1]), max((int)c[-SP - 1], (int)c[-SP])),
                                           max(max((int)c[-SP + 1], (int)c[SP - 1]), max((int)c[SP], (int)c[SP + 1])));
                        is_max = s > nb;                 // s > nb >= 0 implies a corner at th_cur
                        recw = (uint32_t)(X0e + px) | ((uint32_t)(Y0e + py) << 12) | ((uint32_t)s << 24);
                    }
                    const unsigned long long m = __ballot(is_max);
                    const int o = run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (is_max && o < rec.cand_cap) (o < ORBX_CAND_PRIM ? prim : slot)[o] = recw;
                    run += __popcll(m);
                }
                emitted = run;
                direct = true;
            } else {
                mark_maxima(ncorner);
                __syncthreads();
            }
        } else {
            auto list_round = [&](int base) {   // the candidates of rank base .. base + CAP - 1
                if (wv != 0) return;
                unsigned lo = c_lo, hi = c_hi;
                int r = c_incl - c_cnt - base;
                while (lo) { if ((unsigned)r < (unsigned)list_cap) list[r] = (uint16_t)(rowbits | (unsigned)__builtin_ctz(lo)); r++; lo &= lo - 1; }
                while (hi) { if ((unsigned)r < (unsigned)list_cap) list[r] = (uint16_t)(rowbits | 32u | (unsigned)__builtin_ctz(hi)); r++; hi &= hi - 1; }
            };
            for (int base = 0; base < nlist; base += list_cap) {
                list_round(base);
                __syncthreads();
                score_entries(min(list_cap, nlist - base), false);
                __syncthreads();
            }
            STAMP(2);
            for (int base = 0; base < nlist; base += list_cap) {
                list_round(base);
                __syncthreads();
                mark_maxima(min(list_cap, nlist - base));
                __syncthreads();
            }
        }
        STAMP(3);
        if (direct) {   // (wave-uniform) the maxima of this pass are already in the cell's slots
            if (emitted != 0 || th_cur == min_th) break;
            th_cur = min_th;
            direct = false;
            continue;
        }
        {
            unsigned lo = 0, hi = 0;
            if (brow < dh) { const uint2 m = *reinterpret_cast<const uint2 *>(sv + 2 * brow); lo = half_mode ? (m.x >> bsh) & 0xFFFFu : m.x; hi = half_mode ? 0u : m.y; }
            nsurv = __popc(lo) + __popc(hi);
        }
        // the cell falls back to minThFAST only if iniThFAST kept nothing (:991-995); pretest, scores and bitmaps of the
        // second pass are supersets of the first, so nothing has to be cleared
        if (__builtin_amdgcn_readfirstlane(__any(nsurv != 0)) || th_cur == min_th) break;
        th_cur = min_th;
    }
    if (direct) {
        if (lane == 0) *my_cnt = min(emitted, rec.cand_cap);
    } else
    // ---- ordered (row-major) emission into the cell's candidate slots: lane = bitmap segment (row, or half a row)
    if (wv == 0) {
        unsigned lo = 0, hi = 0;
        if (brow < dh) { const uint2 m = *reinterpret_cast<const uint2 *>(sv + 2 * brow); lo = half_mode ? (m.x >> bsh) & 0xFFFFu : m.x; hi = half_mode ? 0u : m.y; }
        const int incl = wave_incl_scan(nsurv);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int o = incl - nsurv;
        const int Y = ini_y + 3 + brow - ORBX_MIN_BORDER, X0 = ini_x + 3 - ORBX_MIN_BORDER + (int)bsh;
        const uint8_t *srow = sc + (brow + 1) * SP + 1 + bsh;
        while (lo) {
            const int px = __builtin_ctz(lo);
            lo &= lo - 1;
            if (o < rec.cand_cap) (o < ORBX_CAND_PRIM ? prim : slot)[o] = (uint32_t)(X0 + px) | ((uint32_t)Y << 12) | ((uint32_t)srow[px] << 24);
            o++;
        }
        while (hi) {
            const int px = 32 + __builtin_ctz(hi);
            hi &= hi - 1;
            if (o < rec.cand_cap) (o < ORBX_CAND_PRIM ? prim : slot)[o] = (uint32_t)(X0 + px) | ((uint32_t)Y << 12) | ((uint32_t)srow[px] << 24);
            o++;
        }
        if (lane == 0) *my_cnt = min(total, rec.cand_cap);
    }
    STAMP(4);
    SPAN_END(0);
#ifdef ORBX_DIAG
    if (lane == 0) atomicAdd(&g_fast_stamp[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + 7], 1ull);
#endif
}

// ---------------------------------------------------------------- K2p: FAST on PAIRS of horizontally adjacent cells (batches)
// One wave takes two cells of a cell row, A and its right neighbour B (the last cell of a row with an odd number of columns goes
// alone).  What one cell per wave pays per cell -- record fetch and decode, tile load and zeroing, the pretest prologue, a prefix sum
// and a bit walk for the list, one for the emission -- is paid once per pair, the score / maximum iterations (64 candidates each)
// run on ONE concatenated candidate list (a 31 x 32 cell lists ~127 pretest candidates: two or three iterations, the last two-thirds
// empty; a pair lists ~254: four or five), and the 3-px halo between A and B is fetched once (72 bytes per tile row for 62 owned
// pixels instead of 2 x 40 for 2 x 31).
// Bit space: everything after the tile load is indexed by (row, bit), bit = tile column - 4.  The tile fetch starts xo = 1 + off bytes
// left of cell A with off = 32 - dwA, so A's detectable pixels are bits [off, 32) and B's are bits [32, 32 + dwB): the A | B boundary
// is a 32-bit word boundary of every bitmap row and a dword boundary of every pretest group, and "which cell" is bit 5 of a bit index.
// (A cell that goes alone sits at bits [0, dwA).)  Per-cell semantics of src/ORBextractor.cc:953-1009 are kept exactly:
//  * non-maximum suppression is per cell (cv::FAST sees one cell at a time: scores outside it count as 0): the score tile has a
//    zero column between the two cells (score column = bit + 1 + (bit >> 5)), so a maximum never looks into the other cell;
//  * the minThFAST fallback is per cell: a second pass lists only the segments of the cell(s) that kept nothing;
//  * each cell's survivors go to its own candidate slots in its own row-major order: the emission scans a packed (A | B << 16) count.
// Bitmap segments: 32-bit halves of a row (lane = row, half; the half is the cell) when the detect area has at most 32 rows, else
// whole 64-bit rows (lane = row; low word = A, high word = B).
#ifndef FAST2_P
#define FAST2_P 72      // tile pitch: 1 + 32 - dwA + 3 + dwA + dwB + 3 <= 71 bytes
#endif
#ifndef FAST2_LIST_CAP
#define FAST2_LIST_CAP 1024 // candidates listed per round: a pair lists ~350 on a textured frame, and the rounds of a fuller list walk it twice without compaction
#endif
#ifndef FAST2_SP
#define FAST2_SP 68     // score pitch: rim + 32 + gap + 32 + rim = 67 bytes; 17 dwords: rows r and r + 32 share banks
#endif
struct PairRec {
    short level, ncells;         // ncells: cells whose count this wave writes (1 or 2); dwa == 0: all of them skipped
    short ini_x, ini_y;          // cell A's rectangle origin (incl. the 3-px halo), level coordinates
    short dwa, dwb;              // detect widths of A and B (0: skipped / absent; dwb != 0 implies dwa == the level's cell width)
    int cell;                    // index of cell A in the image's cell arrays (B = cell + 1)
    int pitch, cand_cap;
    unsigned gpr_magic;          // multiply-high division by gpr = pretest groups per row = (bits used + 3) >> 2
    int th;                      // tile rows (detect rows + 6)
    long long pyr_off, cand_slot;
};
static_assert(sizeof(PairRec) == 48, "PairRec layout");

template <int P, int SP>
__global__ __launch_bounds__(64) void k_fast2(const FastArgs fa, const PairRec *__restrict__ pairs, int total_pairs, PyrRef pr,
                                              int *__restrict__ cell_cnt, uint32_t *__restrict__ cand, uint32_t *__restrict__ cand_prim)
{
    constexpr int DWR = P / 4;      // dwords per tile row
    constexpr int RPL = 64 / DWR;   // whole tile rows per direct load
#ifdef ORBX_FAST_PAD
    asm volatile(".fill %0, 4, 0xBF800000" :: "n"(ORBX_FAST_PAD));
#endif
    uint8_t *tile = fast_smem;
    uint8_t *sc = fast_smem + fa.lds_sc;
    uint16_t *list = reinterpret_cast<uint16_t *>(fast_smem + fa.lds_list);
    uint32_t *bm = reinterpret_cast<uint32_t *>(fast_smem + fa.lds_bm); // candidate bitmap, then survivor bitmap: u64 per row
    uint32_t *sv = bm + 2 * fa.bm_rows;
    const int ini_th = fa.ini_th, min_th = fa.min_th;
    const int b = blockIdx.y, lane = threadIdx.x;
    int pi;
    {   // FAST_XG neighbouring pairs of a cell row on the same XCD (see k_fast)
        const int bx = blockIdx.x, grp = bx / (8 * FAST_XG), r = bx - grp * (8 * FAST_XG);
        pi = grp * (8 * FAST_XG) + (r & 7) * FAST_XG + (r >> 3);
        if (pi >= total_pairs) return;
    }
#ifdef ORBX_DIAG
    unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
#endif
    SPAN_BEGIN();
    int level, ncells, ini_x, ini_y, dwa, dwb, cell0, rpitch, cand_cap, th;
    unsigned gpr_magic;
    long long pyr_off, cand_slot;
    {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(pairs + pi);
        uint32_t w[12];
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = cw[i];
        level = (short)(w[0] & 0xFFFF); ncells = (short)(w[0] >> 16);
        ini_x = (short)(w[1] & 0xFFFF); ini_y = (short)(w[1] >> 16);
        dwa = (short)(w[2] & 0xFFFF); dwb = (short)(w[2] >> 16);
        cell0 = (int)w[3]; rpitch = (int)w[4]; cand_cap = (int)w[5]; gpr_magic = w[6]; th = (int)w[7];
        pyr_off = (long long)(((unsigned long long)w[9] << 32) | w[8]);
        cand_slot = (long long)(((unsigned long long)w[11] << 32) | w[10]);
    }
    int *my_cnt = cell_cnt + (long long)b * fa.total_cells + cell0;
    if (dwa == 0) { // src/ORBextractor.cc:961-976 skip rules, evaluated on the host (a skipped A has no B to its right that is not skipped)
        if (lane < ncells) my_cnt[lane] = 0;
        return;
    }
    const int off = dwb ? 32 - dwa : 0, xo = 1 + off;   // bit of A's first detectable pixel; bytes fetched left of cell A
    const int nbits = off + dwa + dwb;                  // bits [off, nbits) are detectable pixels
    const int tw = dwa + dwb + 6, dh = th - 6;
    const int pitch = level == 0 ? pr.img0_pitch : rpitch;
    const uint8_t *img = level == 0 ? pr.img0 + (long long)b * pr.img0_stride : pr.pyr + (long long)b * pr.pyr_stride + pyr_off;
    // ---- 1. tile (cell A, cell B and the halo around both) -> LDS by direct loads, RPL whole rows per instruction
    {
        const int lr0 = lane / DWR, lc = lane - lr0 * DWR;
        const int ndw = (tw + xo + 3) >> 2;             // dwords per row that hold tile pixels (<= 18)
        const uint8_t *base = img + (long long)ini_y * pitch + (ini_x - xo);
        const unsigned voff = (unsigned)(lr0 * pitch + 4 * lc);
        const int full = th / RPL;
        if (lc < ndw && lr0 < RPL) {
            for (int k = 0; k < full; k++, base += (long long)RPL * pitch)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(base + voff), reinterpret_cast<uint32_t *>(tile + RPL * P * k), 4, 0, 0);
            if (full * RPL + lr0 < th)      // the last, partial group of rows never reads below the cells
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(base + voff), reinterpret_cast<uint32_t *>(tile + RPL * P * full), 4, 0, 0);
        }
    }
    {   // meanwhile: zero score tile (rim and the column between the cells included) and both bitmaps
        uint4 *z = reinterpret_cast<uint4 *>(sc);
        for (int i = lane; i < ((dh + 2) * SP + 15) / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
        uint4 *zb = reinterpret_cast<uint4 *>(bm);
        for (int i = lane; i < fa.bm_rows; i += 64) zb[i] = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(0);
    const uint8_t *t0 = tile + 3 * P + 4;               // detect row 0, bit 0
    // pretest geometry: tile dwords 1 .. gpr of a row hold bits 0 .. 4 gpr - 1; one iteration = rpi whole rows
    const int gpr = (nbits + 3) >> 2;
    const int lq = gpr == 1 ? lane : (int)__umulhi((unsigned)lane, gpr_magic);
    const int rpi = min(gpr == 1 ? 64 : (int)__umulhi(64u, gpr_magic), 8);
    const int lr = min(lq, rpi), gq = lane - lq * gpr;
    unsigned vmask = 0;
    {   // bytes of this lane's group whose bit is a detectable pixel: bits [max(off, 4 gq), min(nbits, 4 gq + 4))
        const int lo = min(max(off - 4 * gq, 0), 4), hi = min(max(nbits - 4 * gq, 0), 4);
        if (lr < rpi && hi > lo) vmask = (0x80808080u >> (8 * (4 - hi))) & (0x80808080u << (8 * lo));
    }
    const int bm_sh = 4 * (gq & 7);
    uint32_t *slot = cand + (long long)b * fa.cand_total + cand_slot;
    uint32_t *prim = cand_prim + ((long long)b * fa.total_cells + cell0) * ORBX_CAND_PRIM;
    const bool half_mode = dh <= 32;
    const int brow = half_mode ? lane >> 1 : lane;
    const int half = half_mode ? lane & 1 : 0;
    const unsigned bsh = 32u * (unsigned)half;
    int th_cur = ini_th;
    unsigned s_lo = 0, s_hi = 0;                        // this lane's survivor bits
    bool need_a = true, need_b = dwb != 0;              // cells listed by the current pass
    for (int pass = 0; pass < 2; pass++) {
        // ---- 2. SWAR pretest (see k_fast): both cells, 4 adjacent bits per lane
        {
            const unsigned s7 = (unsigned)((th_cur + 1) >> 1) * 0x01010101u;
            const uint8_t *pc = tile + (lr + 3) * P + 4 * (1 + gq);
            uint32_t *pb = bm + lr * 2 + (gq >> 3);
            for (int r0 = 0; r0 < dh; r0 += rpi, pc += rpi * P, pb += rpi * 2) {
                const uint32_t *rc = reinterpret_cast<const uint32_t *>(pc);
                const unsigned C = rc[0], Wd = rc[-1], Ed = rc[1], N = rc[3 * DWR], S = rc[-3 * DWR];
                const unsigned Wv = __builtin_amdgcn_alignbyte(C, Wd, 1), Ev = __builtin_amdgcn_alignbyte(Ed, C, 3);
                const unsigned c7 = (C >> 1) & 0x7f7f7f7fu, n7 = (N >> 1) & 0x7f7f7f7fu, u7 = (S >> 1) & 0x7f7f7f7fu,
                               e7 = (Ev >> 1) & 0x7f7f7f7fu, w7 = (Wv >> 1) & 0x7f7f7f7fu;
                const unsigned R = (c7 | 0x80808080u) - s7, RD = R | 0x80808080u;
                const unsigned RB = ((c7 ^ 0x7f7f7f7fu) | 0x80808080u) - s7, KB = (RB | 0x80808080u) - 0x7f7f7f7fu;
                const unsigned dark = ((RD - n7) | (RD - u7)) & ((RD - e7) | (RD - w7)) & R;
                const unsigned bright = ((KB + n7) | (KB + u7)) & ((KB + e7) | (KB + w7)) & RB;
                const unsigned any = (dark | bright) & vmask;
                const unsigned nib = (((any >> 7) * 0x01020408u) >> 24) << bm_sh;
                if (nib) atomicOr(pb, nib);   // rows >= dh of the last iteration land in bitmap rows that are never read
            }
        }
        __syncthreads();
        STAMP(1);
        // ---- 3. bitmap -> ordered list of (row << 6 | bit): lane = segment; only the cells this pass is for
        unsigned c_lo = 0, c_hi = 0;
        if (brow < dh) {
            const uint2 m = *reinterpret_cast<const uint2 *>(bm + 2 * brow);
            if (half_mode) c_lo = half ? (need_b ? m.y : 0u) : (need_a ? m.x : 0u);
            else { c_lo = need_a ? m.x : 0u; c_hi = need_b ? m.y : 0u; }
        }
        const int c_cnt = __popc(c_lo) + __popc(c_hi);
        const int c_incl = wave_incl_scan(c_cnt);
        const int nlist = __builtin_amdgcn_readlane(c_incl, 63);
        const unsigned rowbits = ((unsigned)brow << 6) | bsh;
        // ---- 4. full score of the listed pixels; corners packed to the front of the list in place (order kept)
        auto score_entries = [&](int n, bool compact) -> int {
            int n2 = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                int e = 0, s = 0;
                if (i < n) {
                    e = list[i];
                    const int py = e >> 6, bx = e & 63;
                    s = fast_score_full<P>(t0 + py * P + bx, th_cur);
                    sc[(py + 1) * SP + bx + 1 + (bx >> 5)] = (uint8_t)s;
                }
                if (compact) {
                    const unsigned long long m = __ballot(s > 0);
                    if (s > 0) list[n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint16_t)e;
                    n2 += __popcll(m);
                }
            }
            return compact ? n2 : n;
        };
        // ---- 5. strict 3x3 maximum inside the pixel's own cell -> survivor bitmap
        auto mark_maxima = [&](int n) {
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                if (i < n) {
                    const int e = list[i], py = e >> 6, bx = e & 63;
                    const uint8_t *c = sc + (py + 1) * SP + bx + 1 + (bx >> 5);
                    const int s = c[0];
                    const int nb = max(max(max((int)c[-1], (int)c[1]), max((int)c[-SP - 1], (int)c[-SP])),
                                       max(max((int)c[-SP + 1], (int)c[SP - 1]), max((int)c[SP], (int)c[SP + 1])));
                    if (s > nb) atomicOr(sv + 2 * py + (bx >> 5), 1u << (bx & 31));
                }
            }
        };
        constexpr int list_cap = FAST2_LIST_CAP;
        if (nlist <= list_cap) {
            unsigned lo = c_lo, hi = c_hi;
            uint16_t *lp = list + (c_incl - c_cnt);
            while (lo) { *lp++ = (uint16_t)(rowbits | (unsigned)__builtin_ctz(lo)); lo &= lo - 1; }
            while (hi) { *lp++ = (uint16_t)(rowbits | 32u | (unsigned)__builtin_ctz(hi)); hi &= hi - 1; }
            __syncthreads();
            STAMP(5);
            const int ncorner = score_entries(nlist, true);
            STAMP(6);
            __syncthreads();
            STAMP(2);
            mark_maxima(ncorner);
            __syncthreads();
        } else {
            auto list_round = [&](int base) {   // the candidates of rank base .. base + CAP - 1
                unsigned lo = c_lo, hi = c_hi;
                int r = c_incl - c_cnt - base;
                while (lo) { if ((unsigned)r < (unsigned)list_cap) list[r] = (uint16_t)(rowbits | (unsigned)__builtin_ctz(lo)); r++; lo &= lo - 1; }
                while (hi) { if ((unsigned)r < (unsigned)list_cap) list[r] = (uint16_t)(rowbits | 32u | (unsigned)__builtin_ctz(hi)); r++; hi &= hi - 1; }
            };
            for (int base = 0; base < nlist; base += list_cap) {
                list_round(base);
                __syncthreads();
                score_entries(min(list_cap, nlist - base), false);
                __syncthreads();
            }
            STAMP(2);
            for (int base = 0; base < nlist; base += list_cap) {
                list_round(base);
                __syncthreads();
                mark_maxima(min(list_cap, nlist - base));
                __syncthreads();
            }
        }
        STAMP(3);
        s_lo = 0; s_hi = 0;
        if (brow < dh) { const uint2 m = *reinterpret_cast<const uint2 *>(sv + 2 * brow); s_lo = half_mode ? (half ? m.y : m.x) : m.x; s_hi = half_mode ? 0u : m.y; }
        // a cell falls back to minThFAST only if iniThFAST kept nothing IN THAT CELL (:991-995); the second pass lists only such cells.
        // Pretest bits, scores and survivors of the second pass are supersets of the first: nothing has to be cleared
        if (th_cur == min_th) break;
        const bool any_a = __any(half_mode ? (half == 0 && s_lo != 0) : s_lo != 0);
        const bool any_b = __any(half_mode ? (half == 1 && s_lo != 0) : s_hi != 0);
        need_a = !any_a; need_b = dwb != 0 && !any_b;
        if (!(need_a || need_b)) break;
        th_cur = min_th;
    }
    // ---- ordered (row-major per cell) emission into each cell's candidate slots: lane = bitmap segment
    {
        const int n_lo = __popc(s_lo), n_hi = __popc(s_hi);
        const int v = half_mode ? n_lo << (16 * half) : n_lo | (n_hi << 16);
        const int incl = wave_incl_scan(v);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int excl = incl - v;
        const int Y = ini_y + 3 + brow - ORBX_MIN_BORDER, X0 = ini_x + 3 - ORBX_MIN_BORDER - off;     // x of bit 0
        const uint8_t *srow = sc + (brow + 1) * SP + 1;
        {   // low word: cell A (whole-row segments) or this lane's cell (half-row segments)
            int o = half_mode ? (excl >> (16 * half)) & 0xFFFF : excl & 0xFFFF;
            uint32_t *pm = prim + (half ? ORBX_CAND_PRIM : 0), *sl = slot + (half ? cand_cap : 0);
            unsigned lo = s_lo;
            while (lo) {
                const int bx = (int)bsh + __builtin_ctz(lo);
                lo &= lo - 1;
                if (o < cand_cap) (o < ORBX_CAND_PRIM ? pm : sl)[o] = (uint32_t)(X0 + bx) | ((uint32_t)Y << 12) | ((uint32_t)srow[bx + (bx >> 5)] << 24);
                o++;
            }
        }
        if (!half_mode) {   // high word: cell B
            int o = excl >> 16;
            uint32_t *pm = prim + ORBX_CAND_PRIM, *sl = slot + cand_cap;
            unsigned hi = s_hi;
            while (hi) {
                const int bx = 32 + __builtin_ctz(hi);
                hi &= hi - 1;
                if (o < cand_cap) (o < ORBX_CAND_PRIM ? pm : sl)[o] = (uint32_t)(X0 + bx) | ((uint32_t)Y << 12) | ((uint32_t)srow[bx + 1] << 24);
                o++;
            }
        }
        if (lane == 0) my_cnt[0] = min(total & 0xFFFF, cand_cap);
        if (lane == 1 && ncells == 2) my_cnt[1] = min(total >> 16, cand_cap);
    }
    STAMP(4);
    SPAN_END(0);
#ifdef ORBX_DIAG
    if (lane == 0) atomicAdd(&g_fast_stamp[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + 7], 1ull);
#endif
}

#ifdef ORBX_DIAG
#ifdef ORBX_DIAG_SPANS_ONLY
extern "C" int orbx_diag_fast_phases(unsigned *out /*[16384][8]*/)
{
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_phase), sizeof(unsigned) * 16384 * 8));
    return ORBX_OK;
}
#endif

extern "C" int orbx_diag_spans(unsigned *out /*[2][SPAN_SLOTS][2]*/, int reset)
{
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(uint2) * 2 * SPAN_SLOTS));
    if (reset) {
        static uint2 z[2][SPAN_SLOTS];
        ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_span), z, sizeof z));
    }
    return ORBX_OK;
}

extern "C" int orbx_diag_desc_stamps(unsigned long long *out, int reset)
{
    ORBX_HIP(hipDeviceSynchronize());
    static unsigned long long h[4096 * 8];
    ORBX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_desc_stamp), sizeof h));
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int i = 0; i < 4096; i++) out[k] += h[i * 8 + k]; }
    if (reset) { memset(h, 0, sizeof h); ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_desc_stamp), h, sizeof h)); }
    return ORBX_OK;
}

extern "C" int orbx_diag_tree_timeline(unsigned *out /*[256][4]*/)
{
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_tl), sizeof(unsigned) * 1024));
    static unsigned z[1024];
    ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tree_tl), z, sizeof z));
    return ORBX_OK;
}

extern "C" int orbx_diag_tree_stamps(unsigned long long *out, int reset)
{
    ORBX_HIP(hipDeviceSynchronize());
    static unsigned long long h[4096 * 8];
    ORBX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tree_stamp), sizeof h));
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int i = 0; i < 4096; i++) out[k] += h[i * 8 + k]; }
    if (reset) { memset(h, 0, sizeof h); ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tree_stamp), h, sizeof h)); }
    return ORBX_OK;
}

extern "C" int orbx_diag_fast_stamps(unsigned long long *out, int reset)
{
    ORBX_HIP(hipDeviceSynchronize());
    static unsigned long long h[4096 * 8];
    ORBX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fast_stamp), sizeof h));
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int i = 0; i < 4096; i++) out[k] += h[i * 8 + k]; }
    if (reset) { memset(h, 0, sizeof h); ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fast_stamp), h, sizeof h)); }
    return ORBX_OK;
}
#endif

// ================================================================ K3: quadtree cull (E4)
// ORBextractor::DistributeOctTree (src/ORBextractor.cc:617-915) as a label-propagation problem:
// every point carries the id (= list position) of its leaf; a sweep counts the four children of
// every splitting node with LDS atomics, scans to get the new list positions and relabels the
// points.  Order algebra (validated against the sequential oracle by tests/quadtree_model.py):
//   new list = reverse(children of split nodes in processing order, n1..n4) ++ unsplit nodes;
//   phase 1 processes all nodes with >1 point in list order; phase 2 processes them sorted by
//   (count desc, list position asc) and stops after the split that reaches N leaves.
// One 256-thread workgroup per (level, image).
extern __shared__ __align__(16) unsigned char tree_smem[];

// NT threads per workgroup: 256 for batches (many (level, image) workgroups co-resident per CU), 1024 when a launch has
// fewer workgroups than the chip has CUs (a single frame: the longest workgroup's latency chain IS the kernel time)
// TAB_LDS: node tables in LDS (every ORB-SLAM2 configuration) -- a compile-time fact, so that their accesses are ds_ instructions
// and LDS atomics; behind a pointer chosen at run time they were FLAT instructions (300 per wave through the vector-memory path).
template <int NT, bool TAB_LDS>
__device__ __forceinline__ void tree_body(const Geom *__restrict__ g, const int *__restrict__ cell_cnt,
                                              const uint32_t *__restrict__ cand, uint32_t *__restrict__ g_pts,
                                              uint16_t *__restrict__ g_nid, int *__restrict__ lvl_cnt,
                                              uint32_t *__restrict__ lvl_kp, int lds_pts_cap, int *__restrict__ err_flag,
                                              unsigned char *__restrict__ g_tab, long long g_tab_stride, const uint32_t *__restrict__ cand_prim, int reg_pts);

#ifndef ORBX_TREE_WPE
#define ORBX_TREE_WPE 6     // waves per SIMD the 256-thread form is compiled for (= workgroups per CU): 79 VGPRs, no spills; the LDS (25 KB per
                            // workgroup with the overflow array) holds six anyway (7: 72 VGPRs + 20 bytes of scratch, 0.094 against 0.091 ms)
#endif
// (the 1024-thread form has four waves per SIMD by construction: with the 256-thread form's register cap it spilled)
template <int NT, bool TAB_LDS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 1024 ? 4 : ORBX_TREE_WPE, NT == 1024 ? 4 : ORBX_TREE_WPE))) void k_tree(const Geom *__restrict__ g, const int *__restrict__ cell_cnt,
                                              const uint32_t *__restrict__ cand, uint32_t *__restrict__ g_pts,
                                              uint16_t *__restrict__ g_nid, int *__restrict__ lvl_cnt,
                                              uint32_t *__restrict__ lvl_kp, int lds_pts_cap, int *__restrict__ err_flag,
                                              unsigned char *__restrict__ g_tab, long long g_tab_stride, const uint32_t *__restrict__ cand_prim, int reg_pts)
{
#ifdef ORBX_DIAG_TREE_TWICE     // experiment: the whole tree a second time on the same input (idempotent) -- the second pass runs from a warm instruction cache
    for (int rep = 0; rep < 2; rep++) {
        if (rep) __syncthreads();
        tree_body<NT, TAB_LDS>(g, cell_cnt, cand, g_pts, g_nid, lvl_cnt, lvl_kp, lds_pts_cap, err_flag, g_tab, g_tab_stride, cand_prim, reg_pts);
    }
#else
    tree_body<NT, TAB_LDS>(g, cell_cnt, cand, g_pts, g_nid, lvl_cnt, lvl_kp, lds_pts_cap, err_flag, g_tab, g_tab_stride, cand_prim, reg_pts);
#endif
}

template <int NT, bool TAB_LDS>
__device__ __forceinline__ void tree_body(const Geom *__restrict__ g, const int *__restrict__ cell_cnt,
                                              const uint32_t *__restrict__ cand, uint32_t *__restrict__ g_pts,
                                              uint16_t *__restrict__ g_nid, int *__restrict__ lvl_cnt,
                                              uint32_t *__restrict__ lvl_kp, int lds_pts_cap, int *__restrict__ err_flag,
                                              unsigned char *__restrict__ g_tab, long long g_tab_stride, const uint32_t *__restrict__ cand_prim, int reg_pts)
{
    constexpr int NB = ORBX_NODE_BITS, NMASK = (1 << NB) - 1;
    // points per thread in the register form: 12 on 256 threads (3072 per level); the 1024-thread form of single frames takes 4 (4096: a
    // textured 1241 x 376 level 0 has ~3300 candidates, and a level beyond the register capacity walks its points in the HBM scratch)
    constexpr int REG_PTS = NT == 1024 ? ORBX_TREE_REG_PTS_BIG : ORBX_TREE_REG_PTS;
    constexpr int RP = REG_PTS / NT;
    // x = image, y = level: workgroups are dealt to the 8 XCDs by linear id % 8, so every XCD gets the same mix of
    // levels (x = level would put all level-0 trees, the longest barrier chains, on one XCD), heaviest level first
    const int l = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const LevelGeom &L = g->lv[l];
    const int cap = g->max_node_cap; // multiple of 4
    // node tables (76 B per leaf): in LDS when they fit beside the points (every ORB-SLAM2 configuration: <= ~1900 leaves per
    // level), else in this workgroup's slice of an HBM workspace (any nfeatures the reference accepts up to the 14-bit node
    // id: __syncthreads orders the workgroup's own global stores and loads, the same code runs on either memory)
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const size_t tab_bytes = (size_t)cap * 76;
    unsigned char *tab;
    if constexpr (TAB_LDS) tab = tree_smem; else tab = g_tab + (size_t)wg * (size_t)g_tab_stride;
    int *cnt = reinterpret_cast<int *>(tab);
    int *cnt_n = cnt + cap;
    uint2 *box = reinterpret_cast<uint2 *>(cnt_n + cap);
    uint2 *box_n = box + cap;
    int *cc = reinterpret_cast<int *>(box_n + cap); // [4*cap] child counts, then child positions (16-byte aligned)
    int *cc_n = cc + 4 * cap;                       // the next table's child counts
    int *a1 = cc_n + 4 * cap;                       // processing rank of split nodes
    int *a2 = a1 + cap;                             // children per processed node -> S offsets
    int *a3 = a2 + cap;                             // unsplit flags -> ranks
    int *a4 = a3 + cap;                             // phase-2 gains
    int *ncarr = a4 + cap;                          // non-empty children per node (0 = not split)
    // [max_cells_level + 4], always LDS; in the register form it sits behind the staging area the gather uses (which aliases the
    // node tables: they are not live yet)
    int *cellpref = !TAB_LDS ? reinterpret_cast<int *>(tree_smem)
                             : reinterpret_cast<int *>(tree_smem + (reg_pts && tab_bytes < (size_t)REG_PTS * 4 ? (size_t)REG_PTS * 4 : tab_bytes));
    uint32_t *lpts = reinterpret_cast<uint32_t *>(cellpref + ((g->max_cells_level + 4) & ~3));
    uint16_t *lnid = reinterpret_cast<uint16_t *>(lpts + lds_pts_cap);
    __shared__ int s_w[2 * (NT / 64)];  // wave totals of the block scans; the one-barrier sweeps alternate between the halves
    __shared__ int s_acc;
    __shared__ int s_acc2[2];           // n_to_expand of the one-barrier sweeps, alternating (the idle one is zeroed a sweep ahead)

    int *out_cnt = lvl_cnt + (long long)b * ORBX_MAX_LEVELS + l;
#ifdef ORBX_DIAG
    unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
    unsigned long long _tl_prev = _t_prev; int _tl_n = 0;
#endif
    // ---- gather this level's candidates (cell-row-major, in-cell row-major)
    const int *ccnt = cell_cnt + (long long)b * g->total_cells + L.cell_base;
    // Single-frame form (1024 threads, one cell per thread): the cell's dense candidate record is requested together with its count --
    // its address does not depend on the counts -- so the gather is one global round trip, not two (the tree's time is a chain of such steps)
    constexpr bool PREFETCH = NT == 1024;
    const bool pre = PREFETCH && L.n_cells <= NT;
    uint4 pq0 = make_uint4(0, 0, 0, 0), pq1 = pq0, pq2 = pq0, pq3 = pq0;
    if (pre && tid < L.n_cells) {
        const uint4 *pr = reinterpret_cast<const uint4 *>(cand_prim + ((long long)b * g->total_cells + L.cell_base + tid) * ORBX_CAND_PRIM);
        pq0 = pr[0]; pq1 = pr[1]; pq2 = pr[2]; pq3 = pr[3];
    }
    for (int c = tid; c < L.n_cells; c += NT) cellpref[c] = ccnt[c];
    __syncthreads();
    const int n = lds_excl_scan_nt<NT>(cellpref, L.n_cells, s_w);
    if (n == 0) {
        if (tid == 0) *out_cnt = 0;
        return;
    }
    // Points never move (a point keeps the list position of its leaf as a label), and every pass over them is
    // `for (i = tid; i < n; i += NT)`: with n <= ORBX_TREE_REG_PTS thread tid simply KEEPS its points i = tid + NT * k and their
    // labels in registers -- no LDS for them at all (they were half of the workgroup's LDS, and LDS is what limits the
    // (level, image) workgroups per CU: 4 -> 8), and no LDS round trip per point and sweep.  Bigger levels fall back to arrays
    // (LDS up to lds_pts_cap, else the HBM scratch).
    // (register form with an overflow: a level with up to lds_pts_cap more candidates than the registers hold keeps the excess in a small
    // LDS array -- every textured 1241 x 376 level 0 has 3100-4000 candidates, and a level beyond the capacity walks ALL its points in the
    // HBM scratch)
    const bool in_regs = reg_pts && n <= REG_PTS + lds_pts_cap;
    const int n_over = in_regs && n > REG_PTS ? n - REG_PTS : 0;
    // Which points a thread keeps is free (a point's list index i travels with it); neighbouring LANES take points NT / 64 apart, not
    // neighbours: the list is cell-row-major, neighbours fall into the same quadtree node, and 64 lanes adding to one node's LDS
    // counter serialise (the relabel + classify passes of the first sweeps, 16-64 counters for ~3000 points, were 11 k of 72 k cycles)
    const int pbase = (tid & 63) * (NT / 64) + (tid >> 6);
    uint32_t rp[RP];
    unsigned rn[RP];
#pragma unroll
    for (int k = 0; k < RP; k++) { rp[k] = 0; rn[k] = 0; }
    uint32_t *pts;
    uint16_t *nid;
    if (in_regs) { pts = reinterpret_cast<uint32_t *>(tree_smem); nid = nullptr; }   // staging for the gather only
    else if (n <= lds_pts_cap) { pts = lpts; nid = lnid; }
    else {
        pts = g_pts + (long long)b * g->cand_total + L.cand_off;
        nid = g_nid + (long long)b * g->cand_total + L.cand_off;
    }
    {   // one thread per cell: the copies of different cells are independent loads in flight together
        const uint32_t *src = cand + (long long)b * g->cand_total + L.cand_off;
        for (int c = tid; c < L.n_cells; c += NT) {
            const int beg = cellpref[c], end = c + 1 < L.n_cells ? cellpref[c + 1] : n;
            const uint32_t *s = src + (long long)c * L.cand_cap;
            // the cell's dense 64-byte record as four independent 16-byte loads (an element-wise loop was a chain of dependent
            // load -> store round trips, as long as the fullest cell); only the rare entries beyond it walk the slot block
            const uint4 *pr = reinterpret_cast<const uint4 *>(cand_prim + ((long long)b * g->total_cells + L.cell_base + c) * ORBX_CAND_PRIM);
            const int cn = end - beg;
            if (cn > 0) {
                static_assert(ORBX_CAND_PRIM == 16, "four uint4 per record");
                uint4 q0, q1, q2, q3;
                if (pre) { q0 = pq0; q1 = pq1; q2 = pq2; q3 = pq3; } else { q0 = pr[0]; q1 = pr[1]; q2 = pr[2]; q3 = pr[3]; }
                const uint32_t v[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
                if (in_regs && end > REG_PTS) {     // (part of) the cell lies beyond the register capacity: those points go to the overflow array
#pragma unroll
                    for (int e = 0; e < 16; e++) if (e < cn) { const int i = beg + e; if (i < REG_PTS) pts[i] = v[e]; else lpts[i - REG_PTS] = v[e]; }
                    for (int e = 16; e < cn; e++) { const int i = beg + e; if (i < REG_PTS) pts[i] = s[e]; else lpts[i - REG_PTS] = s[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; e++) if (e < cn) pts[beg + e] = v[e];
                    for (int e = 16; e < cn; e++) pts[beg + e] = s[e];
                }
            }
        }
    }
    if (in_regs) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RP; k++) { const int i = pbase + NT * k; if (i < n) rp[k] = reinterpret_cast<const uint32_t *>(tree_smem)[i]; }
        __syncthreads();        // the staging area becomes the node tables
    }
    // one pass over the points: BODY sees the index i, the packed point p and its label nd (read / write)
    // (the empty asm hides the point's value from loop-invariant code motion: LLVM otherwise extracts x and y of all the thread's points
    // once, keeps those 2 x RP values alive across the sweep loop and spills -- 40 spill stores / 64 reloads in the 256-thread form)
#define FOR_POINTS(...) do { \
        if (in_regs) { \
            _Pragma("unroll") for (int k_ = 0; k_ < RP; k_++) { \
                const int i = pbase + NT * k_; \
                if (i < n) { uint32_t p = rp[k_]; asm volatile("" : "+v"(p)); unsigned nd = rn[k_]; __VA_ARGS__; rn[k_] = nd; } \
            } \
            for (int j_ = tid; j_ < n_over; j_ += NT) { \
                const int i = REG_PTS + j_; (void)i; const uint32_t p = lpts[j_]; (void)p; unsigned nd = lnid[j_]; __VA_ARGS__; lnid[j_] = (uint16_t)nd; \
            } \
        } else { \
            for (int i = tid; i < n; i += NT) { const uint32_t p = pts[i]; (void)p; unsigned nd = nid[i]; __VA_ARGS__; nid[i] = (uint16_t)nd; } \
        } } while (0)
    TSTAMP(0);  // cell counts, prefix, gather
    TLOG(0, n);
    // ---- roots (src/ORBextractor.cc:627-705)
    const int N = L.quota;
    int m;
    // With a handful of roots (3 for a 1241 x 376 level) every point of the level would hit one of 3 LDS addresses: same-address LDS
    // atomics serialise lane by lane, and this pass and the first classification were 27 % of a level-0 tree (36 k cycles).  A thread
    // owns at most 15 points per pass, so it counts them in 4-bit fields of one 64-bit register; the fields are summed over the wave
    // on the DPP path and lane 0 adds each total once: T atomics per wave instead of one per point.
    // points a thread sees per pass, at most (the 4-bit count fields below must hold them)
    const int ppt = in_regs ? (min(n, REG_PTS) + NT - 1) / NT + (n_over + NT - 1) / NT : (n + NT - 1) / NT;
    const bool few_pts_per_thread = ppt <= 15;
    const bool seven_pts_per_thread = ppt <= 7;   // (the 1024-thread form: a thread holds at most 4 + 1 points)
    auto add_packed = [&](unsigned long long acc, int T, int *dst) {
        if (seven_pts_per_thread) {
            // all sixteen fields summed over the wave TOGETHER, widening as the partial sums grow: a field is at most 7, so two lanes'
            // sum still fits its nibble (one DPP step on the packed words), a 16-lane row's fits a byte (three steps on four words of
            // byte fields), the wave's a 16-bit field (the two cross-row steps on eight words).  Lane t then picks field t's total and
            // ONE LDS atomic instruction adds them all (sixteen separate wave sums + atomics were 4 k of a level-0 tree's 60 k cycles)
            unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
            lo += (unsigned)ORBX_DPP((int)lo, 0, 0x111, 0xf, 0xf); hi += (unsigned)ORBX_DPP((int)hi, 0, 0x111, 0xf, 0xf);
            unsigned w[4] = { lo & 0x0F0F0F0Fu, (lo >> 4) & 0x0F0F0F0Fu, hi & 0x0F0F0F0Fu, (hi >> 4) & 0x0F0F0F0Fu };   // fields 0 2 4 6 | 1 3 5 7 | 8 10 12 14 | 9 11 13 15
            unsigned x[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                w[q] += (unsigned)ORBX_DPP((int)w[q], 0, 0x112, 0xf, 0xf);
                w[q] += (unsigned)ORBX_DPP((int)w[q], 0, 0x114, 0xf, 0xe);
                w[q] += (unsigned)ORBX_DPP((int)w[q], 0, 0x118, 0xf, 0xc);
                x[2 * q] = w[q] & 0x00FF00FFu; x[2 * q + 1] = (w[q] >> 8) & 0x00FF00FFu;      // bytes 0 2 | 1 3 of the word
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                x[q] += (unsigned)ORBX_DPP((int)x[q], 0, 0x142, 0xa, 0xf);
                x[q] += (unsigned)ORBX_DPP((int)x[q], 0, 0x143, 0xc, 0xf);
                x[q] = (unsigned)__builtin_amdgcn_readlane((int)x[q], 63);
            }
            // field t: word q = 2 * (t >> 3) + (t & 1), byte bi = (t & 7) >> 1 of it -> x[2 q + (bi & 1)], 16-bit slot bi >> 1
            const int t = tid & 63, q = 2 * ((t >> 3) & 1) + (t & 1), bi = (t & 7) >> 1, xi = 2 * q + (bi & 1);
            unsigned sel = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) sel = xi == j ? x[j] : sel;
            const int val = (int)((sel >> (16 * (bi >> 1))) & 0xFFFFu);
            if (t < T && val) atomicAdd(&dst[t], val);
            return;
        }
        for (int t = 0; t < T; t++) {                                   // T <= 16, wave-uniform
            const int s = wave_sum((int)((acc >> (4 * t)) & 15ull));
            if ((tid & 63) == 0 && s) atomicAdd(&dst[t], s);
        }
    };
    // ---- sweeps.  Invariant at the top of the loop: cc[0..4m) holds the child counts of the current
    // table (cnt/box) and every point label is (node id | child << NB).
    auto classify = [&](int id, uint32_t p, const int *cn, const uint2 *bx_tab, int *cct) -> int {
        int c = 0;
        if (cn[id] > 1) {
            const uint2 bx = bx_tab[id];
            const int x0 = bx.x & 0xFFFF, x1 = bx.x >> 16, y0 = bx.y & 0xFFFF, y1 = bx.y >> 16;
            const int x = p & 0xFFF, y = (p >> 12) & 0xFFF;
            const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1; // ceil(d/2), DivideNode :553-554
            c = (x >= x0 + hx ? 1 : 0) + (y >= y0 + hy ? 2 : 0);
            atomicAdd(&cct[id * 4 + c], 1);
        }
        return c;
    };
    // (1024-thread form only: with 256 threads, twelve points each, the extra arithmetic per point costs more than the barriers it saves:
    // 0.126 against 0.116 ms per 512-image launch)
    if (NT == 1024 && L.n_ini <= 4 && few_pts_per_thread) {
        // Up to four roots (every usual aspect ratio: 3 for 1241 x 376, 1 for 640 x 480): counting them, dropping the empty ones and
        // the first classification are two barrier-to-barrier steps.  The root counts go to a scratch array that every thread then
        // reads whole, so the id of a root (= non-empty roots before it) and the table size need no scan, and a root's box is
        // arithmetic on its index, so the first classification does not wait for the table entries other threads write.
        if (tid < 16) { a3[tid] = 0; cc[tid] = 0; }
        __syncthreads();
        TLOG(2, 0);
        {
            unsigned long long acc = 0;
            FOR_POINTS({
                int r = (int)((float)(p & 0xFFF) / L.hx);
                r = r < 0 ? 0 : r >= L.n_ini ? L.n_ini - 1 : r;
                acc += 1ull << (4 * r);
                nd = (unsigned)r;
            });
            TLOG(3, 0);
            add_packed(acc, L.n_ini, a3);
            TLOG(4, 0);
        }
        __syncthreads();
        TLOG(5, 0);
        const int4 rc = *reinterpret_cast<const int4 *>(a3);          // counts of roots 0..3 (zero beyond n_ini)
        const unsigned nz = (rc.x > 0 ? 1u : 0u) | (rc.y > 0 ? 2u : 0u) | (rc.z > 0 ? 4u : 0u) | (rc.w > 0 ? 8u : 0u);
        m = __popc(nz);
        if (tid < L.n_ini && ((nz >> tid) & 1u)) {
            const int id = __popc(nz & ((1u << tid) - 1u));
            const unsigned x0 = (unsigned)(int)(L.hx * (float)tid), x1 = (unsigned)(int)(L.hx * (float)(tid + 1));
            box[id] = make_uint2(x0 | (x1 << 16), 0u | ((unsigned)L.tree_h << 16));
            cnt[id] = tid == 0 ? rc.x : tid == 1 ? rc.y : tid == 2 ? rc.z : rc.w;
        }
        unsigned long long acc = 0;
        FOR_POINTS({
            const int r = (int)nd, id = __popc(nz & ((1u << r) - 1u));
            const int rcnt = r == 0 ? rc.x : r == 1 ? rc.y : r == 2 ? rc.z : rc.w;
            int c = 0;
            if (rcnt > 1) {
                const int x0 = (int)(L.hx * (float)r), x1 = (int)(L.hx * (float)(r + 1)), y0 = 0, y1 = L.tree_h;
                const int x = p & 0xFFF, y = (p >> 12) & 0xFFF;
                const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
                c = (x >= x0 + hx ? 1 : 0) + (y >= y0 + hy ? 2 : 0);
                acc += 1ull << (4 * (id * 4 + c));
            }
            nd = (unsigned)(id | (c << NB));
        });
        TLOG(6, 0);
        add_packed(acc, 4 * m, cc);
        TLOG(7, 0);
    } else {
        for (int k = tid; k < L.n_ini; k += NT) cc[k] = 0;
        __syncthreads();
        if (L.n_ini <= 16 && few_pts_per_thread) {
            unsigned long long acc = 0;
            FOR_POINTS({
                int r = (int)((float)(p & 0xFFF) / L.hx);
                r = r < 0 ? 0 : r >= L.n_ini ? L.n_ini - 1 : r;
                acc += 1ull << (4 * r);
                nd = (unsigned)r;
            });
            add_packed(acc, L.n_ini, cc);
        } else {
            FOR_POINTS({
                int r = (int)((float)(p & 0xFFF) / L.hx);
                r = r < 0 ? 0 : r >= L.n_ini ? L.n_ini - 1 : r;
                atomicAdd(&cc[r], 1);
                nd = (unsigned)r;
            });
        }
        __syncthreads();
        for (int k = tid; k < L.n_ini; k += NT) a1[k] = cc[k] > 0;
        __syncthreads();
        m = lds_excl_scan_nt<NT>(a1, L.n_ini, s_w);
        for (int k = tid; k < L.n_ini; k += NT)
            if (cc[k] > 0) {
                const int id = a1[k];
                const unsigned x0 = (unsigned)(int)(L.hx * (float)k), x1 = (unsigned)(int)(L.hx * (float)(k + 1));
                box[id] = make_uint2(x0 | (x1 << 16), 0u | ((unsigned)L.tree_h << 16));
                cnt[id] = cc[k];
            }
        FOR_POINTS({ nd = (unsigned)a1[nd]; });
        __syncthreads();

        for (int k = tid; k < 4 * m; k += NT) cc[k] = 0;
        __syncthreads();
        if (4 * m <= 16 && few_pts_per_thread) {       // the same for the first classification: at most 16 (root, child) counters
            unsigned long long acc = 0;
            FOR_POINTS({
                const int id = (int)nd;
                int c = 0;
                if (cnt[id] > 1) {
                    const uint2 bx = box[id];
                    const int x0 = bx.x & 0xFFFF, x1 = bx.x >> 16, y0 = bx.y & 0xFFFF, y1 = bx.y >> 16;
                    const int x = p & 0xFFF, y = (p >> 12) & 0xFFF;
                    const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
                    c = (x >= x0 + hx ? 1 : 0) + (y >= y0 + hy ? 2 : 0);
                    acc += 1ull << (4 * (id * 4 + c));
                }
                nd = (unsigned)(id | (c << NB));
            });
            add_packed(acc, 4 * m, cc);
        } else {
            FOR_POINTS({ const int id = (int)nd; nd = (unsigned)(id | (classify(id, p, cnt, box, cc) << NB)); });
        }

    }
    bool phase2 = false;
    if (tid < 2) s_acc2[tid] = 0;
    TSTAMP(1);  // roots + first classification
    TLOG(1, m);
    for (int sweep = 0;; sweep++) {
        const int prev = m;
        __syncthreads();
        TLOG(10, m);
        int nsplit = 0, S, U;
        // Phase-1 sweep of a table that fits one node per thread (every ORB-SLAM2 setting on 1024 threads, the small levels on 256):
        // node k stays with thread k from its child counts to its children's table entries, so the split flags, the packed scan input
        // and the scan result never go through LDS, and the sweep needs three workgroup barriers instead of six (a level-0 tree of a
        // single frame is a chain of ~50 barrier-to-barrier steps of ~0.4 us each: that chain, not the work, is its 35 us).
        const bool one_per_thread = NT == 1024 && !phase2 && m <= NT;   // (neutral at 256 threads, and its live values push that form into register spills)
        int my_nc = 0, my_run = 0;
        if (one_per_thread) {
            const int par = sweep & 1;
            int v = 0;
            if (tid < m) {
                const int sp = cnt[tid] > 1;
                my_nc = sp ? (cc[4 * tid] > 0) + (cc[4 * tid + 1] > 0) + (cc[4 * tid + 2] > 0) + (cc[4 * tid + 3] > 0) : 0;
                v = sp ? my_nc : (1 << 16);
            }
            const int inc = wave_incl_scan(v);
            if ((tid & 63) == 63) s_w[par * (NT / 64) + (tid >> 6)] = inc;
            __syncthreads();
            int base = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < NT / 64; i++) { const int t = s_w[par * (NT / 64) + i]; if (i < (tid >> 6)) base += t; tot += t; }
            my_run = base + inc - v;
            S = tot & 0xFFFF; U = tot >> 16;
        } else
        if (!phase2) {
            // processing order == list order: one packed scan gives both the children offset of every split
            // node (low 16 bits) and the rank of every unsplit node (high 16 bits)
            for (int k = tid; k < m; k += NT) {
                const int sp = cnt[k] > 1;
                const int ncv = sp ? (cc[4 * k] > 0) + (cc[4 * k + 1] > 0) + (cc[4 * k + 2] > 0) + (cc[4 * k + 3] > 0) : 0;
                ncarr[k] = ncv;
                a2[k] = sp ? ncv : (1 << 16);
            }
            if (tid == 0) s_acc = 0;
            __syncthreads();
            const int tot = lds_excl_scan_nt<NT>(a2, m, s_w);
            S = tot & 0xFFFF; U = tot >> 16;
        } else if (m <= 64) {
            // Phase 2 on a table of at most 64 nodes (1000-feature settings reach it at m = 64; it is the last sweep of nearly every
            // tree): ONE wave orders the nodes, lane = node, everything in registers and DPP -- rank by (count desc, list position
            // asc), gains in rank order, how many splits reach N, children offsets, unsplit ranks -- and publishes the tables the
            // apply step reads.  The workgroup form below takes sixteen barrier-to-barrier steps for the same thing (18.8 k of a
            // level-0 tree's 72 k cycles).
            // (the 64 x 64 comparisons of the ranks are dealt to the workgroup's waves first, 64 / waves "other nodes" each -- in one
            // wave they were a 64-step dependent loop --, summed with one LDS atomic per lane and wave)
            if (NT != 1024) {       // (256 threads: the two extra barriers cost more than the shorter loop saves: 0.137 against 0.119 ms per 512-image launch)
                if (tid < 64) {
                    const int k = tid, ck = k < m ? cnt[k] : 0;
                    int r = 0;
                    for (int k2 = 0; k2 < m; k2++) { const int c2 = __builtin_amdgcn_readlane(ck, k2); r += (c2 > ck) || (c2 == ck && k2 < k); }
                    a1[k] = r;
                }
            } else {
                constexpr int NWV = NT / 64, PER = (64 + NWV - 1) / NWV;
                const int k = tid & 63, wv_ = tid >> 6;
                if (tid < 64) a1[tid] = 0;
                __syncthreads();
                const int ck = k < m ? cnt[k] : 0;
                int part = 0;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const int k2 = wv_ * PER + j;
                    if (k2 < m) {                               // wave-uniform
                        const int c2 = __builtin_amdgcn_readlane(ck, k2 & 63);
                        part += (c2 > ck) || (c2 == ck && k2 < k);
                    }
                }
                if (part && ck > 1) atomicAdd(&a1[k], part);
                __syncthreads();
            }
            if (tid < 64) {
                const int k = tid;
                const int ck = k < m ? cnt[k] : 0;
                const bool cand = ck > 1;
                const int ncv = cand ? (cc[4 * k] > 0) + (cc[4 * k + 1] > 0) + (cc[4 * k + 2] > 0) + (cc[4 * k + 3] > 0) : 0;
                const int r = a1[k];
                const int ncand = __popcll(__ballot(cand));
                if (cand) a4[r] = ncv - 1;                      // gains in processing (rank) order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int g_r = k < ncand ? a4[k] : 0;          // lane = rank from here
                const int px = wave_incl_scan(g_r) - g_r;
                const int less = __popcll(__ballot(k < ncand && prev + px + g_r < N));
                nsplit = min(ncand, less + 1);
                const int nc_r = k < nsplit ? g_r + 1 : 0;      // children of the node of rank k, if it splits
                const int inc = wave_incl_scan(nc_r);
                if (k < nsplit) a2[k] = inc - nc_r;
                const int s_tot = __builtin_amdgcn_readlane(inc, 63);
                const bool sp = cand && r < nsplit;             // lane = node again
                const int uns = k < m && !sp ? 1 : 0;
                const int uinc = wave_incl_scan(uns);
                if (k < m) { ncarr[k] = sp ? ncv : 0; a1[k] = cand ? r : -1; a3[k] = uinc - uns; }
                if (k == 0) { s_w[0] = s_tot; s_w[1] = __builtin_amdgcn_readlane(uinc, 63); }
            }
            __syncthreads();
            S = s_w[0]; U = s_w[1];
            // (s_w[0..1] are next written by a block scan or by this branch, both behind the barrier at the end of the apply step)
        } else {
            // processing order: count desc, list position asc (src/ORBextractor.cc:832-834 with the
            // address tie-break defined as "created later first" == nearer the list front)
            if (tid == 0) s_acc = 0;
            __syncthreads();
            int ncand_local = 0;
            for (int k = tid; k < m; k += NT) {
                const int ck = cnt[k];
                int r = -1, ncv = 0;
                if (ck > 1) {
                    // rank = nodes that come before this one: four counts per LDS read, the reads independent of each other (one count
                    // per dependent read made this loop 15 k of a level-0 tree's 72 k cycles)
                    r = 0;
                    const int m4 = m & ~3;
                    for (int k2 = 0; k2 < m4; k2 += 4) {
                        const int4 c4 = *reinterpret_cast<const int4 *>(cnt + k2);
                        r += ((c4.x > ck) || (c4.x == ck && k2 < k)) + ((c4.y > ck) || (c4.y == ck && k2 + 1 < k)) +
                             ((c4.z > ck) || (c4.z == ck && k2 + 2 < k)) + ((c4.w > ck) || (c4.w == ck && k2 + 3 < k));
                    }
                    for (int k2 = m4; k2 < m; k2++) {
                        const int c2 = cnt[k2];
                        r += (c2 > ck) || (c2 == ck && k2 < k);
                    }
                    ncv = (cc[4 * k] > 0) + (cc[4 * k + 1] > 0) + (cc[4 * k + 2] > 0) + (cc[4 * k + 3] > 0);
                    ncand_local++;
                }
                a1[k] = r;
                ncarr[k] = ncv;
            }
            if (ncand_local) atomicAdd(&s_acc, ncand_local);
            __syncthreads();
            const int ncand = s_acc;
            for (int k = tid; k < m; k += NT)
                if (a1[k] >= 0) { a2[a1[k]] = ncarr[k] - 1; a4[a1[k]] = ncarr[k] - 1; }
            __syncthreads();
            if (tid == 0) s_acc = 0;
            lds_excl_scan_nt<NT>(a2, ncand, s_w);
            int less = 0;
            for (int r = tid; r < ncand; r += NT) less += (prev + a2[r] + a4[r] < N);
            if (less) atomicAdd(&s_acc, less);
            __syncthreads();
            nsplit = min(ncand, s_acc + 1);
            __syncthreads();
            for (int k = tid; k < m; k += NT) {
                const bool sp = a1[k] >= 0 && a1[k] < nsplit;
                if (!sp) ncarr[k] = 0;
                a3[k] = !sp;
            }
            if (tid == 0) s_acc = 0;
            __syncthreads();
            for (int k = tid; k < m; k += NT)
                if (ncarr[k] > 0) a2[a1[k]] = ncarr[k];
            __syncthreads();
            S = lds_excl_scan_nt<NT>(a2, nsplit, s_w);
            U = lds_excl_scan_nt<NT>(a3, m, s_w);
        }
        if (S + U > cap) { // cannot happen (SURVEY.md A.4 bound); never write out of bounds
            if (tid == 0) { atomicExch(err_flag, 1); *out_cnt = 0; }
            return;
        }
        TSTAMP(2);  // order / scans of the sweep
        TLOG(phase2 ? 12 : 11, S + U);
#ifdef ORBX_DIAG
        if (phase2 && blockIdx.y == 0 && tid == 0) atomicAdd(&g_tree_stamp[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + 6], 1ull);
#endif
        // ---- apply: build the next table, turn cc into child positions, zero the next table's counters
        int expand_local = 0;
        for (int k = tid; k < m; k += NT) {
            if ((one_per_thread ? my_nc : ncarr[k]) > 0) {
                const uint2 bx = box[k];
                const int x0 = bx.x & 0xFFFF, x1 = bx.x >> 16, y0 = bx.y & 0xFFFF, y1 = bx.y >> 16;
                const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
                int pos = S - 1 - (one_per_thread ? (my_run & 0xFFFF) : phase2 ? a2[a1[k]] : (a2[k] & 0xFFFF));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int q = cc[4 * k + c];
                    if (q > 0) {
                        const unsigned cx0 = (c & 1) ? x0 + hx : x0, cx1 = (c & 1) ? x1 : x0 + hx;
                        const unsigned cy0 = (c & 2) ? y0 + hy : y0, cy1 = (c & 2) ? y1 : y0 + hy;
                        box_n[pos] = make_uint2(cx0 | (cx1 << 16), cy0 | (cy1 << 16));
                        cnt_n[pos] = q;
                        reinterpret_cast<int4 *>(cc_n)[pos] = make_int4(0, 0, 0, 0);
                        expand_local += q > 1;
                        cc[4 * k + c] = pos;
                        pos--;
                    }
                }
            } else {
                const int pos = S + (one_per_thread ? (my_run >> 16) : phase2 ? a3[k] : (a2[k] >> 16));
                box_n[pos] = box[k];
                cnt_n[pos] = cnt[k];
                reinterpret_cast<int4 *>(cc_n)[pos] = make_int4(0, 0, 0, 0);
                cc[4 * k] = cc[4 * k + 1] = cc[4 * k + 2] = cc[4 * k + 3] = pos;
            }
        }
        if (expand_local) atomicAdd(one_per_thread ? &s_acc2[sweep & 1] : &s_acc, expand_local);
        __syncthreads();
        m = S + U;
        const int n_to_expand = one_per_thread ? s_acc2[sweep & 1] : s_acc;
        if (tid == 0) s_acc2[(sweep & 1) ^ 1] = 0;    // the other accumulator: next used after the next sweep's barriers
        const bool done = m >= N || m == prev;                     // :803-806, :883-884
        if (!phase2 && !done && m + 3 * n_to_expand > N) phase2 = true; // :814
        if (done) {
            // final relabel fused with "one keypoint per leaf: max response, first in list order wins ties" (:895-912): the next
            // table's counters (cc_n) were zeroed by the apply step above for every leaf, so they serve as the per-leaf maxima
            // without a clearing pass and its barrier
            unsigned *bestn = reinterpret_cast<unsigned *>(cc_n);
            FOR_POINTS({
                const int v = (int)nd;
                nd = (unsigned)cc[(v & NMASK) * 4 + (v >> NB)];
                atomicMax(&bestn[nd], ((p >> 24) << 24) | (0xFFFFFFu - (unsigned)i));
            });
            __syncthreads();
            break;
        }
        TSTAMP(3);  // apply
        TLOG(13, m);
        // ---- relabel fused with the next sweep's classification (one pass over the points)
        FOR_POINTS({
            const int v = (int)nd;
            const int id = cc[(v & NMASK) * 4 + (v >> NB)];
            nd = (unsigned)(id | (classify(id, p, cnt_n, box_n, cc_n) << NB));
        });
        { int *t = cnt; cnt = cnt_n; cnt_n = t; }
        { uint2 *t = box; box = box_n; box_n = t; }
        { int *t = cc; cc = cc_n; cc_n = t; }
        TSTAMP(4);  // relabel + classify
        TLOG(14, m);
    }

    // ---- one keypoint per leaf: max response, first in list order wins ties (:895-912)
    const unsigned *best = reinterpret_cast<const unsigned *>(cc_n);
    uint32_t *okp = lvl_kp + (long long)b * g->kp_total + L.kp_off;
    if (in_regs) {      // the winner of a leaf is written by the thread that holds it
        FOR_POINTS({
            const int k = (int)(nd & NMASK);
            if ((best[k] & 0xFFFFFFu) == 0xFFFFFFu - (unsigned)i && k < L.kp_cap) {
                const unsigned x = (p & 0xFFF) + ORBX_MIN_BORDER, y = ((p >> 12) & 0xFFF) + ORBX_MIN_BORDER;
                okp[k] = x | (y << 12) | (p & 0xFF000000u);
            }
        });
    } else {
        for (int k = tid; k < m; k += NT) {
            const uint32_t p = pts[0xFFFFFFu - (best[k] & 0xFFFFFFu)];
            const unsigned x = (p & 0xFFF) + ORBX_MIN_BORDER, y = ((p >> 12) & 0xFFF) + ORBX_MIN_BORDER;
            if (k < L.kp_cap) okp[k] = x | (y << 12) | (p & 0xFF000000u);
        }
    }
#undef FOR_POINTS
    TSTAMP(5);  // final relabel, best per leaf, output
    TLOG(20, m);
#ifdef ORBX_DIAG
    if (blockIdx.y == 0 && tid == 0) atomicAdd(&g_tree_stamp[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + 7], 1ull);
#endif
    if (tid == 0) *out_cnt = min(m, L.kp_cap);
}

// ================================================================ K4: orientation + blur + rBRIEF (E5-E8)
// One wave per keypoint.  The 43x43 unblurred patch is staged in LDS with BORDER_REFLECT_101 at
// the image edge (the reference blurs a clone of the level, src/ORBextractor.cc:1312-1314), the
// intensity centroid is taken on it (IC_Angle, :83-111), the 7x7 sigma=2 fixed-point Gaussian is
// applied to the patch only (never materialising the blurred level; its row pass on the whole patch, its
// column pass only at the 512 steered sample positions), and the 256 pairs are compared with one ballot
// per 64 pairs (computeOrbDescriptor, :116-157).
// Launch constants of k_desc by value (kernel-argument segment, scalar loads that depend on nothing): the level of a slot is
// found by comparing against kp_off[] in registers, and only then one dependent fetch (the level's record) remains before the
// patch address is known.  Fetching them through the Geom pointer was a chain of dependent scalar loads at the start of every
// wave, during which the wave already holds its LDS.
struct DescLevel { int w, h, pitch, kp_off; long long pyr_off; float scale; int patch_size; };
struct DescArgs {
    int nlevels, kp_total;
    unsigned gauss;                   // taps g0 | g1 << 8 | g2 << 16 | g3 << 24 of the handle's 7-tap kernel (symmetric; orbx_gaussian_taps)
    int kp_off[ORBX_MAX_LEVELS];      // first staging slot of level i; INT_MAX for i >= nlevels
    DescLevel lv[ORBX_MAX_LEVELS];
};

// NL = 8 or ORBX_MAX_LEVELS: the level search and the count sums below are unrolled over NL levels (ORB-SLAM2 uses 8)
// The row table of Frame::ComputeStereoMatches (vRowIndices, src/Frame.cc:584-604) as a by-product of the extraction: it depends only
// on the keypoints' rows, octaves and columns, which the quadtree has already fixed, so ONE extra wave per image builds it inside the
// k_desc launch while the other waves compute descriptors (a launch of its own, k_stereo_prep, was 10 us of a single frame's 124 us
// chain).  orbx_stereo_match_batch_device uses it when its caller says so (ORBX_ROWTAB_OF_EXTRACTION: the keypoint buffer still holds what this extraction wrote); any other caller
// of the stereo matcher still gets k_stereo_prep.  Layout (see orbx_stereo.hip): CSR by the keypoint's centre row, row_off[rows + 1],
// one entry (iR | octave << 16, x, minr | maxr << 16, 0) per keypoint.
struct RowTabArgs { int *row_off; uint4 *entries; int ent_cap, rows, on, pad; };
#define ORBX_ROWTAB_MAX_ROWS 600    // a 256-byte level table + two int arrays of rows + 4 entries in k_desc's 5096 bytes of LDS

template <int NL>
__device__ __forceinline__ void desc_rowtab(const DescArgs &da, const int *__restrict__ lc, const uint32_t *__restrict__ kp_img, int cap,
                                            const RowTabArgs &rt, int b, int *cnt, int *cur, int4 *lvtab)
{
    // The wave is alone on its critical path (it must not outlast the descriptor waves of its launch, ~12 us for a single frame), so
    // everything is arranged for few dependent steps: all staging slots are fetched at once (CH per lane, in registers for both
    // passes), the per-level constants come from one LDS read per slot instead of an 8-way select, one LDS atomic per keypoint and pass.
    const int lane = threadIdx.x, rows = rt.rows;
    constexpr int CH = 24;              // 1536 staging slots per trip: every ORB-SLAM2 setting up to ~1400 features in one
    uint32_t pk[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) { const int s = 64 * k + lane; pk[k] = s < da.kp_total ? kp_img[s] : 0u; }
    for (int i = lane; i < rows; i += 64) { cnt[i] = 0; cur[i] = 0; }
    if (lane < NL) {                    // per level: keypoints kept, output index of its first one (= counts of the lower levels), first slot, scale
        int off = 0, c = 0, ko = 0;
        float sc = 1.0f;
#pragma unroll
        for (int i = 0; i < NL; i++) {      // (static indices into the kernel-argument struct: a lane-indexed access would go through scratch)
            const int ci = lc[i];
            if (i < lane) off += ci;
            if (i == lane) { c = ci; ko = da.lv[i].kp_off; sc = da.lv[i].scale; }
        }
        lvtab[lane] = make_int4(c, off, ko, __float_as_int(sc));
    }
    __syncthreads();                    // a one-wave workgroup: orders the LDS passes
    // A slot of level l, position j is output index off[l] + j; x, y = (float)x_l * scale_l, band radius 2 * scale_l (:588-596):
    // exactly the floats k_desc writes into the keypoint record and k_stereo_prep reads back from it.
#define FOR_KEYPOINTS(RELOAD, ...) do { \
        for (int base_ = 0; base_ < da.kp_total; base_ += 64 * CH) { \
            if ((RELOAD) || base_) { \
                _Pragma("unroll") for (int k_ = 0; k_ < CH; k_++) { const int s_ = base_ + 64 * k_ + lane; pk[k_] = s_ < da.kp_total ? kp_img[s_] : 0u; } \
            } \
            _Pragma("unroll") for (int k_ = 0; k_ < CH; k_++) { \
                const int s_ = base_ + 64 * k_ + lane; \
                if (base_ + 64 * k_ < da.kp_total) {    /* wave-uniform */ \
                    int l_ = 0; \
                    _Pragma("unroll") for (int i_ = 1; i_ < NL; i_++) l_ += s_ >= da.kp_off[i_]; \
                    const int4 lv_ = lvtab[l_]; \
                    const float sc_ = __int_as_float(lv_.w); \
                    const int j_ = s_ - lv_.z, ir = lv_.y + j_; \
                    if (s_ < da.kp_total && j_ < lv_.x && ir < cap && ir < rt.ent_cap) { \
                        const uint32_t p_ = pk[k_]; \
                        float fx = (float)(int)(p_ & 0xFFF), fy = (float)(int)((p_ >> 12) & 0xFFF); \
                        if (l_ != 0) { fx *= sc_; fy *= sc_; } \
                        const int crow = min(max((int)floorf(fy), 0), rows - 1); \
                        const int oct = l_; (void)fx; (void)oct; (void)ir; (void)sc_; \
                        __VA_ARGS__; \
                    } \
                } \
            } \
        } } while (0)
    FOR_KEYPOINTS(false, { atomicAdd(&cnt[crow], 1); });
    __syncthreads();
    int carry = 0;
    int *ro = rt.row_off + (long long)b * (rows + 1);
    for (int base = 0; base < rows; base += 64) {       // exclusive scan of the row counts by the wave
        const int i = base + lane, v = i < rows ? cnt[i] : 0;
        const int inc = wave_incl_scan(v);
        if (i < rows) { cnt[i] = carry + inc - v; ro[i] = carry + inc - v; }
        carry += __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 0) ro[rows] = carry;
    __syncthreads();
    uint4 *en = rt.entries + (long long)b * rt.ent_cap;
    FOR_KEYPOINTS(da.kp_total > 64 * CH, {
        const float r_ = 2.0f * sc_;
        const int maxr = min((int)ceilf(fy + r_), rows - 1), minr = max((int)floorf(fy - r_), 0);
        en[cnt[crow] + atomicAdd(&cur[crow], 1)] = make_uint4((unsigned)ir | ((unsigned)oct << 16), __float_as_uint(fx),
                                                              (unsigned)minr | ((unsigned)maxr << 16), 0u);
    });
#undef FOR_KEYPOINTS
}

template <int NL>
__global__ __launch_bounds__(64) void k_desc(const DescArgs da, PyrRef pr, const int *__restrict__ lvl_cnt,
                                             const uint32_t *__restrict__ lvl_kp, orbx_keypoint *__restrict__ out_kps,
                                             uint8_t *__restrict__ out_desc, int *__restrict__ out_n, int cap, int nimg, const RowTabArgs rt,
                                             const int *__restrict__ err_flag, int *__restrict__ flag_out)
{
    // LDS pitches: raw bytes (11 dwords per row), row-pass u16 (column-major, 43 rows per column, 37 columns).  1908 + 3188 bytes
    // round to 5120 = 160 KB / 32: the CU holds its maximum of 32 waves (the kernel is latency bound: with 5600 bytes, 29 waves
    // per CU, it ran 3 % slower; every KB more costs 7 %)
    constexpr int RP = 44, HR = 43;
    constexpr int RAW_BYTES = 43 * RP + 16;              // 1908
    __shared__ __align__(16) uint8_t desc_smem[RAW_BYTES + (37 * HR + 3) * 2];
    uint8_t *raw = desc_smem;
    uint16_t *hb = reinterpret_cast<uint16_t *>(desc_smem + RAW_BYTES);   // + the zero-tap row "43" of the last column, read as part of a dword
    static_assert(RAW_BYTES % 4 == 0 && sizeof(desc_smem) >= 16 * ORBX_MAX_LEVELS + 2 * (ORBX_ROWTAB_MAX_ROWS + 4) * sizeof(int), "row table workspace");
    // Workgroups are dealt round-robin over the 8 XCDs (linear id % 8, speed only): XCD x walks the images x, x + 8, x + 16, ...
    // one after the other, so the patches its waves fetch at any time come from one or two images (1.4 MB of pyramid each)
    // instead of from every image in flight on the chip: the per-XCD L2 (4 MB) then holds them
    const int lane = threadIdx.x;
    // grid = (8 * kp_total, ceil(images / 8)): blockIdx.x = 8 * slot + XCD, blockIdx.y = group of eight images; the linear
    // workgroup id (dispatch order) then has the XCD in its low three bits and the slot running fastest within an XCD
    // (with the row table on, slot "-1" -- the first workgroups dispatched -- is the table wave of each image)
    // A launch of fewer than eight images (a single stereo frame: two) has no empty XCD columns in its grid: blockIdx.x = xg * slot +
    // image, xg = min(8, images) -- dispatching the 6 800 empty workgroups of an 8-wide grid took longer than the 2 000 waves that
    // had work (their starts spread over 6.5 us).
    const unsigned xg = gridDim.y == 1 && nimg < 8 ? (unsigned)nimg : 8u;
    const unsigned sx = xg == 8 ? blockIdx.x >> 3 : blockIdx.x / xg;
    const int slot = (int)sx - rt.on, b = (int)(blockIdx.y * 8u + (blockIdx.x - sx * xg));
    if (b >= nimg) return;
    if (slot < 0) {
        desc_rowtab<NL>(da, lvl_cnt + (long long)b * ORBX_MAX_LEVELS, lvl_kp + (long long)b * da.kp_total, cap, rt, b,
                        reinterpret_cast<int *>(desc_smem) + 4 * ORBX_MAX_LEVELS, reinterpret_cast<int *>(desc_smem) + 4 * ORBX_MAX_LEVELS + ((rt.rows + 4) & ~3),
                        reinterpret_cast<int4 *>(desc_smem));
        return;
    }
    int l = 0;
#pragma unroll
    for (int i = 1; i < NL; i++) l += slot >= da.kp_off[i];
    const DescLevel L = da.lv[l];
    const int *lc = lvl_cnt + (long long)b * ORBX_MAX_LEVELS;   // rows of ORBX_MAX_LEVELS counts, zero beyond nlevels
    // the slot's packed keypoint is fetched together with the level counts (its address does not depend on them):
    // one global round trip less on the critical path of every wave; slots past the level's count hold stale data
    // that is never used
#ifdef ORBX_DIAG
    unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
#endif
    SPAN_BEGIN();
    const uint32_t p = lvl_kp[(long long)b * da.kp_total + slot];
    // the lane's four pattern words (lane-indexed constant data = vector loads) are requested here, with the first
    // memory round trip, not in the sampling phase where they would cost a round trip of their own
    uint32_t pat4[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) pat4[jj] = c_pat4[lane + 64 * jj];
    const uint4 omask = c_omask[lane];
    int off = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { const int c = lc[i]; off += i < l ? c : 0; total += c; }
    if (slot == 0 && lane == 0) {
        out_n[b] = total < cap ? total : cap;
        if (flag_out && b == 0) *flag_out = *err_flag;     // (pipelined frames: the quadtree's error flag rides in the frame's result block)
    }
    const int j = slot - L.kp_off;
    if (j >= lc[l]) return;
    const int idx = off + j;
    if (idx >= cap) return;
    const int x = p & 0xFFF, y = (p >> 12) & 0xFFF, resp = p >> 24;
    const int pitch = l == 0 ? pr.img0_pitch : L.pitch;
    const uint8_t *img = l == 0 ? pr.img0 + (long long)b * pr.img0_stride : pr.pyr + (long long)b * pr.pyr_stride + L.pyr_off;
#ifdef ORBX_DIAG
    asm volatile("" :: "v"(x), "s"(pitch));
    DSTAMP(5); // prologue: level search, level counts, packed keypoint
#endif
    // ---- stage the 43x43 patch at LDS column 0 of every row (unaligned dword loads: the window phase is a constant,
    // so the realignment shifts below are immediates and the row pass reads three dwords per item instead of four)
    constexpr int xo = 0;
    const int x0a = x - 21;
    if (x >= 21 && x + 21 < L.w && y >= 21 && y + 21 < L.h && x0a + RP <= pitch) {
        const uint8_t *src = img + (long long)(y - 21) * pitch + x0a;
        // nine direct loads (global_load_lds_dword: any byte alignment, no VGPR round trip, no ds_write), all in flight together.
        // Lane = (row lane/11, dword lane%11) of a 5-row band (55 lanes), band k covers rows 5k..5k+4 and lands at raw + 220 k + 4 lane:
        // row-major with the 44-byte pitch.
        // scalar band base + one 32-bit lane offset: the bands advance on the scalar unit (a 64-bit vector multiply-add per load otherwise)
        const int lr = lane / 11, lc = lane - lr * 11;
        const unsigned voff = (unsigned)(lr * pitch + 4 * lc);
        if (lane < 55) {
#pragma unroll
            for (int k = 0; k < 8; k++, src += 5 * (long long)pitch)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(src + voff), reinterpret_cast<uint32_t *>(raw + 5 * RP * k), 4, 0, 0);
            if (lr < 3)   // rows 40..42
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(src + voff), reinterpret_cast<uint32_t *>(raw + 5 * RP * 8), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else { // image edge (BORDER_REFLECT_101 of the cloned level, :1312-1314)
        // lane = patch column (its reflected source column computed once), rows walked on the scalar unit, eight byte loads in
        // flight: ~2 vector instructions per row (an element-wise walk with a division and two reflections per byte cost more
        // than the whole rest of the keypoint, for the ~6 % of the keypoints that lie within 21 px of an image edge)
        const int cx = reflect101(x - 21 + min(lane, 42), L.w);
        if (lane < 43) {
#pragma unroll 8
            for (int r = 0; r < 43; r++) {
                const int ry = reflect101(y - 21 + r, L.h);
                raw[r * RP + lane] = img[(long long)ry * pitch + cx];
            }
        }
    }
    DSTAMP(6); // patch loads issued and consumed (the last LDS stores may still be in flight)
    __syncthreads();
    DSTAMP(0);
    // ---- IC_Angle: lane = (row v+15, half); integer moments, order-independent
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        // lane = (row v, half): its 16-pixel window (left half u = -16..-1, right half u = 0..15) is five aligned LDS
        // dwords realigned with v_alignbyte and masked to the circular patch (mask fetched with the first round trip);
        // sum(I) by v_sad_u8 against 0 and sum(k*I), k = 0..15, by v_dot4_u32_u8 against constant weights
        const int v = (lane >> 1) - 15, half = lane & 1;
        const int off = xo + (half ? 21 : 5);     // byte offset of the window in the staged row
        const uint32_t *d = reinterpret_cast<const uint32_t *>(raw + (21 + v) * RP) + (off >> 2);
        const unsigned D0 = d[0], D1 = d[1], D2 = d[2], D3 = d[3], D4 = d[4];
        const int sh = off & 3;
        const unsigned W0 = __builtin_amdgcn_alignbyte(D1, D0, sh) & omask.x, W1 = __builtin_amdgcn_alignbyte(D2, D1, sh) & omask.y,
                       W2 = __builtin_amdgcn_alignbyte(D3, D2, sh) & omask.z, W3 = __builtin_amdgcn_alignbyte(D4, D3, sh) & omask.w;
        const unsigned rs = __builtin_amdgcn_sad_u8(W0, 0u, __builtin_amdgcn_sad_u8(W1, 0u, __builtin_amdgcn_sad_u8(W2, 0u, __builtin_amdgcn_sad_u8(W3, 0u, 0u))));
        const unsigned pk = __builtin_amdgcn_udot4(W0, 0x03020100u, __builtin_amdgcn_udot4(W1, 0x07060504u,
                            __builtin_amdgcn_udot4(W2, 0x0B0A0908u, __builtin_amdgcn_udot4(W3, 0x0F0E0D0Cu, 0u, false), false), false), false);
        m10 = (int)pk - (half ? 0 : 16 * (int)rs);   // u = k - 16 in the left half
        m01 = v * (int)rs;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = dev_fast_atan2((float)m01, (float)m10);
    DSTAMP(1);
    // the steering sine / cosine (a long dependent fp64 chain) is computed here, where it can overlap the LDS traffic of the blur
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    dev_sincos(angle * factor_pi, &sn, &cs);
    // ---- row pass: 4 outputs per item from 3 aligned dwords.  Output k needs bytes k .. k + 6: instead of shifting the data
    // (v_alignbyte) the TAPS are shifted -- ten constant tap words, v_dot4_u32_u8 against each dword an output touches
    const unsigned g0 = da.gauss & 0xFFu, g1 = (da.gauss >> 8) & 0xFFu, g2 = (da.gauss >> 16) & 0xFFu, g3 = da.gauss >> 24;   // symmetric: g4 = g2, g5 = g1, g6 = g0
    const unsigned TA0 = g0 | g1 << 8 | g2 << 16 | g3 << 24, TB0 = g2 | g1 << 8 | g0 << 16;
    const unsigned TA1 = g0 << 8 | g1 << 16 | g2 << 24, TB1 = g3 | g2 << 8 | g1 << 16 | g0 << 24;
    const unsigned TA2 = g0 << 16 | g1 << 24, TB2 = g2 | g3 << 8 | g2 << 16 | g1 << 24, TC2 = g0;
    const unsigned TA3 = g0 << 24, TB3 = g1 | g2 << 8 | g3 << 16 | g2 << 24, TC3 = g1 | g0 << 8;
    // lane = (row r_lo = lane / 10 of a band of six rows, group gq = lane % 10), eight bands: every LDS address of the pass is
    // the lane's base plus an immediate (no per-item index arithmetic); the last band holds row 42 only
    if (lane < 60) {
        const int r_lo = lane / 10, gq = lane - r_lo * 10;
        const uint32_t *d0 = reinterpret_cast<const uint32_t *>(raw + r_lo * RP) + gq;
        uint16_t *w0 = hb + (4 * gq) * HR + r_lo;   // column-major: the column pass reads vertically adjacent values as packed pairs
#pragma unroll
        for (int it = 0; it < 8; it++) {
            if (it < 7 || r_lo == 0) {
                const uint32_t *d = d0 + it * 6 * (RP / 4);
                const unsigned W0 = d[0], W1 = d[1], W2 = d[2]; // the 10 bytes an item needs (4 outputs + 6 taps) start dword-aligned
                unsigned o[4];
                o[0] = __builtin_amdgcn_udot4(W0, TA0, __builtin_amdgcn_udot4(W1, TB0, 0u, false), false);
                o[1] = __builtin_amdgcn_udot4(W0, TA1, __builtin_amdgcn_udot4(W1, TB1, 0u, false), false);
                o[2] = __builtin_amdgcn_udot4(W0, TA2, __builtin_amdgcn_udot4(W1, TB2, __builtin_amdgcn_udot4(W2, TC2, 0u, false), false), false);
                o[3] = __builtin_amdgcn_udot4(W0, TA3, __builtin_amdgcn_udot4(W1, TB3, __builtin_amdgcn_udot4(W2, TC3, 0u, false), false), false);
                w0[6 * it] = (uint16_t)o[0];
                if (gq < 9) {   // the tenth group only owns column 36
#pragma unroll
                    for (int k = 1; k < 4; k++) w0[k * HR + 6 * it] = (uint16_t)o[k];
                }
            }
        }
    }
    __syncthreads();
    DSTAMP(2);
    // ---- column pass ONLY at the 512 sampled positions (8 per lane) instead of on all 37 x 37: the seven row-pass values of
    // a sample are contiguous in its column (column-major hb), fetched as four aligned dwords and realigned by the row parity
    // with one v_alignbit each (shift in a register; one misaligned ds_read_b128 instead returns the right bytes on gfx950 but ran
    // the kernel 36 % slower); an output is four v_dot2_u32_u16 against the packed symmetric taps (g0,g1)(g2,g3)(g2,g1)(g0,0)
    // with the rounding constant as the first accumulator.  Row 43 is padding: it only ever meets the zero tap.
    typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
    const u16x2v G01 = __builtin_bit_cast(u16x2v, g0 | (g1 << 16)), G23 = __builtin_bit_cast(u16x2v, g2 | (g3 << 16)),
                 G21 = __builtin_bit_cast(u16x2v, g2 | (g1 << 16)), G0 = __builtin_bit_cast(u16x2v, g0);
    // Rounding: cvRound(v) = round-half-even = the low bits of v + 1.5 * 2^23 (|v| < 2^22; one packed add for both coordinates,
    // no v_rndne / v_cvt).  With rb = bits(row + M), qb = bits(col + M): 44 * (low 24 bits of qb) + rb is the u16 index of
    // (18 + col, 18 + row) in hb plus a constant.
    const float MAGIC = 12582912.f;   // 0x4B400000
    auto blurred = [&](unsigned rb, unsigned qb) -> unsigned {
        const unsigned i16 = __umul24(qb, (unsigned)HR) + rb - (0x400000u * HR + 0x4B400000u) + 18u * (HR + 1);
        const unsigned ba = i16 << 1, sh = ba << 3;   // v_alignbit / v_lshrrev use the low 5 bits of the shift: 16 * (row parity)
        const uint32_t *d = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(hb) + (ba & ~3u));
        const unsigned D0 = d[0], D1 = d[1], D2 = d[2], D3 = d[3];
        unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, __builtin_amdgcn_alignbit(D1, D0, sh)), G01, 1u << 15, false); // sums stay below 2^25
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, __builtin_amdgcn_alignbit(D2, D1, sh)), G23, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, __builtin_amdgcn_alignbit(D3, D2, sh)), G21, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, D3 >> (sh & 31u)), G0, acc, false);
        const unsigned v = acc >> 16;
        return v > 255u ? 255u : v;
    };
    const float a = cs, bb = sn;
    unsigned long long words[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const uint32_t pw = pat4[jj];
        const float x0 = (float)(signed char)(pw & 0xFF), y0 = (float)(signed char)((pw >> 8) & 0xFF),
                    x1 = (float)(signed char)((pw >> 16) & 0xFF), y1 = (float)(signed char)(pw >> 24);
        // (x*b + y*a, x*a - y*b) as two packed fp32 multiplies and one packed add (v_pk_mul_f32 / v_pk_add_f32 round each
        // component like the scalar forms; y*(-b) == -(y*b) exactly, so the subtraction is unchanged)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 BA = { bb, a }, AnB = { a, -bb }, MM = { MAGIC, MAGIC };
        const f32x2 R0 = (f32x2{ x0, x0 } * BA + f32x2{ y0, y0 } * AnB) + MM, R1 = (f32x2{ x1, x1 } * BA + f32x2{ y1, y1 } * AnB) + MM;
        const unsigned t0 = blurred(__float_as_uint(R0.x), __float_as_uint(R0.y)), t1 = blurred(__float_as_uint(R1.x), __float_as_uint(R1.y));
        words[jj] = __ballot(t0 < t1);
    }
    if (lane == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(out_desc + ((long long)b * cap + idx) * 32);
        d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
        orbx_keypoint kp;
        kp.x = (float)x; kp.y = (float)y;
        if (l != 0) { kp.x *= L.scale; kp.y *= L.scale; } // :1326-1334
        kp.size = (float)L.patch_size;
        kp.angle = angle;
        kp.response = (float)resp;
        kp.octave = l;
        kp.class_id = -1;
        out_kps[(long long)b * cap + idx] = kp;
    }
    DSTAMP(4);
    SPAN_END(1);
#ifdef ORBX_DIAG
    if (lane == 0) atomicAdd(&g_desc_stamp[((blockIdx.x * 131 + blockIdx.y) & 4095) * 8 + 7], 1ull);
#endif
}

// ================================================================ host side

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// cv::resize coefficient tables, reference call site src/ORBextractor.cc:1366
// Stored interleaved, one (ofs, c0, c1, 0) quad of int16 per destination index: one 8-byte load per index on the device.
static void linear_tables(int ssize, int dsize, int16_t *quads)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floor((double)f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        int v0 = orbx_cv_round((1.f - f) * 2048), v1 = orbx_cv_round(f * 2048);
        quads[4 * d] = (int16_t)s;
        quads[4 * d + 1] = (int16_t)(v0 < -32768 ? -32768 : v0 > 32767 ? 32767 : v0);
        quads[4 * d + 2] = (int16_t)(v1 < -32768 ? -32768 : v1 > 32767 ? 32767 : v1);
        quads[4 * d + 3] = 0;
    }
}

template <class T>
static int ensure(T **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return ORBX_OK;
    if (*p) { ORBX_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    ORBX_HIP(hipMalloc((void **)p, need ? need : 16));
    *cap = need;
    return ORBX_OK;
}

static const size_t kTreeLdsLimit = 150 * 1024;
static size_t tree_tab_bytes(const Geom &G) { return (size_t)G.max_node_cap * (4 + 4 + 8 + 8 + 16 + 16 + 4 * 5); }
static size_t tree_fixed_lds(const Geom &G) { return (size_t)((G.max_cells_level + 4) & ~3) * 4 + 64; }
// the node tables (76 B per leaf) go to LDS when they fit there together with the cell prefix array and at least 3072 points
static bool tree_tab_in_lds(const Geom &G) { return tree_tab_bytes(G) + tree_fixed_lds(G) + (size_t)3072 * 6 <= kTreeLdsLimit; }
static size_t tree_lds_bytes(const Geom &G, int lds_pts_cap)
{
    return (tree_tab_in_lds(G) ? tree_tab_bytes(G) : 0) + tree_fixed_lds(G) + (size_t)lds_pts_cap * 6;
}
// LDS point capacity of k_tree: about a level's typical candidate count, bounded so that several (level, image)
// workgroups fit one CU; levels with more candidates keep their points in the HBM scratch
static int lds_pts_cap(const Geom &G)
{
    int c = (G.lv[0].w * G.lv[0].h / 160 + 1023) & ~1023; // P_0/160: measured best at 512 images per launch (1241x376: 3072)
    c = c < 3072 ? 3072 : c > 12288 ? 12288 : c;
    while (c > 3072 && tree_lds_bytes(G, c) > kTreeLdsLimit) c -= 1024;
    return c;
}

// Register form of k_tree (points and labels in VGPRs, see the kernel): for image sizes whose levels normally hold at most
// ORBX_TREE_REG_PTS candidates (the same P_0/160 rule) and node tables that fit the LDS; bigger levels of such an image go to the
// HBM scratch.  The workgroup's LDS is then the node tables (which double as the gather's staging area) + the cell prefix array.
static bool tree_reg_mode(const Geom &G) { return tree_tab_in_lds(G) && lds_pts_cap(G) <= ORBX_TREE_REG_PTS; }
// (register form: the capacity of the overflow array -- points beyond the register capacity of a level; 6 bytes each)
static int tree_launch_pts_cap(const Geom &G) { return tree_reg_mode(G) ? ORBX_TREE_OVER_PTS : lds_pts_cap(G); }
static size_t tree_launch_lds(const Geom &G)
{
    if (!tree_reg_mode(G)) return tree_lds_bytes(G, lds_pts_cap(G));
    return std::max(tree_tab_bytes(G), (size_t)ORBX_TREE_REG_PTS_BIG * 4) + tree_fixed_lds(G) + (size_t)ORBX_TREE_OVER_PTS * 6;   // (the 1024-thread form stages 4096 points)
}


// Host tables of k_pyr_group: for every group, per tile column (row) of its last level and per step k = 0 (the source level) .. n
// the region [lo, hi) the workgroup holds and the part [lo, own_hi) it writes out.  Appends to `tabs`; returns false when a region does
// not fit the LDS (the geometry then keeps per-level launches).
static bool build_pyr_groups(orbx_extractor *e, const Geom &G, std::vector<int16_t> &tabs)
{
    e->n_pyr_groups = 0;
    int sizes[ORBX_MAX_LEVELS], nsizes = 0;
    {
        const char *env = getenv("ORBX_PYR_GROUPS");        // experiments: "2,5" = levels 1-2, then 3-7; "0" = per-level launches only
        const char *p = env && *env ? env : "2,5";
        while (*p && nsizes < ORBX_MAX_LEVELS) {
            const int v = atoi(p);
            if (v < 1) return false;
            sizes[nsizes++] = std::min(v, ORBX_PYR_GROUP_MAX);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        if (!nsizes) return false;
    }
    for (int l = 1; l < G.nlevels; l++) if (G.lv[l].resize_lds == 2) return false;     // exact 2x levels are area averages
    int first = 1, gi = 0;
    while (first < G.nlevels) {
        const int n = std::min(sizes[std::min(gi, nsizes - 1)], G.nlevels - first);
        orbx_extractor::PyrGroup &P = e->pyr_groups[gi];
        P.first = first; P.n = n;
        size_t lds[2] = { 0, 0 };
        std::vector<int> ext_axis[2];
        for (int axis = 0; axis < 2; axis++) {
            const int T = axis ? PG_TH : PG_TW;
            auto dim = [&](int lvl) { return axis ? G.lv[lvl].h : G.lv[lvl].w; };
            const int last = first + n - 1, tiles = (dim(last) + T - 1) / T, rec = axis ? PG_CY_REC : PG_CX_REC;
            std::vector<int> lo((size_t)(n + 1) * tiles), hi(lo.size()), own(lo.size());
            for (int t = 0; t < tiles; t++) { lo[(size_t)n * tiles + t] = t * T; hi[(size_t)n * tiles + t] = own[(size_t)n * tiles + t] = std::min(t * T + T, dim(last)); }
            for (int k = n - 1; k >= 0; k--) {
                const int lvl = first - 1 + k, D = dim(lvl);                    // the level of step k; step k + 1 reads it through its table
                const int16_t *tb = &tabs[axis ? G.lv[lvl + 1].tab_y : G.lv[lvl + 1].tab_x];
                std::vector<int> nlo(tiles), nhi(tiles);
                for (int t = 0; t < tiles; t++) {
                    nlo[t] = tb[4 * lo[(size_t)(k + 1) * tiles + t]];
                    nhi[t] = std::min(tb[4 * (hi[(size_t)(k + 1) * tiles + t] - 1)] + 1, D - 1) + 1;
                    if (nlo[t] < 0 || nhi[t] <= nlo[t] || (t && nlo[t] < nlo[t - 1])) return false;   // not a monotone down-scaling table
                }
                for (int t = 0; t < tiles; t++) {
                    size_t i = (size_t)k * tiles + t;
                    if (k == 0) { lo[i] = nlo[t] & ~3; hi[i] = nhi[t]; own[i] = nhi[t]; }
                    else {
                        lo[i] = t ? nlo[t] : 0;
                        own[i] = t + 1 < tiles ? nlo[t + 1] : D;
                        hi[i] = std::max(nhi[t], own[i]);
                    }
                }
            }
            // records + the LDS need of the even / odd steps
            const int off = (int)tabs.size();
            (axis ? P.tab_cy : P.tab_cx) = off;
            tabs.resize(tabs.size() + (size_t)tiles * rec * (n + 1), 0);
            std::vector<int> ext(n + 1, 0);                                    // largest extent of a step over the tiles (x: LDS pitch, y: rows)
            for (int t = 0; t < tiles; t++)
                for (int k = 0; k <= n; k++) {
                    const size_t i = (size_t)k * tiles + t;
                    int16_t *r = &tabs[off + ((size_t)t * (n + 1) + k) * rec];
                    if (hi[i] > 32767) return false;
                    r[0] = (int16_t)lo[i]; r[1] = (int16_t)hi[i]; r[2] = (int16_t)own[i];
                    int extent = hi[i] - lo[i];
                    if (!axis) {
                        int div;
                        if (k == 0) {
                            const int pitch_lim = lo[i] + (((dim(first - 1) - lo[i]) + 3) & ~3);      // align4(w) as seen from lo: never past the pitch
                            const int ndw = (std::min(lo[i] + ((hi[i] - lo[i] + 3) & ~3), pitch_lim) - lo[i]) / 4;
                            r[3] = (int16_t)ndw; div = ndw; extent = 4 * ndw;
                        } else { div = (extent + 3) >> 2; extent = (extent + 3) & ~3; }     // groups of four pixels per row
                        const unsigned magic = (unsigned)((0x100000000ull + (unsigned)div - 1) / (unsigned)div);
                        r[4] = (int16_t)(magic & 0xFFFF); r[5] = (int16_t)(magic >> 16);
                    }
                    ext[k] = std::max(ext[k], extent);
                }
            if (!axis) P.tiles_x = tiles; else P.tiles_y = tiles;
            ext_axis[axis] = ext;
        }
        for (int k = 0; k < n; k++) lds[k & 1] = std::max(lds[k & 1], (size_t)ext_axis[0][k] * ext_axis[1][k]);   // pitch x rows; step n is not kept
        P.lds_b = (int)align_up(lds[0], 16);
        P.lds_bytes = P.lds_b + (int)align_up(lds[1], 16);
        if (P.lds_bytes > PG_LDS_LIMIT) return false;
        first += n; gi++;
    }
    e->n_pyr_groups = gi;
    return true;
}

int orbx_prepare_geometry(orbx_extractor *e, int w, int h)
{
    if (e->geom.w == w && e->geom.h == h) return ORBX_OK;
    if (w > e->max_w || h > e->max_h || w > ORBX_MAX_DIM || h > ORBX_MAX_DIM) {
        orbx_set_error("image %dx%d exceeds the extractor's maximum %dx%d", w, h, e->max_w, e->max_h);
        return ORBX_E_INVALID;
    }
    Geom G;
    memset(&G, 0, sizeof G);
    G.nlevels = e->nlevels; G.w = w; G.h = h;
    size_t tab_units = 0;
    for (int l = 0; l < e->nlevels; l++) {
        LevelGeom &L = G.lv[l];
        L.w = orbx_cv_round((float)w * e->isf[l]);   // src/ORBextractor.cc:1353
        L.h = orbx_cv_round((float)h * e->isf[l]);
        const int min_b = ORBX_MIN_BORDER, max_bx = L.w - ORBX_EDGE + 3, max_by = L.h - ORBX_EDGE + 3;
        const float width = (float)(max_bx - min_b), height = (float)(max_by - min_b);
        if (max_bx - min_b < 30 || max_by - min_b < 30) {
            orbx_set_error("level %d (%dx%d) is smaller than one 30-px FAST cell", l, L.w, L.h);
            return ORBX_E_TOO_SMALL;
        }
        L.n_cols = (int)(width / 30.f);               // :946-951
        L.n_rows = (int)(height / 30.f);
        L.w_cell = (int)ceilf(width / L.n_cols);
        L.h_cell = (int)ceilf(height / L.n_rows);
        if (L.w_cell > 59 || L.h_cell > 59) { orbx_set_error("internal: cell larger than 59"); return ORBX_E_INVALID; }
        L.n_cells = L.n_cols * L.n_rows;
        L.cell_base = G.total_cells;
        G.total_cells += L.n_cells;
        L.cand_cap = ((L.w_cell + 1) / 2) * ((L.h_cell + 1) / 2); // strict 3x3 maxima cannot be adjacent
        L.cand_off = G.cand_total;
        G.cand_total += (long long)L.n_cells * L.cand_cap;
        L.quota = e->quota[l];
        L.tree_w = max_bx - min_b; L.tree_h = max_by - min_b;
        L.n_ini = (int)roundf((float)L.tree_w / L.tree_h); // :627
        if (L.n_ini < 1) { orbx_set_error("level %d: aspect ratio gives zero quadtree roots (reference divides by zero)", l); return ORBX_E_TOO_SMALL; }
        L.hx = (float)L.tree_w / L.n_ini;                  // :628
        int nc = L.quota + 3 > 4 * L.n_ini ? L.quota + 3 : 4 * L.n_ini;
        L.node_cap = (nc + 4 + 3) & ~3;
        if (L.node_cap >= (1 << ORBX_NODE_BITS)) {
            orbx_set_error("level %d asks for %d features: the quadtree labels hold %d leaves per level (nfeatures <= ~%d at scale factor 1.2)",
                           l, L.quota, (1 << ORBX_NODE_BITS) - 8, 70000);
            return ORBX_E_INVALID;
        }
        L.kp_cap = L.node_cap; L.kp_off = G.kp_total; G.kp_total += L.kp_cap;
        L.scale = e->sf[l];
        L.patch_size = (int)(31 * e->sf[l]);               // :1023
        if (l >= 1) {
            L.pitch = (int)align_up(L.w, 64);
            L.pyr_off = G.pyr_bytes;
            G.pyr_bytes += (long long)L.pitch * L.h;
            L.tab_x = (int)tab_units; tab_units += 4 * (size_t)L.w;   // int16 units, multiples of 4: 8-byte aligned quads
            L.tab_y = (int)tab_units; tab_units += 4 * (size_t)L.h;
            // k_resize's per-tile records: one per tile row (first source row, count, emit entries), one per tile column
            L.tab_ty = (int)tab_units; tab_units += (size_t)RS_TY_REC * ((L.h + RS_TH - 1) / RS_TH);
            L.tab_tx = (int)tab_units; tab_units += 4 * (size_t)((L.w + RS_TW - 1) / RS_TW);
        }
        if (L.n_cells > G.max_cells_level) G.max_cells_level = L.n_cells;
        if (L.node_cap > G.max_node_cap) G.max_node_cap = L.node_cap;
    }
    G.pyr_bytes = (long long)align_up((size_t)G.pyr_bytes, 256);
    {   // LDS carve of k_fast, sized by the largest cell over the levels
        int max_th = 0, max_dh = 0, max_npx = 0;
        for (int l = 0; l < e->nlevels; l++) {
            const LevelGeom &L = G.lv[l];
            if (L.h_cell + 6 > max_th) max_th = L.h_cell + 6;
            if (L.h_cell > max_dh) max_dh = L.h_cell;
            if (L.w_cell * L.h_cell > max_npx) max_npx = L.w_cell * L.h_cell;
        }
        int max_w_cell = 0;
        for (int l = 0; l < e->nlevels; l++) if (G.lv[l].w_cell > max_w_cell) max_w_cell = G.lv[l].w_cell;
        // tile row = 1 + w_cell + 6 pixels rounded up to dwords <= 48 bytes; score row = w_cell + 2 <= 40
        G.fast_small = (((max_w_cell + 7 + 3) & ~3) <= 48 && max_w_cell + 2 <= 40) ? 1 : 0;
        const int tp = G.fast_small ? 48 : ORBX_TILE_PITCH, sp = G.fast_small ? 40 : ORBX_SCORE_PITCH;
        // tile rows: the cell, the whole rows of the last direct load, and the row overrun of the last pretest iteration
        // (up to 7 rows of at least 8 groups) plus its S neighbour three rows further down
        // (row dh - 1 + 8 of the pretest reads its S neighbour at tile row th + 7)
        // The overrun rows are only ever READ (their flags land in bitmap rows nobody looks at), so they need no storage of their
        // own: they alias whatever follows the tile (score tile and list, always more than 8 rows' worth).
        G.fast_lds_sc = (int)align_up((size_t)max_th * tp + 8, 16);
        G.fast_lds_list = G.fast_lds_sc + (int)align_up((size_t)(max_dh + 2) * sp, 16);
        G.fast_lds_bm = G.fast_lds_list + (int)align_up((size_t)std::min(max_npx, ORBX_FAST_LIST_CAP) * 2 + 16, 16);
        G.fast_bm_rows = (max_dh + 9 + 1) & ~1;          // even: the two bitmaps are zeroed as one run of 16-byte stores
        G.fast_lds_bytes = G.fast_lds_bm + 2 * G.fast_bm_rows * 8;
        // several waves per cell (small launches: LDS is no limit there): the list holds every pixel of the largest cell -- one round always
        G.fast_list_cap_big = std::max(max_npx, ORBX_FAST_LIST_CAP);
        G.fast_lds_bm_big = G.fast_lds_list + (int)align_up((size_t)G.fast_list_cap_big * 2 + 16, 16);
        G.fast_lds_bytes_big = G.fast_lds_bm_big + 2 * G.fast_bm_rows * 8;
    }
    // resize tables
    std::vector<int16_t> tabs(tab_units);
    std::vector<char> emit_ok(e->nlevels, 1);
    std::vector<std::vector<int16_t>> emit(e->nlevels);
    for (int l = 1; l < e->nlevels; l++) {
        LevelGeom &L = G.lv[l];
        const LevelGeom &S = G.lv[l - 1];
        linear_tables(S.w, L.w, &tabs[L.tab_x]);
        linear_tables(S.h, L.h, &tabs[L.tab_y]);
        // emit table of k_resize: source row sy -> the output row y whose LOWER source row min(sy0 + 1, S.h - 1) is sy:
        // (y | same << 13 | two << 14, b0, b1, 0), y = -1 for none
        emit[l].assign(4 * (size_t)S.h, 0);
        int16_t *ts = emit[l].data();
        const int16_t *ty = &tabs[L.tab_y];
        for (int sy = 0; sy < S.h; sy++) ts[4 * sy] = -1;
        for (int y = 0; y < L.h; y++) {
            const int sy0 = ty[4 * y], rb = sy0 + 1 < S.h - 1 ? sy0 + 1 : S.h - 1, same = sy0 == rb;
            if (ts[4 * rb] < 0) {
                ts[4 * rb] = (int16_t)(y | (same ? 0x2000 : 0)); ts[4 * rb + 1] = ty[4 * y + 1]; ts[4 * rb + 2] = ty[4 * y + 2];
            } else if (same && (ts[4 * rb] & 0xFFF) == y - 1 && !(ts[4 * rb] & 0x6000) && ty[4 * y + 1] == 2048 && ty[4 * y + 2] == 0) {
                ts[4 * rb] |= 0x4000;       // second row on the clamped last source row
            } else emit_ok[l] = 0;          // not a down-scaling table: such a level takes k_resize_direct
        }
    }
    for (int l = 1; l < e->nlevels; l++) { // does every output tile's source rectangle fit k_resize's LDS tile?
        LevelGeom &L = G.lv[l];
        const LevelGeom &S = G.lv[l - 1];
        const int16_t *tx = &tabs[L.tab_x], *ty = &tabs[L.tab_y];
        bool ok = true;
        for (int x0 = 0; x0 < L.w && ok; x0 += RS_TW) {
            const int xl = (x0 + RS_TW < L.w ? x0 + RS_TW : L.w) - 1;
            const int smax = tx[4 * xl] + 1 < S.w - 1 ? tx[4 * xl] + 1 : S.w - 1;
            if (((smax - tx[4 * x0]) / 4 + 1) * 4 > RS_PITCH) ok = false;
        }
        for (int y0 = 0; y0 < L.h && ok; y0 += RS_TH) {
            const int yl = (y0 + RS_TH < L.h ? y0 + RS_TH : L.h) - 1;
            const int smax = ty[4 * yl] + 1 < S.h - 1 ? ty[4 * yl] + 1 : S.h - 1;
            if (smax - ty[4 * y0] + 1 > RS_ROWS) ok = false;
        }
        L.resize_lds = (S.w == 2 * L.w && S.h == 2 * L.h) ? 2 : (ok && emit_ok[l] && L.h < 4096) ? 1 : 0;   // 2: area-average kernel path (k_resize_direct)
        if (L.resize_lds != 1) continue;
        // per-tile records (see k_resize): everything a tile needs before its loads, in one fetch
        for (int by = 0, y0 = 0; y0 < L.h; by++, y0 += RS_TH) {
            int16_t *r = &tabs[L.tab_ty + (size_t)by * RS_TY_REC];
            const int yl = (y0 + RS_TH < L.h ? y0 + RS_TH : L.h) - 1;
            const int sy_min = ty[4 * y0], sy_max = ty[4 * yl] + 1 < S.h - 1 ? ty[4 * yl] + 1 : S.h - 1, nrows = sy_max - sy_min + 1;
            r[0] = (int16_t)sy_min; r[1] = (int16_t)nrows; r[2] = r[3] = 0;
            for (int k = 0; k < RS_ROWS; k++) {
                int16_t *q = r + 4 + 4 * k;
                q[0] = -1; q[1] = q[2] = q[3] = 0;
                if (k >= nrows) continue;
                const int16_t *t = &emit[l][4 * (size_t)(sy_min + k)];
                if (t[0] < 0) continue;
                const int y = t[0] & 0xFFF, in1 = y >= y0 && y <= yl, in2 = (t[0] & 0x4000) && y + 1 >= y0 && y + 1 <= yl;
                if (!in1 && !in2) continue;
                q[0] = (int16_t)(y | (t[0] & 0x2000) | (in1 ? 0 : 0x1000) | (in2 ? 0x4000 : 0)); q[1] = t[1]; q[2] = t[2];
            }
        }
        for (int bx = 0, x0 = 0; x0 < L.w; bx++, x0 += RS_TW) {
            int16_t *r = &tabs[L.tab_tx + 4 * (size_t)bx];
            const int xl = (x0 + RS_TW < L.w ? x0 + RS_TW : L.w) - 1;
            const int sx_min = tx[4 * x0], sx_max = tx[4 * xl] + 1 < S.w - 1 ? tx[4 * xl] + 1 : S.w - 1, need = sx_max - sx_min + 1;
            const int nfull = std::min((need + 15) >> 4, (S.w - sx_min) >> 4), tail = std::max(need - 16 * nfull, 0);
            r[0] = (int16_t)sx_min; r[1] = (int16_t)nfull; r[2] = (int16_t)tail; r[3] = 0;
        }
    }
    if (!build_pyr_groups(e, G, tabs)) e->n_pyr_groups = 0;
    std::vector<CellRec> cells(G.total_cells);
    for (int l = 0; l < e->nlevels; l++) {
        const LevelGeom &L = G.lv[l];
        const int max_bx = L.w - ORBX_MIN_BORDER, max_by = L.h - ORBX_MIN_BORDER;
        for (int ci = 0; ci < L.n_cells; ci++) {
            CellRec &c = cells[L.cell_base + ci];
            const int row = ci / L.n_cols, col = ci % L.n_cols;
            const int ini_y = ORBX_MIN_BORDER + row * L.h_cell, ini_x = ORBX_MIN_BORDER + col * L.w_cell; // :957-971
            const int max_y = ini_y + L.h_cell + 6 < max_by ? ini_y + L.h_cell + 6 : max_by;
            const int max_x = ini_x + L.w_cell + 6 < max_bx ? ini_x + L.w_cell + 6 : max_bx;
            c.level = (short)l;
            c.ini_x = (short)ini_x; c.ini_y = (short)ini_y; c.tw = (short)(max_x - ini_x); c.th = (short)(max_y - ini_y);
            // src/ORBextractor.cc:961-976 skip rules (note the asymmetric 3 / 6)
            c.skip = (ini_y >= max_by - 3 || ini_x >= max_bx - 6 || c.tw - 6 <= 0 || c.th - 6 <= 0) ? 1 : 0;
            c.pitch = L.pitch; c.cand_cap = L.cand_cap; c.pyr_off = L.pyr_off;
            { const int gpr = (c.tw - 6 + 3) >> 2; c.gpr_magic = gpr > 0 ? 0xFFFFFFFFu / (unsigned)gpr + 1u : 0u; }
            c.cand_slot = L.cand_off + (long long)ci * L.cand_cap;
        }
    }
    // k_fast2: pairs of horizontally adjacent cells (batches).  Usable when every cell's detect area fits 32 bits x 64 rows.
    std::vector<PairRec> pairs;
    {
        bool ok = true;
        for (int l = 0; l < e->nlevels; l++) if (G.lv[l].w_cell > 32 || G.lv[l].h_cell > 56) ok = false;
        G.fast2_ok = ok ? 1 : 0;
        G.total_pairs = 0;
        if (ok) {
            // two groups of levels, each a launch with its own LDS carve: detect areas of at most 32 rows (the 30-px grid's usual cells), then the
            // taller ones (a level whose cell rows do not divide evenly: up to 40 rows at 1241x376) -- sized together, the tall tiles cost every
            // wave of the launch a sixth of its occupancy
            for (int grp = 0; grp < 2; grp++) {
                int max_th = 0, max_dh = 0;
                G.fast2_first[grp] = (int)pairs.size();
                for (int l = 0; l < e->nlevels; l++) {
                    const LevelGeom &L = G.lv[l];
                    if ((L.h_cell <= 32 ? 0 : 1) != grp) continue;
                    max_th = std::max(max_th, L.h_cell + 6); max_dh = std::max(max_dh, L.h_cell);
                    for (int row = 0; row < L.n_rows; row++)
                        for (int col = 0; col < L.n_cols; col += 2) {
                            const CellRec &a = cells[L.cell_base + row * L.n_cols + col];
                            const bool has_b = col + 1 < L.n_cols;
                            PairRec q;
                            memset(&q, 0, sizeof q);
                            q.level = (short)l; q.ncells = has_b ? 2 : 1;
                            q.ini_x = a.ini_x; q.ini_y = a.ini_y; q.th = a.th;
                            q.dwa = a.skip ? 0 : (short)(a.tw - 6);
                            q.dwb = 0;
                            if (has_b && !a.skip) {
                                const CellRec &bc = cells[L.cell_base + row * L.n_cols + col + 1];
                                if (!bc.skip) {
                                    q.dwb = (short)(bc.tw - 6);
                                    // B starts one cell width right of A, on the same rows: the pair's tile is one rectangle
                                    if (bc.ini_x != a.ini_x + L.w_cell || bc.ini_y != a.ini_y || bc.th != a.th || a.tw != L.w_cell + 6) ok = false;
                                }
                            }
                            if (q.dwa > 32 || q.dwb > 32 || q.dwa < 0 || q.dwb < 0 || (q.dwb && q.dwa != L.w_cell)) ok = false;
                            const int off = q.dwb ? 32 - q.dwa : 0;
                            if (a.ini_x - (1 + off) < 0) ok = false;                   // the fetch starts 1 + off bytes left of cell A
                            const int nbits = off + q.dwa + q.dwb, gpr = (nbits + 3) >> 2;
                            q.gpr_magic = gpr > 0 ? 0xFFFFFFFFu / (unsigned)gpr + 1u : 0u;
                            q.cell = L.cell_base + row * L.n_cols + col;
                            q.pitch = L.pitch; q.cand_cap = L.cand_cap; q.pyr_off = L.pyr_off; q.cand_slot = a.cand_slot;
                            pairs.push_back(q);
                        }
                }
                G.fast2_count[grp] = (int)pairs.size() - G.fast2_first[grp];
                // LDS carve (pitches FAST2_P / FAST2_SP): tile | score tile | candidate list | candidate + survivor bitmaps; the pretest's
                // overrun rows alias what follows the tile (see k_fast)
                G.fast2_lds_sc[grp] = (int)align_up((size_t)max_th * FAST2_P + 8, 16);
                G.fast2_lds_list[grp] = G.fast2_lds_sc[grp] + (int)align_up((size_t)(max_dh + 2) * FAST2_SP, 16);
                G.fast2_lds_bm[grp] = G.fast2_lds_list[grp] + (int)align_up((size_t)FAST2_LIST_CAP * 2 + 16, 16);
                G.fast2_bm_rows[grp] = (max_dh + 9 + 1) & ~1;
                G.fast2_lds_bytes[grp] = G.fast2_lds_bm[grp] + 2 * G.fast2_bm_rows[grp] * 8;
            }
            if (!ok) { G.fast2_ok = 0; pairs.clear(); }
            G.total_pairs = (int)pairs.size();
        }
    }
    if (tree_launch_lds(G) > kTreeLdsLimit) {
        orbx_set_error("internal: %d FAST cells per level do not fit the quadtree kernel's LDS", G.max_cells_level);
        return ORBX_E_INVALID;
    }
    ORBX_HIP(orbx_use_device(e->device));
    {   // earlier launches (possibly on a caller's non-blocking stream) still read d_geom / d_tabs / d_cells and the workspaces
        const int qrc = orbx_quiesce(e);
        if (qrc) return qrc;
    }
    const size_t B = e->max_batch;
    int rc;
    if ((rc = ensure(&e->d_tabs, &e->tabs_cap, tabs.size() * 2))) return rc;
    if ((rc = ensure(&e->d_cells, &e->cells_cap, cells.size() * sizeof(CellRec)))) return rc;
    if (!pairs.empty() && (rc = ensure(&e->d_pairs, &e->pairs_cap, pairs.size() * sizeof(PairRec)))) return rc;
    if ((rc = ensure(&e->d_pyr, &e->pyr_cap, (size_t)G.pyr_bytes * B))) return rc;
    if ((rc = ensure(&e->d_cell_cnt, &e->cell_cnt_cap, (size_t)G.total_cells * B * 4))) return rc;
    if ((rc = ensure(&e->d_cand, &e->cand_cap, (size_t)G.cand_total * B * 4))) return rc;
    if ((rc = ensure(&e->d_cand_prim, &e->cand_prim_cap, (size_t)G.total_cells * B * ORBX_CAND_PRIM * 4))) return rc;
    {
        size_t need = (size_t)G.cand_total * B;
        if (need > e->tree_cap || !e->d_tree_pts) {
            if (e->d_tree_pts) ORBX_HIP(hipFree(e->d_tree_pts));
            if (e->d_tree_nid) ORBX_HIP(hipFree(e->d_tree_nid));
            e->d_tree_pts = nullptr; e->d_tree_nid = nullptr;
            ORBX_HIP(hipMalloc((void **)&e->d_tree_pts, need * 4));
            ORBX_HIP(hipMalloc((void **)&e->d_tree_nid, need * 2));
            e->tree_cap = need;
        }
    }
    if ((rc = ensure(&e->d_lvl_kp, &e->lvl_kp_cap, (size_t)G.kp_total * B * 4))) return rc;
    if (!tree_tab_in_lds(G) && (rc = ensure(&e->d_tree_tab, &e->tree_tab_cap, align_up(tree_tab_bytes(G), 256) * e->nlevels * B))) return rc;
    e->rt_kps = nullptr;
    if (G.lv[0].h <= ORBX_ROWTAB_MAX_ROWS && !getenv("ORBX_NO_ROWTAB")) {
        // row table by-product of k_desc: one 16-byte entry per keypoint (orbx_stereo.hip)
        e->rt_ent_cap = G.kp_total;
        if ((rc = ensure(&e->d_rt_off, &e->rt_off_cap, (size_t)(G.lv[0].h + 1) * B * sizeof(int)))) return rc;
        if ((rc = ensure(&e->d_rt_entries, &e->rt_entries_cap, (size_t)e->rt_ent_cap * B * 16))) return rc;
    } else if (e->d_rt_off) {
        ORBX_HIP(hipFree(e->d_rt_off)); e->d_rt_off = nullptr; e->rt_off_cap = 0;
    }
    ORBX_HIP(hipMemcpy(e->d_tabs, tabs.data(), tabs.size() * 2, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(e->d_cells, cells.data(), cells.size() * sizeof(CellRec), hipMemcpyHostToDevice));
    if (!pairs.empty()) ORBX_HIP(hipMemcpy(e->d_pairs, pairs.data(), pairs.size() * sizeof(PairRec), hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(e->d_geom, &G, sizeof G, hipMemcpyHostToDevice));
    {
        const void *kt[4] = { reinterpret_cast<const void *>(k_tree<256, true>), reinterpret_cast<const void *>(k_tree<256, false>),
                              reinterpret_cast<const void *>(k_tree<1024, true>), reinterpret_cast<const void *>(k_tree<1024, false>) };
        for (const void *f : kt) ORBX_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_launch_lds(G)));
    }
    e->geom = G;
    return ORBX_OK;
}

static int upload_constants(orbx_extractor *e)
{
    uint32_t pat[256];
    for (int i = 0; i < 256; i++)
        pat[i] = (uint32_t)(uint8_t)ORB_PAT_X0[i] | ((uint32_t)(uint8_t)ORB_PAT_Y0[i] << 8) | ((uint32_t)(uint8_t)ORB_PAT_X1[i] << 16) |
                 ((uint32_t)(uint8_t)ORB_PAT_Y1[i] << 24);
    ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_pat4), pat, sizeof pat));
    {   // k_desc's orientation lanes: lane = (row v = lane/2 - 15, half = lane & 1); the left half covers u = -16..-1 and
        // keeps u >= -umax[|v|], the right half covers u = 0..15 and keeps u <= umax[|v|] (src/ORBextractor.cc:91-108)
        uint8_t m[64][16];
        memset(m, 0, sizeof m);
        for (int lane = 0; lane < 62; lane++) {
            const int v = (lane >> 1) - 15, d = e->umax[v < 0 ? -v : v];
            for (int k = 0; k < 16; k++) {
                const int u = (lane & 1) ? k : k - 16;
                m[lane][k] = (u >= -d && u <= d) ? 0xFF : 0;
            }
        }
        ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_omask), m, sizeof m));
    }
    return ORBX_OK;
}

// The 7-tap sigma = 2 kernel of cv::GaussianBlur(.., Size(7, 7), 2, 2, BORDER_REFLECT_101) on 8-bit images (src/ORBextractor.cc:1311)
// as the 8-bit fixed-point integers the OpenCV generation named by `profile` filters with.  Both generations run the same
// arithmetic around the taps -- exact integer row pass, column pass (sum + 2^15) >> 16 -- so the table IS the profile:
//   ORBX_CV_PROFILE_3_2   (OpenCV <= 3.4.1): cvRound(k * 256) of the float kernel, not renormalised: 18 34 49 55 49 34 18 (sum 257)
//   ORBX_CV_PROFILE_3_4_2 (OpenCV >= 3.4.2 / 4.x, the bit-exact fixed-point path): rounded from the outside in with the
//                          rounding error carried along, centre = 256 - the rest: 18 34 48 56 48 34 18 (sum 256)
// (both restated from memory of OpenCV -- parity unpinned).
extern "C" int orbx_gaussian_taps(int profile, int taps[7])
{
    if (!taps || (profile != ORBX_CV_PROFILE_3_2 && profile != ORBX_CV_PROFILE_3_4_2)) { orbx_set_error("orbx_gaussian_taps: unknown profile %d", profile); return ORBX_E_INVALID; }
    const double scale2x = -0.5 / (2.0 * 2.0);
    if (profile == ORBX_CV_PROFILE_3_2) {
        float cf[7]; double sum = 0;
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; cf[i] = (float)exp(scale2x * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); taps[i] = (int)lrint((double)cf[i] * 256.0); }
    } else {
        double k[7], sum = 0, err = 0;
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; k[i] = exp(scale2x * x * x); sum += k[i]; }
        int rest = 0;
        for (int i = 0; i < 3; i++) {
            const double adj = k[i] / sum * 256.0 + err;
            const int v = (int)lrint(adj);
            err = adj - v;
            taps[i] = taps[6 - i] = v;
            rest += 2 * v;
        }
        taps[3] = 256 - rest;
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_set_cv_profile(orbx_extractor *e, int profile)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    int taps[7];
    const int rc = orbx_gaussian_taps(profile, taps);
    if (rc) return rc;
    for (int i = 0; i < 4; i++) e->gauss[i] = taps[i];   // launch constants of k_desc: later launches use them, earlier ones keep theirs
    e->cv_profile = profile;
    return ORBX_OK;
}

extern "C" int orbx_extractor_set_pyramid_group_limit(orbx_extractor *e, int max_images)
{
    if (!e || max_images < 0) { orbx_set_error("orbx_extractor_set_pyramid_group_limit: invalid argument"); return ORBX_E_INVALID; }
    e->pyr_group_max_images = max_images;       // a launch constant of later extractions; results do not depend on it
    e->pyr_group_mid_images = max_images == 0 ? 0 : e->pyr_group_mid_cfg;       // 0 = one launch per level, always; any other limit: the configured mid size again
    for (orbx_extractor *x : e->lanes) if (x) { x->pyr_group_max_images = max_images; x->pyr_group_mid_images = e->pyr_group_mid_images; }
    return ORBX_OK;
}

extern "C" int orbx_extractor_create(orbx_extractor **out, int nfeatures, float scale_factor, int nlevels,
                                     int ini_th, int min_th, int device, int max_w, int max_h, int max_batch)
{
    if (!out) { orbx_set_error("out is NULL"); return ORBX_E_INVALID; }
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !(scale_factor > 1.0f) || max_batch < 1 ||
        max_w < 1 || max_h < 1 || ini_th < 1 || min_th < 1 || ini_th > 255 || min_th > ini_th) {
        orbx_set_error("orbx_extractor_create: invalid parameter");
        return ORBX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        orbx_set_error("no usable HIP device (requested %d of %d); liborbx has no CPU fallback", device, ndev);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_extractor *e = new orbx_extractor();
    e->prof_mask = ~0u;
    orbx_extractor_set_cv_profile(e, ORBX_CV_PROFILE_3_2);   // the OpenCV the reference was tested with (README.md:68)
    e->device = device; e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th; e->min_th = min_th;
    e->scale_factor = scale_factor; e->max_w = max_w; e->max_h = max_h; e->max_batch = max_batch;
    {   // launches of up to this many images build the pyramid with k_pyr_group (2 launches instead of 7); more: k_resize per level
        const char *env = getenv("ORBX_PYR_GROUP_MAX_IMAGES");
        e->pyr_group_max_images = env && *env ? atoi(env) : 8;
        const char *mid = getenv("ORBX_PYR_GROUP_MID_IMAGES");
        e->pyr_group_mid_images = e->pyr_group_mid_cfg = mid && *mid ? atoi(mid) : 24;     // (8 frames: 57.2 -> 59.8 k frames/s; 16 frames: the same; 32: slower)
        const char *kpw = getenv("ORBX_STEREO_KPW");      // tests: k_stereo's one / four keypoints per wave on the same input
        e->stereo_kpw_forced = kpw && (*kpw == '1' || *kpw == '4') ? *kpw - '0' : 0;
        const char *pl = getenv("ORBX_PIPE_LANES"), *pi = getenv("ORBX_PIPE_INLINE");
        // defaults (examples/stereo_stream on one camera stream): four lanes, transport by copy kernel on the lane's stream: 19 k frames/s;
        // copy engines on two copy streams: 15 k whatever the lanes; ORBX_PIPE_INLINE=0 / ORBX_PIPE_KCOPY=0 select the older forms
        e->pipe_lanes = pl && *pl >= '1' && *pl <= '0' + ORBX_PIPE_DEPTH ? *pl - '0' : ORBX_PIPE_DEPTH;
        e->pipe_inline = !(pi && *pi == '0');
        const char *pk = getenv("ORBX_PIPE_KCOPY");
        e->pipe_kcopy = !(pk && *pk == '0');
        const char *fw = getenv("ORBX_FAST_WAVES");      // tests / experiments: force k_fast's waves per cell
        e->fast_waves = fw && *fw >= '1' && *fw <= '4' ? *fw - '0' : 0;
        // ORBX_FAST_PAIR=1: the pair kernel (k_fast2) where the geometry allows.  Off by default: bit-exact, 4 % fewer VALU instructions per cell and
        // less halo traffic, but 5-10 % SLOWER than one cell per wave on MI355X (the kernel sits at the knee between VALU issue
        // and latency, and a pair wave's 7.8 KB of LDS leaves 21 waves per CU against 28)
        const char *fp = getenv("ORBX_FAST_PAIR");
        e->fast_pair = fp && *fp == '1' ? 1 : 0;
    }
    // src/ORBextractor.cc:436-461
    e->sf[0] = 1.0f; e->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { e->sf[i] = (float)(e->sf[i - 1] * e->scale_factor); e->sig2[i] = e->sf[i] * e->sf[i]; }
    for (int i = 0; i < nlevels; i++) { e->isf[i] = 1.0f / e->sf[i]; e->isig2[i] = 1.0f / e->sig2[i]; }
    // :468-493
    float factor = (float)(1.0f / e->scale_factor);
    float n_desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { e->quota[l] = orbx_cv_round(n_desired); sum += e->quota[l]; n_desired *= factor; }
    e->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    // :510-533
    {
        int v, v0, vmax = (int)floor(15 * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(15 * sqrtf(2.f) / 2);
        const double hp2 = 15 * 15;
        for (v = 0; v <= vmax; ++v) e->umax[v] = (int)lrint(sqrt(hp2 - v * v));
        for (v = 15, v0 = 0; v >= vmin; --v) { while (e->umax[v0] == e->umax[v0 + 1]) ++v0; e->umax[v] = v0; ++v0; }
    }
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { orbx_set_error("hipStreamCreate failed: %s", hipGetErrorString(he)); delete e; return ORBX_E_HIP; }
    int rc = upload_constants(e);
    if (rc == ORBX_OK && hipMalloc((void **)&e->d_geom, sizeof(Geom)) != hipSuccess) { orbx_set_error("hipMalloc failed"); rc = ORBX_E_HIP; }
    if (rc == ORBX_OK && hipMalloc((void **)&e->d_lvl_cnt, sizeof(int) * (size_t)max_batch * ORBX_MAX_LEVELS + 16) != hipSuccess) { orbx_set_error("hipMalloc failed"); rc = ORBX_E_HIP; }
    if (rc == ORBX_OK && hipHostMalloc((void **)&e->h_flag, 64, hipHostMallocDefault) != hipSuccess) { orbx_set_error("hipHostMalloc failed"); rc = ORBX_E_HIP; }
    if (rc != ORBX_OK) { orbx_extractor_destroy(e); return rc; }
    *e->h_flag = 0;
    // last int of d_lvl_cnt is the kernel error flag
    hipMemset(e->d_lvl_cnt, 0, sizeof(int) * (size_t)max_batch * ORBX_MAX_LEVELS + 16);
    *out = e;
    return ORBX_OK;
}

extern "C" void orbx_debug_pipe_prof_print();
extern "C" void orbx_extractor_destroy(orbx_extractor *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    if (e->pipe_counted) orbx_debug_pipe_prof_print();
    for (orbx_extractor *&x : e->lanes) if (x) { orbx_extractor_destroy(x); x = nullptr; }
    if (e->pipe_counted) orbx_pipe_handle_released();
    if (e->stream) hipStreamSynchronize(e->stream);
    for (auto &ev : e->prof_ev) { if (ev.owns_a && ev.a) hipEventDestroy(ev.a); if (ev.b) hipEventDestroy(ev.b); }
    for (auto ev : e->prof_pool) hipEventDestroy(ev);
    if (e->ev_switch) hipEventDestroy(e->ev_switch);
    void *ptrs[] = { e->d_cand_prim, e->d_tree_tab, e->d_cells, e->d_pairs, e->d_geom, e->d_tabs, e->d_pyr, e->d_stage_in, e->d_cell_cnt, e->d_cand, e->d_tree_pts, e->d_tree_nid,
                     e->d_lvl_cnt, e->d_lvl_kp, e->d_out_kps, e->d_out_desc, e->d_out_n, e->d_out_ur, e->d_out_depth, e->d_st_dist, e->d_st_entries, e->d_rt_off, e->d_rt_entries, e->d_st_arrive };
    for (void *p : ptrs) if (p) hipFree(p);
    for (void *p : e->scratch) if (p) hipFree(p);
    for (PipeSlot &s : e->pipe) {
        if (s.ev_d2h) hipEventSynchronize(s.ev_d2h);
        void *dp[] = { s.d_in, s.d_out };
        for (void *p : dp) if (p) hipFree(p);
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        hipEvent_t evs[] = { s.ev_h2d, s.ev_done, s.ev_d2h };
        for (hipEvent_t ev : evs) if (ev) hipEventDestroy(ev);
    }
    if (e->copy_in) hipStreamDestroy(e->copy_in);
    if (e->copy_out) hipStreamDestroy(e->copy_out);
    if (e->h_stage_in) hipHostFree(e->h_stage_in);
    if (e->h_out) hipHostFree(e->h_out);
    if (e->h_flag) hipHostFree(e->h_flag);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int orbx_get_levels(const orbx_extractor *e) { return e ? e->nlevels : ORBX_E_INVALID; }
extern "C" float orbx_get_scale_factor(const orbx_extractor *e) { return e ? (float)e->scale_factor : 0.f; }
extern "C" int orbx_get_scale_tables(const orbx_extractor *e, float *s, float *is, float *g2, float *ig2)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    for (int i = 0; i < e->nlevels; i++) {
        if (s) s[i] = e->sf[i];
        if (is) is[i] = e->isf[i];
        if (g2) g2[i] = e->sig2[i];
        if (ig2) ig2[i] = e->isig2[i];
    }
    return ORBX_OK;
}
extern "C" int orbx_get_features_per_level(const orbx_extractor *e, int *q)
{
    if (!e || !q) { orbx_set_error("null argument"); return ORBX_E_INVALID; }
    for (int i = 0; i < e->nlevels; i++) q[i] = e->quota[i];
    return ORBX_OK;
}

extern "C" int orbx_max_keypoints(const orbx_extractor *e, int w, int h)
{
    if (!e || w < 1 || h < 1) { orbx_set_error("invalid argument"); return ORBX_E_INVALID; }
    int total = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const int lw = orbx_cv_round((float)w * e->isf[l]), lh = orbx_cv_round((float)h * e->isf[l]);
        const int tw = lw - 32, th = lh - 32;
        if (tw < 30 || th < 30) { orbx_set_error("image too small"); return ORBX_E_TOO_SMALL; }
        const int n_ini = (int)roundf((float)tw / th);
        const int nc = e->quota[l] + 3 > 4 * n_ini ? e->quota[l] + 3 : 4 * n_ini;
        total += (nc + 4 + 3) & ~3;
    }
    return total;
}

static_assert((ORBX_PIPE_DEPTH & (ORBX_PIPE_DEPTH - 1)) == 0, "tickets wrap at 2^31: the depth must divide it");

int orbx_use_stream(orbx_extractor *e, hipStream_t s)
{
    if (e->last_launch_stream && e->last_launch_stream != s) {
        if (!e->ev_switch) ORBX_HIP(hipEventCreateWithFlags(&e->ev_switch, hipEventDisableTiming));
        ORBX_HIP(hipEventRecord(e->ev_switch, e->last_launch_stream));
        ORBX_HIP(hipStreamWaitEvent(s, e->ev_switch, 0));
    }
    e->last_launch_stream = s;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_extractor *e, const void *d_imgs, size_t img_stride, size_t pitch,
                                         int batch, int w, int h, void *d_kps, void *d_desc, int cap, void *d_n_out,
                                         void *stream)
{
    if (!e || !d_imgs || !d_kps || !d_desc || !d_n_out || batch < 1 || batch > e->max_batch || w < 1 || h < 1 ||
        pitch < (size_t)w || (batch > 1 && img_stride < pitch * (size_t)h)) {
        orbx_set_error("orbx_extract_batch_device: invalid argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(orbx_use_device(e->device));
    int rc = orbx_prepare_geometry(e, w, h);
    if (rc) return rc;
    const Geom &G = e->geom;
    if (cap < G.kp_total) {
        orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d", cap, G.kp_total);
        return ORBX_E_CAPACITY;
    }
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    if ((rc = orbx_use_stream(e, s))) return rc;
    PyrRef pr;
    pr.img0 = (const uint8_t *)d_imgs; pr.img0_stride = (long long)img_stride; pr.img0_pitch = (int)pitch;
    pr.pyr = e->d_pyr; pr.pyr_stride = G.pyr_bytes;
    e->last_img0 = pr.img0; e->last_img_stride = img_stride; e->last_pitch = pitch; e->last_batch = batch;

    // Pyramid: launches of a frame or two build it in two grouped launches; mid-sized launches keep one launch per level for the big levels
    // (1, 2) and take the small ones (3 .. 7, latency-bound even for dozens of images) in one grouped launch; batches: one launch per level
    const int regime = e->n_pyr_groups > 0 && batch <= e->pyr_group_max_images ? 2
                     : e->n_pyr_groups > 1 && batch <= e->pyr_group_mid_images ? 1 : 0;
    auto launch_level = [&](int l) {
        const LevelGeom &L = G.lv[l];
        orbx_prof_begin(e, ORBX_STAGE_RESIZE, s);
        if (L.resize_lds == 1)
        {
            const LevelGeom &S = G.lv[l - 1];
            ResizeArgs ra;
            ra.d_w = L.w; ra.d_pitch = L.pitch; ra.s_w = S.w; ra.s_pitch = S.pitch; ra.s_level0 = l == 1;
            ra.tab_x = L.tab_x; ra.tab_tx = L.tab_tx; ra.tab_ty = L.tab_ty; ra.d_off = L.pyr_off; ra.s_off = S.pyr_off;
            hipLaunchKernelGGL(k_resize, dim3(batch, (L.w + RS_TW - 1) / RS_TW, (L.h + RS_TH - 1) / RS_TH), dim3(RS_NT), 0, s, ra, pr, e->d_pyr, e->d_tabs);
        }
        else
            hipLaunchKernelGGL(k_resize_direct, dim3((L.pitch / 4 + 63) / 64, (L.h + 3) / 4, batch), dim3(256), 0, s,
                               e->d_geom, l, pr, e->d_pyr, e->d_tabs);
        orbx_prof_end(e, s);
    };
    auto launch_group = [&](int gi) {
        const orbx_extractor::PyrGroup &P = e->pyr_groups[gi];
        const LevelGeom &S = G.lv[P.first - 1];
        PyrGroupArgs ga;
        memset(&ga, 0, sizeof ga);
        ga.n = P.n; ga.s_level0 = P.first == 1; ga.s_w = S.w; ga.s_h = S.h; ga.s_pitch = S.pitch; ga.s_off = S.pyr_off;
        ga.tab_cx = P.tab_cx; ga.tab_cy = P.tab_cy; ga.lds_b = P.lds_b;
        for (int k = 0; k < P.n; k++) {
            const LevelGeom &L = G.lv[P.first + k];
            ga.lv[k].w = L.w; ga.lv[k].h = L.h; ga.lv[k].pitch = L.pitch; ga.lv[k].tab_x = L.tab_x; ga.lv[k].tab_y = L.tab_y; ga.lv[k].pyr_off = L.pyr_off;
        }
        orbx_prof_begin(e, ORBX_STAGE_RESIZE, s);
        hipLaunchKernelGGL(k_pyr_group, dim3(P.tiles_x, P.tiles_y, batch), dim3(PG_NT), (size_t)P.lds_bytes, s, ga, pr, e->d_pyr, e->d_tabs);
        orbx_prof_end(e, s);
    };
    if (regime == 0) for (int l = 1; l < G.nlevels; l++) launch_level(l);
    else
        for (int gi = 0; gi < e->n_pyr_groups; gi++) {
            if (regime == 2 || gi >= 1) launch_group(gi);
            else for (int l = e->pyr_groups[gi].first; l < e->pyr_groups[gi].first + e->pyr_groups[gi].n; l++) launch_level(l);
        }
    orbx_prof_begin(e, ORBX_STAGE_FAST, s);
    e->last_fast_form = 1;
    {
        FastArgs fa;
        fa.total_cells = G.total_cells; fa.lds_sc = G.fast_lds_sc; fa.lds_list = G.fast_lds_list; fa.lds_bm = G.fast_lds_bm;
        fa.bm_rows = G.fast_bm_rows; fa.ini_th = e->ini_th; fa.min_th = e->min_th; fa.cand_total = G.cand_total; fa.list_cap = ORBX_FAST_LIST_CAP;
        const dim3 grid((G.total_cells + 8 * FAST_XG - 1) / (8 * FAST_XG) * (8 * FAST_XG), batch);
        if (G.fast_small)
        {
            // a frame or two: several waves per cell (the launch lasts as long as its fullest cell); batches: one
            const long long waves1 = (long long)G.total_cells * batch;
            const int nw = e->fast_waves ? e->fast_waves : waves1 * 4 <= 16384 ? 4 : waves1 * 2 <= 16384 ? 2 : 1;    // (tools/sweep_small.sh: one frame 4, two frames 2, more 1)
            int lds_bytes = G.fast_lds_bytes;
            if (nw > 1) { fa.list_cap = G.fast_list_cap_big; fa.lds_bm = G.fast_lds_bm_big; lds_bytes = G.fast_lds_bytes_big; }
            // one wave per PAIR of horizontally adjacent cells (k_fast2): opt-in experiment form (ORBX_FAST_PAIR=1)
            if (G.fast2_ok && e->fast_pair == 1) {
                e->last_fast_form = 2;
                fa.list_cap = FAST2_LIST_CAP;
                for (int grp = 0; grp < 2; grp++) {
                    if (!G.fast2_count[grp]) continue;
                    fa.lds_sc = G.fast2_lds_sc[grp]; fa.lds_list = G.fast2_lds_list[grp]; fa.lds_bm = G.fast2_lds_bm[grp]; fa.bm_rows = G.fast2_bm_rows[grp];
                    const dim3 grid2((G.fast2_count[grp] + 8 * FAST_XG - 1) / (8 * FAST_XG) * (8 * FAST_XG), batch);
                    hipLaunchKernelGGL((k_fast2<FAST2_P, FAST2_SP>), grid2, dim3(64), G.fast2_lds_bytes[grp], s, fa, (const PairRec *)e->d_pairs + G.fast2_first[grp],
                                       G.fast2_count[grp], pr, e->d_cell_cnt, e->d_cand, e->d_cand_prim);
                }
            } else
#define LAUNCH_FAST(NW_) hipLaunchKernelGGL((k_fast<48, 40, NW_>), grid, dim3(64 * NW_), lds_bytes, s, fa, e->d_cells, pr, e->d_cell_cnt, e->d_cand, e->d_cand_prim)
            if (nw == 4) LAUNCH_FAST(4); else if (nw == 3) LAUNCH_FAST(3); else if (nw == 2) LAUNCH_FAST(2);
            else if (grid.x <= 65535) hipLaunchKernelGGL((k_fast<48, 40, 1, true>), dim3(batch, grid.x), dim3(64), lds_bytes, s, fa, e->d_cells, pr, e->d_cell_cnt, e->d_cand, e->d_cand_prim);
            else LAUNCH_FAST(1);
#undef LAUNCH_FAST
        }
        else
            hipLaunchKernelGGL((k_fast<ORBX_TILE_PITCH, ORBX_SCORE_PITCH, 1>), grid, dim3(64), G.fast_lds_bytes, s, fa, e->d_cells, pr, e->d_cell_cnt,
                               e->d_cand, e->d_cand_prim);     // (cells wider than 38 px: scale factors far from 1.2; kept on the (cell, image) grid)
    }
    orbx_prof_end(e, s);
    int *err_flag = e->d_lvl_cnt + (size_t)e->max_batch * ORBX_MAX_LEVELS;
    orbx_prof_begin(e, ORBX_STAGE_TREE, s);
    {
        // 256 threads for batches (many (level, image) workgroups co-resident per CU); launches of few workgroups: 1024 threads
        // each shorten the per-workgroup chain (a single stereo frame: 51 -> 19 us)
        // (up to 512 workgroups -- 64 images of 8 levels: 16 frames 78 -> 85.5 k frames/s, 32 frames 106.5 -> 109 k; 1024 workgroups: slower)
        const bool big = batch * G.nlevels <= 512, lds = tree_tab_in_lds(G);
        void (*kern)(const Geom *, const int *, const uint32_t *, uint32_t *, uint16_t *, int *, uint32_t *, int, int *, unsigned char *, long long,
                     const uint32_t *, int) =
            big ? (lds ? k_tree<1024, true> : k_tree<1024, false>) : (lds ? k_tree<256, true> : k_tree<256, false>);
        hipLaunchKernelGGL(kern, dim3(batch, G.nlevels), dim3(big ? 1024 : 256), tree_launch_lds(G), s, e->d_geom,
                           e->d_cell_cnt, e->d_cand, e->d_tree_pts, e->d_tree_nid, e->d_lvl_cnt, e->d_lvl_kp, tree_launch_pts_cap(G), err_flag,
                           lds ? nullptr : e->d_tree_tab, (long long)align_up(tree_tab_bytes(G), 256), e->d_cand_prim, tree_reg_mode(G) ? 1 : 0);
    }
    orbx_prof_end(e, s);
    orbx_prof_begin(e, ORBX_STAGE_DESC, s);
    DescArgs da;
    memset(&da, 0, sizeof da);
    da.nlevels = G.nlevels; da.kp_total = G.kp_total;
    da.gauss = (unsigned)e->gauss[0] | (unsigned)e->gauss[1] << 8 | (unsigned)e->gauss[2] << 16 | (unsigned)e->gauss[3] << 24;
    for (int i = 0; i < ORBX_MAX_LEVELS; i++) {
        da.kp_off[i] = i < G.nlevels ? G.lv[i].kp_off : INT_MAX;
        if (i < G.nlevels) {
            const LevelGeom &L = G.lv[i];
            da.lv[i].w = L.w; da.lv[i].h = L.h; da.lv[i].pitch = L.pitch; da.lv[i].kp_off = L.kp_off; da.lv[i].pyr_off = L.pyr_off;
            da.lv[i].scale = L.scale; da.lv[i].patch_size = L.patch_size;
        }
    }
    RowTabArgs rt;
    memset(&rt, 0, sizeof rt);
    e->rt_kps = nullptr;
    if (e->d_rt_off && cap < 65536) {          // the stereo row table rides along (see desc_rowtab)
        rt.row_off = e->d_rt_off; rt.entries = (uint4 *)e->d_rt_entries; rt.ent_cap = e->rt_ent_cap; rt.rows = G.lv[0].h; rt.on = 1;
        e->rt_kps = d_kps; e->rt_cap = cap; e->rt_batch = batch;
    }
    hipLaunchKernelGGL((G.nlevels <= 8 ? k_desc<8> : k_desc<ORBX_MAX_LEVELS>), dim3((batch < 8 ? batch : 8) * (G.kp_total + rt.on), (batch + 7) / 8), dim3(64), 0, s, da, pr, e->d_lvl_cnt, e->d_lvl_kp,
                       (orbx_keypoint *)d_kps, (uint8_t *)d_desc, (int *)d_n_out, cap, batch, rt, (const int *)err_flag, e->flag_out);
    orbx_prof_end(e, s);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbx_sync(orbx_extractor *e, void *stream)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    // the kernel error flag rides the same stream into pinned memory: one synchronisation, no blocking pageable copy
    int *d_flag = e->d_lvl_cnt + (size_t)e->max_batch * ORBX_MAX_LEVELS;
    ORBX_HIP(hipMemcpyAsync(e->h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), s)); // the error belongs to the work synchronised here, not to later frames
    ORBX_HIP(hipStreamSynchronize(s));
    e->prof_chain = false; // the stream idles from here on: the next launch must not share its begin event with the last one
    const int flag = *e->h_flag;
    if (flag) { orbx_set_error("quadtree kernel reported a node-table overflow"); return ORBX_E_CAPACITY; }
    return ORBX_OK;
}

int orbx_quiesce(orbx_extractor *e)
{
    for (orbx_extractor *x : e->lanes) if (x) { const int lrc = orbx_quiesce(x); if (lrc) return lrc; }
    ORBX_HIP(hipStreamSynchronize(e->stream));
    if (e->last_launch_stream && e->last_launch_stream != e->stream) ORBX_HIP(hipStreamSynchronize(e->last_launch_stream));
    if (e->copy_in) ORBX_HIP(hipStreamSynchronize(e->copy_in));
    if (e->copy_out) ORBX_HIP(hipStreamSynchronize(e->copy_out));
    return ORBX_OK;
}

int orbx_ensure_out_staging(orbx_extractor *e, int batch, int cap)
{
    if (e->d_out_kps && e->out_cap >= cap && e->out_batch >= batch) return ORBX_OK;
    { const int qrc = orbx_quiesce(e); if (qrc) return qrc; }
    void **ps[] = { &e->d_out_kps, &e->d_out_desc, &e->d_out_n, (void **)&e->d_out_ur, (void **)&e->d_out_depth };
    for (void **p : ps) if (*p) { ORBX_HIP(hipFree(*p)); *p = nullptr; }
    const size_t n = (size_t)batch * cap;
    ORBX_HIP(hipMalloc(&e->d_out_kps, n * sizeof(orbx_keypoint)));
    ORBX_HIP(hipMalloc(&e->d_out_desc, n * 32));
    ORBX_HIP(hipMalloc(&e->d_out_n, sizeof(int) * batch));
    ORBX_HIP(hipMalloc((void **)&e->d_out_ur, n * 4));
    ORBX_HIP(hipMalloc((void **)&e->d_out_depth, n * 4));
    e->out_cap = cap; e->out_batch = batch;
    return ORBX_OK;
}

static int ensure_pinned(uint8_t **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return ORBX_OK;
    if (*p) { ORBX_HIP(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    ORBX_HIP(hipHostMalloc((void **)p, need ? need : 16, hipHostMallocDefault));
    *cap = need;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_extractor *e, const uint8_t *const *imgs, int batch, int w, int h, size_t stride,
                                  orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out)
{
    if (!e || !imgs || !kps || !desc || !n_out || batch < 1 || batch > e->max_batch || w < 0 || h < 0) {
        orbx_set_error("orbx_extract_batch: invalid argument");
        return ORBX_E_INVALID;
    }
    if (w == 0 || h == 0) { for (int i = 0; i < batch; i++) n_out[i] = 0; return ORBX_OK; } // reference :1264
    if (stride < (size_t)w) { orbx_set_error("stride < width"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    int rc = orbx_prepare_geometry(e, w, h);
    if (rc) return rc;
    const int need = e->geom.kp_total;
    if (cap < need) { orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d", cap, need); return ORBX_E_CAPACITY; }
    const size_t pitch = align_up(w, 64), img_bytes = pitch * h;
    if ((rc = ensure(&e->d_stage_in, &e->stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = ensure_pinned(&e->h_stage_in, &e->h_stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = orbx_ensure_out_staging(e, e->max_batch, need))) return rc;
    // pinned output image: [n | keypoints | descriptors] per batch
    const size_t o_kps = align_up(sizeof(int) * (size_t)e->max_batch, 64);
    const size_t o_desc = o_kps + align_up(sizeof(orbx_keypoint) * (size_t)need * e->max_batch, 64);
    const size_t out_bytes = o_desc + (size_t)32 * need * e->max_batch;
    if ((rc = ensure_pinned(&e->h_out, &e->h_out_cap, out_bytes))) return rc;
    // repitch into pinned memory on the host (a pageable 2-D copy is executed row by row by the runtime), one H2D copy
    for (int i = 0; i < batch; i++) {
        if (!imgs[i]) { orbx_set_error("imgs[%d] is NULL", i); return ORBX_E_INVALID; }
        uint8_t *dst = e->h_stage_in + img_bytes * i;
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * pitch, imgs[i] + (size_t)y * stride, (size_t)w);
    }
    ORBX_HIP(hipMemcpyAsync(e->d_stage_in, e->h_stage_in, img_bytes * batch, hipMemcpyHostToDevice, e->stream));
    e->prof_chain = false; // the copy above is not part of the first launch
    rc = orbx_extract_batch_device(e, e->d_stage_in, img_bytes, pitch, batch, w, h, e->d_out_kps, e->d_out_desc, need, e->d_out_n, nullptr);
    if (rc) return rc;
    // whole capacity back in three copies, one synchronisation
    ORBX_HIP(hipMemcpyAsync(e->h_out, e->d_out_n, sizeof(int) * batch, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_kps, e->d_out_kps, sizeof(orbx_keypoint) * (size_t)need * batch, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_desc, e->d_out_desc, (size_t)32 * need * batch, hipMemcpyDeviceToHost, e->stream));
    rc = orbx_sync(e, nullptr);
    if (rc) return rc;
    const int *hn = reinterpret_cast<const int *>(e->h_out);
    for (int i = 0; i < batch; i++) {
        n_out[i] = hn[i];
        memcpy(kps + (size_t)i * cap, e->h_out + o_kps + sizeof(orbx_keypoint) * (size_t)need * i, sizeof(orbx_keypoint) * (size_t)hn[i]);
        memcpy(desc + (size_t)i * cap * 32, e->h_out + o_desc + (size_t)32 * need * i, (size_t)32 * hn[i]);
    }
    return ORBX_OK;
}

// cv::cvtColor colour -> grey for 8U (call sites src/Tracking.cc:177-202): fixed point, yuv_shift 14
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ src, int w, int h, int spitch, int channels, int rgb_order,
                                              uint8_t *__restrict__ dst, int dpitch)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h || x4 >= dpitch) return;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 + i;
        if (x < w) {
            const uint8_t *p = src + (long long)y * spitch + (long long)x * channels;
            const int r = rgb_order ? p[0] : p[2], g = p[1], bl = rgb_order ? p[2] : p[0];
            out |= (uint32_t)((r * 4899 + g * 9617 + bl * 1868 + (1 << 13)) >> 14) << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t *>(dst + (long long)y * dpitch + x4) = out;
}

extern "C" int orbx_extract_color(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride, int channels, int rgb_order,
                                  orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, uint8_t *gray_out, size_t gray_stride)
{
    if (!e || !kps || !desc || !n_out || w < 0 || h < 0 || (channels != 3 && channels != 4) || (gray_out && gray_stride < (size_t)w)) {
        orbx_set_error("orbx_extract_color: invalid argument");
        return ORBX_E_INVALID;
    }
    if (w == 0 || h == 0) { *n_out = 0; return ORBX_OK; }
    if (!img || stride < (size_t)w * channels) { orbx_set_error("orbx_extract_color: bad image / stride"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    int rc = orbx_prepare_geometry(e, w, h);
    if (rc) return rc;
    const int need = e->geom.kp_total;
    if (cap < need) { orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d", cap, need); return ORBX_E_CAPACITY; }
    const size_t pitch = align_up(w, 64), img_bytes = pitch * h;
    const size_t cpitch = align_up((size_t)w * channels, 64), cbytes = cpitch * h;
    if ((rc = ensure(&e->d_stage_in, &e->stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = ensure_pinned(&e->h_stage_in, &e->h_stage_in_cap, cbytes > img_bytes * e->max_batch ? cbytes : img_bytes * e->max_batch))) return rc;
    void *d_color;
    if ((rc = orbx_scratch(e, 6, cbytes, &d_color))) return rc;
    if ((rc = orbx_ensure_out_staging(e, e->max_batch, need))) return rc;
    const size_t o_kps = align_up(sizeof(int) * (size_t)e->max_batch, 64);
    const size_t o_desc = o_kps + align_up(sizeof(orbx_keypoint) * (size_t)need * e->max_batch, 64);
    const size_t out_bytes = o_desc + (size_t)32 * need * e->max_batch;
    if ((rc = ensure_pinned(&e->h_out, &e->h_out_cap, out_bytes > img_bytes ? out_bytes : img_bytes))) return rc;
    for (int y = 0; y < h; y++) memcpy(e->h_stage_in + (size_t)y * cpitch, img + (size_t)y * stride, (size_t)w * channels);
    ORBX_HIP(hipMemcpyAsync(d_color, e->h_stage_in, cbytes, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_gray, dim3((unsigned)((pitch / 4 + 63) / 64), (h + 3) / 4), dim3(256), 0, e->stream, (const uint8_t *)d_color, w, h, (int)cpitch,
                       channels, rgb_order, e->d_stage_in, (int)pitch);
    e->prof_chain = false;
    rc = orbx_extract_batch_device(e, e->d_stage_in, img_bytes, pitch, 1, w, h, e->d_out_kps, e->d_out_desc, need, e->d_out_n, nullptr);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(e->h_out, e->d_out_n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_kps, e->d_out_kps, sizeof(orbx_keypoint) * (size_t)need, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_desc, e->d_out_desc, (size_t)32 * need, hipMemcpyDeviceToHost, e->stream));
    rc = orbx_sync(e, nullptr);
    if (rc) return rc;
    const int n = *reinterpret_cast<const int *>(e->h_out);
    *n_out = n;
    memcpy(kps, e->h_out + o_kps, sizeof(orbx_keypoint) * (size_t)n);
    memcpy(desc, e->h_out + o_desc, (size_t)32 * n);
    if (gray_out) ORBX_HIP(hipMemcpy2D(gray_out, gray_stride, e->d_stage_in, pitch, w, h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// EuRoC stereo front end (Examples/Stereo/stereo_euroc.cc:136-137 then Tracking::GrabImageStereo): the raw grey frame is
// uploaded, rectified on device (orbx_remap.hip) straight into the level-0 staging, and extracted.
extern "C" int orbx_extract_rectified(orbx_extractor *e, const orbx_rectifier *r, const uint8_t *img, int w, int h, size_t stride,
                                      orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, uint8_t *rect_out, size_t rect_stride)
{
    int rw = 0, rh = 0;
    if (!e || !r || !img || !kps || !desc || !n_out || w < 1 || h < 1 || stride < (size_t)w || orbx_rectifier_size(r, &rw, &rh) ||
        (rect_out && rect_stride < (size_t)rw)) {
        orbx_set_error("orbx_extract_rectified: invalid argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(orbx_use_device(e->device));
    int rc = orbx_prepare_geometry(e, rw, rh);
    if (rc) return rc;
    const int need = e->geom.kp_total;
    if (cap < need) { orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d", cap, need); return ORBX_E_CAPACITY; }
    const size_t pitch = align_up(rw, 64), img_bytes = pitch * rh;
    const size_t spitch = align_up((size_t)w, 64), sbytes = spitch * h;
    if ((rc = ensure(&e->d_stage_in, &e->stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = ensure_pinned(&e->h_stage_in, &e->h_stage_in_cap, sbytes > img_bytes * e->max_batch ? sbytes : img_bytes * e->max_batch))) return rc;
    void *d_raw;
    if ((rc = orbx_scratch(e, 6, sbytes, &d_raw))) return rc;
    if ((rc = orbx_ensure_out_staging(e, e->max_batch, need))) return rc;
    const size_t o_kps = align_up(sizeof(int) * (size_t)e->max_batch, 64);
    const size_t o_desc = o_kps + align_up(sizeof(orbx_keypoint) * (size_t)need * e->max_batch, 64);
    const size_t out_bytes = o_desc + (size_t)32 * need * e->max_batch;
    if ((rc = ensure_pinned(&e->h_out, &e->h_out_cap, out_bytes > img_bytes ? out_bytes : img_bytes))) return rc;
    for (int y = 0; y < h; y++) memcpy(e->h_stage_in + (size_t)y * spitch, img + (size_t)y * stride, (size_t)w);
    ORBX_HIP(hipMemcpyAsync(d_raw, e->h_stage_in, sbytes, hipMemcpyHostToDevice, e->stream));
    if ((rc = orbx_remap_batch_device(r, d_raw, sbytes, spitch, 1, e->d_stage_in, img_bytes, pitch, e->stream))) return rc;
    e->prof_chain = false;
    rc = orbx_extract_batch_device(e, e->d_stage_in, img_bytes, pitch, 1, rw, rh, e->d_out_kps, e->d_out_desc, need, e->d_out_n, nullptr);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(e->h_out, e->d_out_n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_kps, e->d_out_kps, sizeof(orbx_keypoint) * (size_t)need, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_desc, e->d_out_desc, (size_t)32 * need, hipMemcpyDeviceToHost, e->stream));
    rc = orbx_sync(e, nullptr);
    if (rc) return rc;
    const int n = *reinterpret_cast<const int *>(e->h_out);
    *n_out = n;
    memcpy(kps, e->h_out + o_kps, sizeof(orbx_keypoint) * (size_t)n);
    memcpy(desc, e->h_out + o_desc, (size_t)32 * n);
    if (rect_out) ORBX_HIP(hipMemcpy2D(rect_out, rect_stride, e->d_stage_in, pitch, rw, rh, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// One stereo frame through host pointers in one call: what Frame::Frame(imLeft, imRight, ...) does with two ExtractORB
// threads and ComputeStereoMatches (src/Frame.cc:82-97): both eyes in one H2D copy and one batch-of-2 extraction, the
// stereo matcher on the still device-resident keypoints, everything back in one group of copies behind ONE
// synchronisation (the separate calls cost three synchronisations and an extra round trip of both eyes' features).
extern "C" int orbx_extract_stereo(orbx_extractor *e, const uint8_t *img_left, const uint8_t *img_right, int w, int h, size_t stride,
                                   float bf, float min_z, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                                   float *u_right, float *depth)
{
    if (!e || !img_left || !img_right || !kps || !desc || !n_out || !u_right || !depth || w < 1 || h < 1 || stride < (size_t)w ||
        !(min_z > 0)) {
        orbx_set_error("orbx_extract_stereo: invalid argument");
        return ORBX_E_INVALID;
    }
    if (e->max_batch < 2) { orbx_set_error("orbx_extract_stereo needs an extractor created with max_batch >= 2"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    int rc = orbx_prepare_geometry(e, w, h);
    if (rc) return rc;
    const int need = e->geom.kp_total;
    if (cap < need) { orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d", cap, need); return ORBX_E_CAPACITY; }
    const size_t pitch = align_up(w, 64), img_bytes = pitch * h;
    if ((rc = ensure(&e->d_stage_in, &e->stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = ensure_pinned(&e->h_stage_in, &e->h_stage_in_cap, img_bytes * e->max_batch))) return rc;
    if ((rc = orbx_ensure_out_staging(e, e->max_batch, need))) return rc;
    const size_t o_kps = 64, o_desc = o_kps + align_up(sizeof(orbx_keypoint) * (size_t)need * 2, 64), o_ur = o_desc + (size_t)64 * need,
                 o_z = o_ur + align_up(4 * (size_t)need, 64), out_bytes = o_z + align_up(4 * (size_t)need, 64);
    if ((rc = ensure_pinned(&e->h_out, &e->h_out_cap, out_bytes))) return rc;
    const uint8_t *eyes[2] = { img_left, img_right };
    for (int i = 0; i < 2; i++) {
        uint8_t *dst = e->h_stage_in + img_bytes * i;
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * pitch, eyes[i] + (size_t)y * stride, (size_t)w);
    }
    ORBX_HIP(hipMemcpyAsync(e->d_stage_in, e->h_stage_in, img_bytes * 2, hipMemcpyHostToDevice, e->stream));
    e->prof_chain = false;
    rc = orbx_extract_batch_device(e, e->d_stage_in, img_bytes, pitch, 2, w, h, e->d_out_kps, e->d_out_desc, need, e->d_out_n, nullptr);
    if (rc) return rc;
    orbx_keypoint *dk = (orbx_keypoint *)e->d_out_kps;
    uint8_t *dd = (uint8_t *)e->d_out_desc;
    int *dn = (int *)e->d_out_n;
    // the right keypoints are what the launch above wrote into the handle's own buffer: its by-product row table serves when it built one
    rc = orbx_stereo_match_batch_device(e, 0, e, 1, 1, dk, dd, dn, dk + need, dd + (size_t)32 * need, dn + 1, need, bf, min_z,
                                        e->d_out_ur, e->d_out_depth,
                                        orbx_stereo_row_table_available(e, dk + need, 1, 1, need) ? ORBX_ROWTAB_OF_EXTRACTION : ORBX_ROWTAB_FROM_KEYPOINTS, nullptr);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(e->h_out, dn, sizeof(int) * 2, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_kps, dk, sizeof(orbx_keypoint) * (size_t)need * 2, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_desc, dd, (size_t)64 * need, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_ur, e->d_out_ur, 4 * (size_t)need, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(e->h_out + o_z, e->d_out_depth, 4 * (size_t)need, hipMemcpyDeviceToHost, e->stream));
    rc = orbx_sync(e, nullptr);
    if (rc) return rc;
    const int *hn = reinterpret_cast<const int *>(e->h_out);
    for (int i = 0; i < 2; i++) {
        n_out[i] = hn[i];
        memcpy(kps + (size_t)i * cap, e->h_out + o_kps + sizeof(orbx_keypoint) * (size_t)need * i, sizeof(orbx_keypoint) * (size_t)hn[i]);
        memcpy(desc + (size_t)i * cap * 32, e->h_out + o_desc + (size_t)32 * need * i, (size_t)32 * hn[i]);
    }
    memcpy(u_right, e->h_out + o_ur, 4 * (size_t)hn[0]);
    memcpy(depth, e->h_out + o_z, 4 * (size_t)hn[0]);
    return ORBX_OK;
}

// ---- pipelined host-pointer stereo frames (a camera stream fed from host memory)

extern "C" int orbx_pipeline_depth(void) { return ORBX_PIPE_DEPTH; }

extern "C" void *orbx_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { orbx_set_error("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" void orbx_pinned_free(void *p) { if (p) hipHostFree(p); }

// the address a kernel reads pinned (page-locked, mapped) host memory at, or nullptr for anything else
static const uint8_t *pinned_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return a.type == hipMemoryTypeHost ? (const uint8_t *)a.devicePointer : nullptr;
}

static int pipe_slot_prepare(orbx_extractor *e, PipeSlot &s, size_t in_bytes, int need)
{
    if (!s.ev_h2d) {
        ORBX_HIP(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
        ORBX_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        ORBX_HIP(hipEventCreateWithFlags(&s.ev_d2h, hipEventDisableTiming));
    }
    if (in_bytes > s.in_cap) {
        if (s.h_in) ORBX_HIP(hipHostFree(s.h_in));
        if (s.d_in) ORBX_HIP(hipFree(s.d_in));
        s.h_in = nullptr; s.d_in = nullptr; s.in_cap = 0;
        ORBX_HIP(hipHostMalloc((void **)&s.h_in, in_bytes, hipHostMallocDefault));
        ORBX_HIP(hipHostGetDevicePointer((void **)&s.h_in_dev, s.h_in, 0));
        ORBX_HIP(hipMalloc((void **)&s.d_in, in_bytes));
        s.in_cap = in_bytes;
    }
    const size_t o_kps = 64, o_desc = o_kps + align_up(sizeof(orbx_keypoint) * 2 * (size_t)need, 64), o_ur = o_desc + (size_t)64 * need,
                 o_z = o_ur + align_up(4 * (size_t)need, 64), out_bytes = o_z + align_up(4 * (size_t)need, 64);
    if (need > s.out_cap) {
        if (s.d_out) ORBX_HIP(hipFree(s.d_out));
        s.d_out = nullptr; s.out_cap = 0;
        ORBX_HIP(hipMalloc((void **)&s.d_out, out_bytes));
        s.out_cap = need;
    }
    if (out_bytes > s.h_out_cap) {
        if (s.h_out) ORBX_HIP(hipHostFree(s.h_out));
        s.h_out = nullptr; s.h_out_cap = 0;
        ORBX_HIP(hipHostMalloc((void **)&s.h_out, out_bytes, hipHostMallocDefault));
        ORBX_HIP(hipHostGetDevicePointer((void **)&s.h_out_dev, s.h_out, 0));
        s.h_out_cap = out_bytes;
    }
    // the views follow `need` (the layout of this frame), not the capacity the block was allocated for
    s.d_n = s.d_out; s.d_kps = s.d_out + o_kps; s.d_desc = s.d_out + o_desc;
    s.d_ur = reinterpret_cast<float *>(s.d_out + o_ur); s.d_z = reinterpret_cast<float *>(s.d_out + o_z);
    return ORBX_OK;
}

// Frame transport of the pipelined forms by KERNEL: the caller's pinned images are read over PCIe by a copy kernel on the frame's own
// lane stream, and the result block is written to the slot's pinned buffer the same way.  The copy engines move a 466 KB image in
// ~25 us each (0.93 MB per stereo frame: a ceiling of ~15 k frames/s whatever the number of lanes) and need a stream hop with two
// events each way; a kernel with enough loads in flight moves the frame in ~20 us and is just one more launch of the chain.
__global__ __launch_bounds__(256) void k_copy_bytes(const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1, uint8_t *__restrict__ dst0,
                                                    uint8_t *__restrict__ dst1, unsigned long long bytes)
{
    const uint8_t *src = blockIdx.y ? src1 : src0;
    uint8_t *dst = blockIdx.y ? dst1 : dst0;
    const unsigned long long n16 = bytes >> 4, stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        uint4 v;
        __builtin_memcpy(&v, src + 16 * i, 16);             // any byte alignment (global memory takes unaligned dwordx4)
        __builtin_memcpy(dst + 16 * i, &v, 16);
    }
    if (blockIdx.x == 0) for (unsigned long long i = (n16 << 4) + threadIdx.x; i < bytes; i += 256) dst[i] = src[i];
}

static std::atomic<int> g_pipe_handles{0};   // handles of this process that have submitted pipelined frames and still exist
void orbx_pipe_handle_released() { g_pipe_handles.fetch_sub(1, std::memory_order_relaxed); }

// one frame into the next pipeline slot: eyes = 2 (stereo: both extractions + ComputeStereoMatches) or 1 (mono: extraction only)
// ORBX_PIPE_PROF: host nanoseconds per section of pipe_submit / pipe_wait, summed over every client thread (relaxed atomics: diagnostic only)
static std::atomic<long long> g_pp[8]; static std::atomic<long> g_pp_n;
static inline double pp_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_pp_on = getenv("ORBX_PIPE_PROF") != nullptr;
struct PpScope {    // charges the time since construction / the last next() to section k
    int k; double t0;
    explicit PpScope(int k_) : k(k_), t0(g_pp_on ? pp_now() : 0) {}
    void next(int k_) { if (g_pp_on) { const double t = pp_now(); g_pp[k].fetch_add((long long)((t - t0) * 1e3), std::memory_order_relaxed); t0 = t; } k = k_; }
    ~PpScope() { if (g_pp_on) g_pp[k].fetch_add((long long)((pp_now() - t0) * 1e3), std::memory_order_relaxed); }
};
extern "C" void orbx_debug_pipe_prof_print()
{
    const long frames = g_pp_n.load();
    if (!g_pp_on || !frames) return;
    const char *nm[8] = { "setdevice+lane+geometry", "pinned test + slot", "upload enqueue", "extract launches", "stereo launch", "download enqueue + event", "wait: event sync", "wait: copy out" };
    for (int i = 0; i < 8; i++) fprintf(stderr, "pipe prof: %-28s %7.2f us per frame\n", nm[i], g_pp[i].load() * 1e-3 / frames);
}

static int pipe_submit(orbx_extractor *e, const uint8_t *img_left, const uint8_t *img_right, int eyes, int w, int h, size_t stride,
                       float bf, float min_z, int *ticket)
{
    if (g_pp_on) g_pp_n.fetch_add(1, std::memory_order_relaxed);
    PpScope pp(0);
#define PP_NEXT(K) pp.next(K)
    ORBX_HIP(orbx_use_device(e->device));
    // Kernel lanes: a single stereo frame is a chain of dependent launches (~70 us) that keeps a few percent of the chip busy, so
    // consecutive frames go round the handle and its shadow handles (own stream, own pyramid / candidate / quadtree workspaces,
    // created on first use) and their chains overlap.  Slots, tickets and the order of results are unchanged.
    // (With the copy-engine transport of round 2 lanes only paid for a lone handle -- four client threads: 11.3 k frames/s with one lane
    // each, 8.2 k with two --; with the kernel transport they pay for every handle: two camera streams 14.7 k -> 17.3 k, four 20 k either way.)
    if (!e->pipe_counted) { e->pipe_counted = true; g_pipe_handles.fetch_add(1, std::memory_order_relaxed); }
    orbx_extractor *x = e;
    const int nl = e->pipe_lanes, li = (int)(e->pipe_next % (unsigned)nl);
    if (li) {
        orbx_extractor *&sh = e->lanes[li - 1];
        if (!sh) {
            const int lrc = orbx_extractor_create(&sh, e->nfeatures, (float)e->scale_factor, e->nlevels, e->ini_th, e->min_th, e->device, e->max_w, e->max_h, 2);
            if (lrc) return lrc;
        }
        x = sh;
        if (x->cv_profile != e->cv_profile) orbx_extractor_set_cv_profile(x, e->cv_profile);
        x->pyr_group_max_images = e->pyr_group_max_images; x->pyr_group_mid_images = e->pyr_group_mid_images;
    }
    int rc = orbx_prepare_geometry(x, w, h);   // waits for everything in flight only when the image size changes
    if (rc) return rc;
    PP_NEXT(1);
    PipeSlot &s = e->pipe[e->pipe_next % ORBX_PIPE_DEPTH];
    if (s.busy) { orbx_set_error("all %d pipeline slots are in flight: wait for the oldest ticket first", ORBX_PIPE_DEPTH); return ORBX_E_INVALID; }
    const int need = x->geom.kp_total;
    const uint8_t *pin[2] = { stride == (size_t)w ? pinned_device_ptr(img_left) : nullptr, stride == (size_t)w && eyes == 2 ? pinned_device_ptr(img_right) : nullptr };
    const bool in_place = pin[0] && (eyes == 1 || pin[1]);
    const size_t pitch = in_place ? (size_t)w : align_up(w, 64), img_bytes = pitch * h;
    if (!e->copy_in) {
        ORBX_HIP(hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking));
        ORBX_HIP(hipStreamCreateWithFlags(&e->copy_out, hipStreamNonBlocking));
    }
    if ((rc = pipe_slot_prepare(e, s, 2 * align_up(w, 64) * (size_t)h, need))) return rc;
    // upload: the slot's device input was last read by the kernels of the frame that used it ORBX_PIPE_DEPTH submissions ago,
    // which its _wait has already seen finish (ev_d2h follows ev_done), so the copy stream may overwrite it right away
    const uint8_t *eye_ptr[2] = { img_left, img_right };
    PP_NEXT(2);
    // (inline form: upload, kernels and download of a frame all on its lane's stream -- no events, no stream hops; the lanes overlap each other)
    hipStream_t s_in = e->pipe_inline ? x->stream : e->copy_in, s_out = e->pipe_inline ? x->stream : e->copy_out;
    const bool kcopy = e->pipe_inline && e->pipe_kcopy;
    if (in_place && kcopy) {
        hipLaunchKernelGGL(k_copy_bytes, dim3(128, eyes), dim3(256), 0, x->stream, pin[0], pin[1], s.d_in, s.d_in + img_bytes, (unsigned long long)img_bytes);
    } else if (in_place) {
        for (int i = 0; i < eyes; i++) ORBX_HIP(hipMemcpyAsync(s.d_in + img_bytes * i, eye_ptr[i], img_bytes, hipMemcpyHostToDevice, s_in));
    } else {
        for (int i = 0; i < eyes; i++) {
            uint8_t *dst = s.h_in + img_bytes * i;
            if (stride == pitch) memcpy(dst, eye_ptr[i], img_bytes);
            else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * pitch, eye_ptr[i] + (size_t)y * stride, (size_t)w);
        }
        if (kcopy) hipLaunchKernelGGL(k_copy_bytes, dim3(128, eyes), dim3(256), 0, x->stream, (const uint8_t *)s.h_in_dev, (const uint8_t *)s.h_in_dev + img_bytes, s.d_in, s.d_in + img_bytes,
                                      (unsigned long long)img_bytes);
        else ORBX_HIP(hipMemcpyAsync(s.d_in, s.h_in, img_bytes * eyes, hipMemcpyHostToDevice, s_in));
    }
    if (!e->pipe_inline) {
        ORBX_HIP(hipEventRecord(s.ev_h2d, e->copy_in));
        ORBX_HIP(hipStreamWaitEvent(x->stream, s.ev_h2d, 0));
    }
    x->prof_chain = false;
    orbx_keypoint *dk = (orbx_keypoint *)s.d_kps;
    uint8_t *dd = (uint8_t *)s.d_desc;
    int *dn = (int *)s.d_n;
    // the kernel error flag of this frame travels with its counts (k_desc drops it into the slot's block).  It is sticky on the device
    // (a node-table overflow is a configuration error, not a per-frame event): pipe_wait clears it when it reports it
    x->flag_out = dn + 2;
    PP_NEXT(3);
    rc = orbx_extract_batch_device(x, s.d_in, img_bytes, pitch, eyes, w, h, s.d_kps, s.d_desc, need, s.d_n, nullptr);
    x->flag_out = nullptr;
    if (rc) return rc;
    PP_NEXT(4);
    if (eyes == 2) {
        rc = orbx_stereo_match_batch_device(x, 0, x, 1, 1, dk, dd, dn, dk + need, dd + (size_t)32 * need, dn + 1, need, bf, min_z, s.d_ur, s.d_z,
                                            orbx_stereo_row_table_available(x, dk + need, 1, 1, need) ? ORBX_ROWTAB_OF_EXTRACTION : ORBX_ROWTAB_FROM_KEYPOINTS, nullptr);
        if (rc) return rc;
    }
    PP_NEXT(5);
    // download: one copy of the slot's block (counts + flag, both eyes' keypoints and descriptors, uRight, depth)
    const size_t o_kps = 64, o_desc = o_kps + align_up(sizeof(orbx_keypoint) * 2 * (size_t)need, 64), o_ur = o_desc + (size_t)64 * need,
                 o_z = o_ur + align_up(4 * (size_t)need, 64);
    if (!e->pipe_inline) {
        ORBX_HIP(hipEventRecord(s.ev_done, x->stream));
        ORBX_HIP(hipStreamWaitEvent(e->copy_out, s.ev_done, 0));
    }
    {
        const size_t out_bytes = eyes == 2 ? o_z + 4 * (size_t)need : o_desc + (size_t)32 * eyes * need;
        if (kcopy) hipLaunchKernelGGL(k_copy_bytes, dim3(64, 1), dim3(256), 0, x->stream, (const uint8_t *)s.d_out, (const uint8_t *)nullptr, s.h_out_dev, (uint8_t *)nullptr,
                                      (unsigned long long)out_bytes);
        else ORBX_HIP(hipMemcpyAsync(s.h_out, s.d_out, out_bytes, hipMemcpyDeviceToHost, s_out));
    }
    s.lane = x;
    ORBX_HIP(hipEventRecord(s.ev_d2h, s_out));
    // tickets are the low 31 bits of an unsigned submit counter: never negative, and (the depth divides 2^31) still congruent to the slot
    s.busy = true; s.cap = need; s.ticket = (int)(e->pipe_next & 0x7FFFFFFFu); s.eyes = eyes;
    *ticket = s.ticket;
    e->pipe_next++;
    return ORBX_OK;
#undef PP_NEXT
}

static int pipe_wait(orbx_extractor *e, int ticket, int eyes, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, float *u_right, float *depth)
{
    PipeSlot &s = e->pipe[ticket % ORBX_PIPE_DEPTH];
    if (!s.busy || s.ticket != ticket || s.eyes != eyes) { orbx_set_error("ticket %d is not in flight (or was submitted through the other form)", ticket); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    const int need = s.cap;
    if (cap < need) { orbx_set_error("keypoint capacity %d < orbx_max_keypoints() = %d (the ticket stays valid)", cap, need); return ORBX_E_CAPACITY; }
    { PpScope w6(6); ORBX_HIP(hipEventSynchronize(s.ev_d2h)); }
    PpScope w7(7);
    s.busy = false;
    const size_t o_kps = 64, o_desc = o_kps + align_up(sizeof(orbx_keypoint) * 2 * (size_t)need, 64), o_ur = o_desc + (size_t)64 * need,
                 o_z = o_ur + align_up(4 * (size_t)need, 64);
    const int *hn = reinterpret_cast<const int *>(s.h_out);
    if (hn[2]) {
        orbx_extractor *x = s.lane ? s.lane : e;
        hipMemsetAsync(x->d_lvl_cnt + (size_t)x->max_batch * ORBX_MAX_LEVELS, 0, sizeof(int), x->stream);
        orbx_set_error("quadtree kernel reported a node-table overflow");
        return ORBX_E_CAPACITY;
    }
    for (int i = 0; i < eyes; i++) {
        n_out[i] = hn[i];
        memcpy(kps + (size_t)i * cap, s.h_out + o_kps + sizeof(orbx_keypoint) * (size_t)need * i, sizeof(orbx_keypoint) * (size_t)hn[i]);
        memcpy(desc + (size_t)i * cap * 32, s.h_out + o_desc + (size_t)32 * need * i, (size_t)32 * hn[i]);
    }
    if (eyes == 2) {
        memcpy(u_right, s.h_out + o_ur, 4 * (size_t)hn[0]);
        memcpy(depth, s.h_out + o_z, 4 * (size_t)hn[0]);
    }
    return ORBX_OK;
}

// Every pipeline slot and every kernel lane (the handle and its shadow handles: own stream, pyramid / candidate / quadtree / stereo workspaces,
// pinned result block) of the pipelined forms is made and touched by running ONE frame of a textured scratch image through each, so that the
// frames that follow all see the steady-state latency.  Called by the first submit of a handle (and again when the image size changes); a
// caller that wants even its first frame on time calls it beforehand.  Round 3 made lanes and slots lazily, one per frame: the first four
// frames of a stream -- inside any timed window -- took 3-19 ms each against 0.2 ms.
extern "C" int orbx_pipeline_warm(orbx_extractor *e, int w, int h)
{
    if (!e || w < 1 || h < 1) { orbx_set_error("orbx_pipeline_warm: invalid argument"); return ORBX_E_INVALID; }
    if (e->max_batch < 2) { orbx_set_error("orbx_pipeline_warm needs an extractor created with max_batch >= 2"); return ORBX_E_INVALID; }
    if (e->pipe_warm_w == w && e->pipe_warm_h == h) return ORBX_OK;
    for (const PipeSlot &s : e->pipe) if (s.busy) { orbx_set_error("orbx_pipeline_warm: frames are in flight"); return ORBX_E_INVALID; }
    std::vector<uint8_t> img((size_t)w * h);
    unsigned lcg = 2463534242u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            lcg = lcg * 1664525u + 1013904223u;
            img[(size_t)y * w + x] = (uint8_t)((((x >> 4) * 37 + (y >> 4) * 91) & 127) + 40 + (lcg >> 29));   // 16-px blocks: corners on every level
        }
    e->pipe_warm_w = w; e->pipe_warm_h = h;      // (set first: the submits below must not come back here)
    const int need = orbx_max_keypoints(e, w, h);
    int rc = need < 0 ? need : ORBX_OK;
    std::vector<orbx_keypoint> kps(rc ? 0 : 2 * (size_t)need);
    std::vector<uint8_t> desc(rc ? 0 : (size_t)64 * need);
    std::vector<float> ur(rc ? 0 : (size_t)need), z(rc ? 0 : (size_t)need);
    int tickets[ORBX_PIPE_DEPTH], n[2], nsub = 0;
    for (int i = 0; i < ORBX_PIPE_DEPTH && !rc; i++) {          // one frame per slot; the lanes go round with the slots
        rc = pipe_submit(e, img.data(), img.data(), 2, w, h, (size_t)w, 386.1448f, 0.5372f, &tickets[i]);
        if (!rc) nsub++;
    }
    for (int i = 0; i < nsub; i++) {
        const int wrc = pipe_wait(e, tickets[i], 2, kps.data(), desc.data(), need, n, ur.data(), z.data());
        if (wrc && !rc) rc = wrc;
    }
    if (rc) { e->pipe_warm_w = 0; e->pipe_warm_h = 0; }
    return rc;
}

extern "C" int orbx_extract_stereo_submit(orbx_extractor *e, const uint8_t *img_left, const uint8_t *img_right, int w, int h, size_t stride,
                                          float bf, float min_z, int *ticket)
{
    if (!e || !img_left || !img_right || !ticket || w < 1 || h < 1 || stride < (size_t)w || !(min_z > 0)) {
        orbx_set_error("orbx_extract_stereo_submit: invalid argument");
        return ORBX_E_INVALID;
    }
    if (e->max_batch < 2) { orbx_set_error("orbx_extract_stereo_submit needs an extractor created with max_batch >= 2"); return ORBX_E_INVALID; }
    if (e->pipe_warm_w != w || e->pipe_warm_h != h) {           // first frame of this size: make every lane and slot now, not one per frame
        bool idle = true;
        for (const PipeSlot &s : e->pipe) idle = idle && !s.busy;
        if (idle) { const int wrc = orbx_pipeline_warm(e, w, h); if (wrc) return wrc; }
    }
    return pipe_submit(e, img_left, img_right, 2, w, h, stride, bf, min_z, ticket);
}

extern "C" int orbx_extract_stereo_wait(orbx_extractor *e, int ticket, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                                        float *u_right, float *depth)
{
    if (!e || !kps || !desc || !n_out || !u_right || !depth || ticket < 0) { orbx_set_error("orbx_extract_stereo_wait: invalid argument"); return ORBX_E_INVALID; }
    return pipe_wait(e, ticket, 2, kps, desc, cap, n_out, u_right, depth);
}

extern "C" int orbx_extract_submit(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride, int *ticket)
{
    if (!e || !img || !ticket || w < 1 || h < 1 || stride < (size_t)w) { orbx_set_error("orbx_extract_submit: invalid argument"); return ORBX_E_INVALID; }
    return pipe_submit(e, img, nullptr, 1, w, h, stride, 0.f, 1.f, ticket);
}

extern "C" int orbx_extract_wait(orbx_extractor *e, int ticket, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out)
{
    if (!e || !kps || !desc || !n_out || ticket < 0) { orbx_set_error("orbx_extract_wait: invalid argument"); return ORBX_E_INVALID; }
    return pipe_wait(e, ticket, 1, kps, desc, cap, n_out, nullptr, nullptr);
}

extern "C" int orbx_extract(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride,
                            orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out)
{
    const uint8_t *imgs[1] = { img };
    if (!img && w > 0 && h > 0) { orbx_set_error("img is NULL"); return ORBX_E_INVALID; }
    return orbx_extract_batch(e, imgs, 1, w, h, stride, kps, desc, cap, n_out);
}

extern "C" int orbx_pyramid_level(orbx_extractor *e, int image_index, int level, uint8_t *dst, size_t dst_stride, int *w, int *h)
{
    if (!e || !e->geom.w || !e->last_img0 || level < 0 || level >= e->nlevels || image_index < 0 || image_index >= e->last_batch) {
        orbx_set_error("orbx_pyramid_level: no pyramid / bad index");
        return ORBX_E_INVALID;
    }
    const LevelGeom &L = e->geom.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return ORBX_OK;
    if (dst_stride < (size_t)L.w) { orbx_set_error("dst_stride < level width"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    const uint8_t *src; size_t pitch;
    if (level == 0) { src = e->last_img0 + e->last_img_stride * image_index; pitch = e->last_pitch; }
    else { src = e->d_pyr + (size_t)e->geom.pyr_bytes * image_index + L.pyr_off; pitch = L.pitch; }
    ORBX_HIP(hipMemcpy2D(dst, dst_stride, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// which FAST kernel the most recent extraction launched: 1 = k_fast (a cell per wave, or several waves per cell), 2 = k_fast2 (a pair of cells per wave)
extern "C" int orbx_debug_fast_form(const orbx_extractor *e) { return e ? e->last_fast_form : ORBX_E_INVALID; }

extern "C" int orbx_debug_level_counts(orbx_extractor *e, int image_index, int32_t *counts)
{
    if (!e || !counts || image_index < 0 || image_index >= e->last_batch) { orbx_set_error("bad argument"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    ORBX_HIP(hipMemcpy(counts, e->d_lvl_cnt + (size_t)image_index * ORBX_MAX_LEVELS, sizeof(int) * e->nlevels, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_debug_candidates(orbx_extractor *e, int image_index, int level, int32_t *x, int32_t *y, int32_t *resp, int cap, int *n)
{
    if (!e || !n || !e->geom.w || level < 0 || level >= e->nlevels || image_index < 0 || image_index >= e->last_batch) {
        orbx_set_error("bad argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(orbx_use_device(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    const Geom &G = e->geom;
    const LevelGeom &L = G.lv[level];
    std::vector<int> cnt(L.n_cells);
    std::vector<uint32_t> slots((size_t)L.n_cells * L.cand_cap);
    ORBX_HIP(hipMemcpy(cnt.data(), e->d_cell_cnt + (size_t)image_index * G.total_cells + L.cell_base, sizeof(int) * L.n_cells, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(slots.data(), e->d_cand + (size_t)image_index * G.cand_total + L.cand_off, slots.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> prim((size_t)L.n_cells * ORBX_CAND_PRIM);
    ORBX_HIP(hipMemcpy(prim.data(), e->d_cand_prim + ((size_t)image_index * G.total_cells + L.cell_base) * ORBX_CAND_PRIM, prim.size() * 4, hipMemcpyDeviceToHost));
    int k = 0;
    for (int c = 0; c < L.n_cells; c++)
        for (int i = 0; i < cnt[c]; i++, k++)
            if (k < cap && x && y && resp) {
                const uint32_t p = i < ORBX_CAND_PRIM ? prim[(size_t)c * ORBX_CAND_PRIM + i] : slots[(size_t)c * L.cand_cap + i];
                x[k] = p & 0xFFF; y[k] = (p >> 12) & 0xFFF; resp[k] = p >> 24;
            }
    *n = k;
    return ORBX_OK;
}

int orbx_scratch(orbx_extractor *e, int slot, size_t bytes, void **out)
{
    if (bytes > e->scratch_cap[slot] || !e->scratch[slot]) {
        { const int qrc = orbx_quiesce(e); if (qrc) return qrc; }
        if (e->scratch[slot]) { ORBX_HIP(hipFree(e->scratch[slot])); e->scratch[slot] = nullptr; e->scratch_cap[slot] = 0; }
        ORBX_HIP(hipMalloc(&e->scratch[slot], bytes ? bytes : 16));
        e->scratch_cap[slot] = bytes;
    }
    *out = e->scratch[slot];
    return ORBX_OK;
}
