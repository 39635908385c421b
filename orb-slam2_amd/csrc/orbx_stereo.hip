// orbx_stereo.hip — Frame::ComputeStereoMatches (reference src/Frame.cc:577-751) for gfx950.
//
// K5a k_stereo_prep: the row table vRowIndices of :584-604 as a CSR in HBM, one workgroup per pair.
// K5 k_stereo: one wave per left keypoint.  Coarse stage: the right keypoints listed for the left
// keypoint's row are tested 64 at a time against the octave / disparity-range predicates of
// :644-649, 256-bit Hamming with v_bcnt, wave min of (dist<<16 | iR) reproduces "first minimum in
// ascending iR" (:654-658).  Fine stage: 11x11
// centre-subtracted L1 SAD over 11 shifts on the unblurred pyramids (:666-703), parabola (:709-716),
// disparity / depth (:719-733) by the same wave.
// K6 median cut: per pair, radix-select the median SAD and drop matches >= 1.5*1.4*median (:737-750) -- run by the LAST workgroup of
// k_stereo to finish a pair (device-scope arrival counter), not by a launch of its own.
// The row table normally comes with the extraction (desc_rowtab in orbx_extract.hip); k_stereo_prep is the form for keypoints that
// did not come from the right extractor's last launch.
#include "orbx_device.h"
#include <stdlib.h>
#include <algorithm>

struct StereoTabs {
    float sf[ORBX_MAX_LEVELS]; float isf[ORBX_MAX_LEVELS];
    unsigned long long reach_pk;    // 4 bits per left level l: centre rows within +- that many can hold a right keypoint of octave l - 1 .. l + 1
                                    // whose band covers the row (15 = no bound below the launch's `reach`); packed: the level is a per-lane value
};

// pixel of the reference's padded pyramid image: the 19-px margin is BORDER_REFLECT_101 of the level
__device__ __forceinline__ int lvl_px(const uint8_t *img, int pitch, int w, int h, int x, int y)
{
    return img[(long long)reflect101(y, h) * pitch + reflect101(x, w)];
}

// a keypoint's results as write-through stores (sc1): visible to the pair's last workgroup on another XCD once acknowledged
template <bool FOLD>
__device__ __forceinline__ void st_store(Published<float> u_right, Published<float> depth, Published<int> st_dist, long long og, float u, float z, int sad)
{
    if (FOLD) { u_right.put(og, u); depth.put(og, z); st_dist.put(og, sad); }       // relaxed publish protocol (orbx_device.h)
    else {                                                                          // the cut is a launch of its own: ordinary stores
        u_right.plain_across_launches()[og] = u; depth.plain_across_launches()[og] = z; st_dist.plain_across_launches()[og] = sad;
    }
}

// Row table of src/Frame.cc:584-604 (vRowIndices).  The reference lists right keypoint iR in every image row of its band
// [minr, maxr] = [floor(y - r), ceil(y + r)], r = 2 * scale[octave] (7000-17000 list entries for 1000 keypoints).  Here a keypoint is
// listed ONCE, under the row of its centre, with its band in the entry (iR | octave << 16, x, minr | maxr << 16, 0): CSR by centre row,
// row_off[rows + 1] + one 16-byte entry per keypoint.  The coarse stage scans the contiguous run of centre rows that can reach the left
// keypoint's row (+- ceil(2 * scale[top]) + 1) and tests minr <= row <= maxr -- the same candidate set as the reference's per-row lists
// (order is irrelevant: the stage reduces with min(dist << 16 | iR), "first minimum in ascending iR"), built with one LDS atomic per
// keypoint instead of one per (keypoint, row).
// k_stereo_prep builds it for keypoints handed in by the caller; keypoints that come straight from the right extractor's last launch
// already have theirs (desc_rowtab in orbx_extract.hip: same layout).
extern __shared__ __align__(16) int prep_smem[];

__global__ __launch_bounds__(256) void k_stereo_prep(const Geom *__restrict__ g, const orbx_keypoint *__restrict__ kR,
                                                     const int *__restrict__ nR, int cap, StereoTabs tabs,
                                                     int *__restrict__ row_off, uint4 *__restrict__ entries, int ent_cap)
{
    __shared__ int s_w[4];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int rows = g->lv[0].h, n_r = min(nR[p], ent_cap);
    int *cnt = prep_smem, *cur = prep_smem + ((rows + 4) & ~3);
    const orbx_keypoint *kr = kR + (long long)p * cap;
    for (int i = tid; i < rows; i += 256) { cnt[i] = 0; cur[i] = 0; }
    __syncthreads();
    for (int ir = tid; ir < n_r; ir += 256) atomicAdd(&cnt[min(max((int)floorf(kr[ir].y), 0), rows - 1)], 1);
    __syncthreads();
    const int total = lds_excl_scan(cnt, rows, s_w);
    int *ro = row_off + (long long)p * (rows + 1);
    for (int i = tid; i < rows; i += 256) ro[i] = cnt[i];
    if (tid == 0) ro[rows] = total;
    uint4 *en = entries + (long long)p * ent_cap;
    for (int ir = tid; ir < n_r; ir += 256) {
        const orbx_keypoint q = kr[ir];
        const float y = q.y, r = 2.0f * tabs.sf[q.octave & (ORBX_MAX_LEVELS - 1)];
        const int maxr = min((int)ceilf(y + r), rows - 1), minr = max((int)floorf(y - r), 0);
        const int c = min(max((int)floorf(y), 0), rows - 1);
        en[cnt[c] + atomicAdd(&cur[c], 1)] = make_uint4((unsigned)ir | ((unsigned)q.octave << 16), __float_as_uint(q.x),
                                                        (unsigned)minr | ((unsigned)maxr << 16), 0u);
    }
}

// Median cut of one stereo pair by one 256-thread workgroup.  The SADs are fetched once (device-scope loads: they were written by
// workgroups on other XCDs) and stay in registers for the three passes; the two 256-bin radix steps are resolved by wave 0 in parallel
// (a serial walk of the histogram by one thread was a chain of up to 256 dependent LDS reads).
__device__ __forceinline__ void hist_select(const int *hist, int k, int *sel, int *rem)   // called by wave 0: first bin whose cumulative count exceeds k
{
    const int lane = threadIdx.x;
    const int4 v = reinterpret_cast<const int4 *>(hist)[lane];
    const int s = v.x + v.y + v.z + v.w, inc = wave_incl_scan(s);
    const unsigned long long m = __ballot(inc > k);
    const int L = m ? __builtin_ctzll(m) : 63;
    if (lane == L) {
        int acc = inc - s, b = 0;
        if (acc + v.x > k) b = 0;
        else if (acc + v.x + v.y > k) { b = 1; acc += v.x; }
        else if (acc + v.x + v.y + v.z > k) { b = 2; acc += v.x + v.y; }
        else { b = 3; acc += v.x + v.y + v.z; }
        *sel = 4 * L + b; *rem = k - acc;
    }
}

// Runs either as the consumer of the relaxed publish protocol (the pair's last workgroup inside k_stereo<FOLD>) or as a launch of its own
// (k_stereo_cut); both read and write the three arrays through Published<T> only.
__device__ __forceinline__ void stereo_cut(int n_l, long long o, Published<float> u_right, Published<float> depth, Published<int> st_dist)
{
    __shared__ __align__(16) int hist[256];
    __shared__ int s_sel, s_rem, s_cnt;
    const int tid = threadIdx.x;
    constexpr int K = 8;                        // keypoints per thread held in registers (n_l <= 2048: every ORB-SLAM2 setting)
    int d[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = tid + 256 * k;
        d[k] = i < n_l ? st_dist.get(o + i) : -1;
    }
    hist[tid] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int local = 0;
#pragma unroll
    for (int k = 0; k < K; k++) if (d[k] >= 0) { atomicAdd(&hist[d[k] >> 8], 1); local++; }
    for (int i = tid + 256 * K; i < n_l; i += 256) {
        const int v = st_dist.get(o + i);
        if (v >= 0) { atomicAdd(&hist[v >> 8], 1); local++; }
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    const int nvd = s_cnt;
    if (nvd == 0) return; // reference indexes an empty vector here; nothing to cut
    if (tid < 64) hist_select(hist, nvd / 2, &s_sel, &s_rem);   // vDistIdx[size/2] of the ascending sort
    __syncthreads();
    const int hi = s_sel, rem = s_rem;
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) if (d[k] >= 0 && (d[k] >> 8) == hi) atomicAdd(&hist[d[k] & 255], 1);
    for (int i = tid + 256 * K; i < n_l; i += 256) {
        const int v = st_dist.get(o + i);
        if (v >= 0 && (v >> 8) == hi) atomicAdd(&hist[v & 255], 1);
    }
    __syncthreads();
    if (tid < 64) hist_select(hist, rem, &s_sel, &s_rem);
    __syncthreads();
    const float median = (float)((hi << 8) | s_sel);
    const float th_dist = 1.5f * 1.4f * median;
#pragma unroll
    for (int k = 0; k < K; k++)
        if (d[k] >= 0 && !((float)d[k] < th_dist)) { u_right.put(o + tid + 256 * k, -1.0f); depth.put(o + tid + 256 * k, -1.0f); }
    for (int i = tid + 256 * K; i < n_l; i += 256) {
        const int v = st_dist.get(o + i);
        if (v >= 0 && !((float)v < th_dist)) { u_right.put(o + i, -1.0f); depth.put(o + i, -1.0f); }
    }
}

// FOLD: the median cut runs in the pair's last workgroup (launches of few pairs: one launch less in a single frame's chain).  Batches
// keep it a launch of its own (k_stereo_cut): the write-through stores and the arrival step at the end of EVERY workgroup cost a batch
// more (0.156 against 0.149 ms per 256 pairs) than the extra launch.
#ifndef STX_G
#define STX_G 16
#endif
template <bool FOLD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_stereo(const Geom *__restrict__ g, PyrRef prL, PyrRef prR, int img_l0, int img_r0,
                                                const orbx_keypoint *__restrict__ kL, const uint32_t *__restrict__ dL,
                                                const int *__restrict__ nL, const orbx_keypoint *__restrict__ kR,
                                                const uint32_t *__restrict__ dR, const int *__restrict__ nR, int cap,
                                                float bf, float max_d, StereoTabs tabs, Published<float> u_right,
                                                Published<float> depth, Published<int> st_dist,
                                                const int *__restrict__ row_off, const uint4 *__restrict__ entries, int ent_cap, int reach,
                                                int *__restrict__ arrive, int kpw, int xmap_gx, int xmap_pairs)
{
    // A wave takes FOUR left keypoints.  Coarse stage: one keypoint per 16-lane row (a row of the table holds ~25
    // candidates, so a whole wave per keypoint left most lanes idle and paid the dependent load chain
    // keypoint -> row offsets -> entries -> descriptors once per keypoint); the four chains run side by side and the
    // minimum / its owner are found with row-local DPP butterflies.  Fine stage: the wave then walks its matched
    // keypoints (0..4) one at a time with all 64 lanes on the 11x11x11 SAD.
    constexpr int SW_BYTES = 11 * 16 + 11 * 28 + 4; // per wave: left window rows (16 B) + right window rows (28 B)
    __shared__ __align__(16) uint8_t s_win[4 * ((SW_BYTES + 15) & ~15)];
    // kpw = keypoints per wave: 4 for batches; 1 when the launch has few pairs (a single frame leaves most of the chip idle, and the
    // fine stages of a wave's keypoints run one after the other: with one keypoint per wave they all run side by side)
    // Grid (keypoint groups, pairs), or -- batches -- one dimension in which an XCD owns whole stereo pairs (workgroup L runs on XCD L % 8;
    // STX_G pairs interleaved per XCD): a pair's right-eye descriptors, row table and level windows then cross the fabric into ONE L2
    // instead of into all eight
    int p, bx;
    if (xmap_gx > 0) {
        const unsigned L = blockIdx.x, slot = L >> 3, per = (unsigned)xmap_gx * STX_G, grp = slot / per, rem = slot - grp * per;
        bx = (int)(rem / STX_G); p = (int)((grp * STX_G + rem % STX_G) * 8u + (L & 7u));
        if (p >= xmap_pairs) return;
    } else { p = blockIdx.y; bx = blockIdx.x; }
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const int il_base = (bx * 4 + (threadIdx.x >> 6)) * kpw;
    const int n_l = nL[p];
    if (bx * 4 * kpw >= n_l) return;    // the whole workgroup (the pair's last workgroup is counted among ceil(n_l / (4 kpw)))
    if (il_base < n_l) {                        // wave-uniform
    const int il = il_base + sub;
    const bool have = il < n_l && sub < kpw;
    const long long ol = (long long)p * cap + (have ? il : il_base);
    const uint32_t *dr = dR + (long long)p * cap * 8;
    const orbx_keypoint kp = kL[ol];
    const int level_l = min(max(kp.octave, 0), g->nlevels - 1); // never index the level tables out of range
    const float vl = kp.y, ul = kp.x;
    const int n_rows = g->lv[0].h;
    const int row = (int)vl;
    const float min_u = ul - max_d, max_u = ul; // minD = 0
    const bool active = have && row >= 0 && row < n_rows && !(max_u < 0);
    unsigned mine = 0xFFFFFFFFu;
    float mine_x = 0.f;
    {
        uint32_t a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = dL[ol * 8 + i];
        const int *ro = row_off + (long long)p * (n_rows + 1);
        const uint4 *en = entries + (long long)p * ent_cap;
        // centre rows that can reach this row: one contiguous run of the table.  Only octaves level_l - 1 .. level_l + 1 can match
        // (:644-649), and their bands are at most 2 * scale[level_l + 1] wide: the run is as short as the left keypoint's level allows
        const int rl = (int)((tabs.reach_pk >> (4 * level_l)) & 15ull);
        const int rch = rl == 15 ? reach : min(reach, rl);
        const int b0 = active ? ro[max(row - rch, 0)] : 0, e1 = active ? ro[min(row + rch, n_rows - 1) + 1] : 0;
        for (int base = b0; __any(base < e1); base += 16) { // right keypoints whose row band holds this row (:622)
            const int j = base + sl;
            if (j < e1) {
                const uint4 ent = en[j];
                const int ir = (int)(ent.x & 0xFFFFu), oct = (int)(short)(ent.x >> 16);
                const float qx = __uint_as_float(ent.y);
                const int minr = (int)(ent.z & 0xFFFFu), maxr = (int)(ent.z >> 16);
                if (row >= minr && row <= maxr && oct >= level_l - 1 && oct <= level_l + 1 && qx >= min_u && qx <= max_u) {
                    uint32_t bq[8];
                    const uint4 *src = reinterpret_cast<const uint4 *>(dr + (long long)ir * 8);
                    const uint4 v0 = src[0], v1 = src[1];
                    bq[0] = v0.x; bq[1] = v0.y; bq[2] = v0.z; bq[3] = v0.w;
                    bq[4] = v1.x; bq[5] = v1.y; bq[6] = v1.z; bq[7] = v1.w;
                    const int dist = hamming256(a, bq);
                    if (dist < 100) { // bestDist starts at TH_HIGH, strict <
                        const unsigned key = ((unsigned)dist << 16) | (unsigned)ir;
                        if (key < mine) { mine = key; mine_x = qx; }
                    }
                }
            }
        }
    }
    const unsigned best_row = row16_min_u32(mine);                                   // per 16-lane row
    const unsigned bx_row = row16_or_u32(mine == best_row && mine != 0xFFFFFFFFu ? __float_as_uint(mine_x) : 0u); // the key holds iR: one owner
    for (int gk = 0; gk < kpw; gk++) { // fine stage, one keypoint at a time (all values wave-uniform from here)
        if (il_base + gk >= n_l) break;
        const unsigned best = (unsigned)__builtin_amdgcn_readlane((int)best_row, gk * 16);
        const int best_dist = best == 0xFFFFFFFFu ? 100 : (int)(best >> 16);
        const float ur0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)bx_row, gk * 16));
        const float kx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(kp.x), gk * 16));
        const float ky = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(kp.y), gk * 16));
        const int lvl = __builtin_amdgcn_readlane(level_l, gk * 16);
        const long long og = (long long)p * cap + il_base + gk;
        if (!(best_dist < 75)) { // thOrbDist = (TH_HIGH+TH_LOW)/2: no coarse match
            if (lane == 0) st_store<FOLD>(u_right, depth, st_dist, og, -1.0f, -1.0f, -1);
            continue;
        }
        float out_u = -1.0f, out_z = -1.0f;
        int out_sad = -1;
        const float sfac = tabs.isf[lvl];
        const float sul = roundf(kx * sfac), svl = roundf(ky * sfac), sur0 = roundf(ur0 * sfac);
        const LevelGeom &LV = g->lv[lvl];
        const float iniu = sur0 + 5 - 5, endu = sur0 + 5 + 5 + 1;
        if (!(iniu < 0 || endu >= (float)LV.w)) {
            int pl, prr;
            const uint8_t *imL = orbx_level_ptr(prL, LV, lvl, img_l0 + p, &pl);
            const uint8_t *imR = orbx_level_ptr(prR, LV, lvl, img_r0 + p, &prr);
            const int cy = (int)svl, cxl = (int)sul;
            // each lane owns window pixels e = lane and lane+64 (< 121)
            // (lanes 57..63 have no second pixel: they take the window centre, pixel 60, again -- it contributes |0 - 0| to every shift)
            const int e0 = lane, e1 = lane + 64 < 121 ? lane + 64 : 60;
            const int dy0 = e0 / 11 - 5, dx0 = e0 % 11 - 5;
            const int dy1 = e1 / 11 - 5, dx1 = e1 % 11 - 5;
            const int cx0 = (int)sur0;
            int lc, il0, il1, r0[11], r1[11];   // raw pixels: left centre, the lane's two left pixels, their 11 right partners each
            // Fast path (every window of a keypoint the extractor produced): both windows lie inside the level, so
            // they are fetched as aligned dwords (left 11 rows x 4 dwords = 1 load instruction, right 11 x 7 = 2) into
            // the wave's LDS slice and the 24 per-lane pixels are LDS byte reads.  Gathering them straight from
            // global memory is 25 byte-gather instructions of 64 scattered addresses each per wave, which made the
            // kernel texture-addresser bound.  Windows that touch the border take the reflect-101 byte path.
            const bool inside = cy - 5 >= 0 && cy + 5 < LV.h && cxl - 5 >= 0 && cxl + 5 < LV.w && cx0 - 10 >= 0 && cx0 + 10 < LV.w &&
                                ((((uintptr_t)imL | (uintptr_t)imR | (unsigned)pl | (unsigned)prr) & 3) == 0);
            if (inside) {
                uint8_t *wl = s_win + (threadIdx.x >> 6) * ((SW_BYTES + 15) & ~15), *wr = wl + 11 * 16;
                const int xla = (cxl - 5) & ~3, xra = (cx0 - 10) & ~3;
                // right: 77 dwords (clamped at the row end: the extra bytes of a clamped dword are never read)
                uint32_t vr0, vr1 = 0, vlw = 0;
                {
                    const int rr = lane / 7, cc = lane - rr * 7;
                    const int xo_ = min(xra + 4 * cc, (prr - 4));
                    vr0 = *reinterpret_cast<const uint32_t *>(imR + (long long)(cy - 5 + rr) * prr + xo_);
                    if (lane < 13) {
                        const int i2 = lane + 64, r2 = i2 / 7, c2 = i2 - r2 * 7;
                        vr1 = *reinterpret_cast<const uint32_t *>(imR + (long long)(cy - 5 + r2) * prr + min(xra + 4 * c2, prr - 4));
                    }
                    if (lane < 44) {
                        const int r3 = lane >> 2, c3 = lane & 3;
                        vlw = *reinterpret_cast<const uint32_t *>(imL + (long long)(cy - 5 + r3) * pl + min(xla + 4 * c3, pl - 4));
                    }
                    __builtin_amdgcn_wave_barrier();                      // the previous keypoint's reads are done (in-order LDS)
                    reinterpret_cast<uint32_t *>(wr)[lane] = vr0;        // row pitch 28 bytes = 7 dwords: index = r*7 + c
                    if (lane < 13) reinterpret_cast<uint32_t *>(wr)[lane + 64] = vr1;
                    if (lane < 44) reinterpret_cast<uint32_t *>(wl)[lane] = vlw; // row pitch 16 bytes
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ol_ = cxl - xla, or_ = cx0 - xra; // byte offset of the window centre column in a staged row
                // (volatile: one ds_read_u8 per pixel.  Left to itself the compiler fetches the 11 consecutive bytes as unaligned dwords and
                // spends a shift and a mask per pixel on taking them apart -- 50 vector instructions per keypoint in a kernel that is bound
                // by them, to save LDS instructions it has room for)
                typedef const volatile __attribute__((address_space(3))) uint8_t *lds_bytes;
                const lds_bytes vl_ = (lds_bytes)wl, vr_ = (lds_bytes)wr;
                lc = vl_[5 * 16 + ol_];
                il0 = vl_[(dy0 + 5) * 16 + ol_ + dx0];
                il1 = vl_[(dy1 + 5) * 16 + ol_ + dx1];
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    r0[k] = vr_[(dy0 + 5) * 28 + or_ + dx0 + k - 5];
                    r1[k] = vr_[(dy1 + 5) * 28 + or_ + dx1 + k - 5];
                }
                // (and opaque 32-bit values from here: otherwise the two paths are merged as bytes and every pixel is masked with 0xff after the merge)
                asm volatile("" : "+v"(lc), "+v"(il0), "+v"(il1));
#pragma unroll
                for (int k = 0; k < 11; k++) asm volatile("" : "+v"(r0[k]), "+v"(r1[k]));
            } else {
                lc = lvl_px(imL, pl, LV.w, LV.h, cxl, cy);
                il0 = lvl_px(imL, pl, LV.w, LV.h, cxl + dx0, cy + dy0);
                il1 = lvl_px(imL, pl, LV.w, LV.h, cxl + dx1, cy + dy1);
                // all right-image loads first (independent, in flight together): the 11 shifts of window pixel
                // (dx,dy) are the 11 consecutive columns sur0+dx-5 .. sur0+dx+5 of row cy+dy
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    r0[k] = lvl_px(imR, prr, LV.w, LV.h, cx0 + dx0 + k - 5, cy + dy0);
                    r1[k] = lvl_px(imR, prr, LV.w, LV.h, cx0 + dx1 + k - 5, cy + dy1);
                }
            }
            int dists[11];
            // |(IL - IL_centre) - (IR - IR_centre)| (src/Frame.cc:694-703: both windows minus their centre pixels, L1 norm) is
            // |(IL + IR_centre) - (IR + IL_centre)|: two sums below 511, i.e. u16 -- v_sad_u16 takes the lane's two pixels (low / high half)
            // in ONE instruction.  Lane 60 is the window centre (dy = 0, dx = 0): its r0[k] is the centre pixel of shift k.
            unsigned lpk = (unsigned)il0 | ((unsigned)il1 << 16), lcc = (unsigned)lc * 0x00010001u;
            asm volatile("" : "+v"(lcc));      // (computed once: otherwise it is re-formed as a multiply-add inside each of the 11 shifts)
            auto shift_sad = [&](int k) -> unsigned {
                const unsigned rc = (unsigned)__builtin_amdgcn_readlane(r0[k], 60) * 0x00010001u;
                const unsigned rpk = (unsigned)r0[k] | ((unsigned)r1[k] << 16);
                return __builtin_amdgcn_sad_u16(lpk + rc, rpk + lcc, 0u);
            };
#pragma unroll
            for (int k = 0; k < 11; k += 2) {
                const unsigned sa = shift_sad(k);
                if (k + 1 < 11) { // two shifts per reduction: each total is <= 121*510 < 2^16
                    const unsigned sb = shift_sad(k + 1);
                    const unsigned tot = (unsigned)wave_sum((int)(sa | (sb << 16)));
                    dists[k] = (int)(tot & 0xFFFFu);
                    dists[k + 1] = (int)(tot >> 16);
                } else {
                    dists[k] = wave_sum((int)sa);
                }
            }
            int best_sad = 0x7FFFFFFF, best_inc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++)
                if (dists[k] < best_sad) { best_sad = dists[k]; best_inc = k - 5; }
            if (!(best_inc == -5 || best_inc == 5)) {
                float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int k = 1; k < 10; k++)
                    if (k == best_inc + 5) { d1 = (float)dists[k - 1]; d2 = (float)dists[k]; d3 = (float)dists[k + 1]; }
                const float delta = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                if (!(delta < -1 || delta > 1)) {
                    float best_ur = tabs.sf[lvl] * ((float)sur0 + (float)best_inc + delta);
                    float disparity = kx - best_ur;
                    if (disparity >= 0 && disparity < max_d) {
                        if (disparity <= 0) { disparity = 0.01f; best_ur = (float)((double)kx - 0.01); }
                        out_z = bf / disparity;
                        out_u = best_ur;
                        out_sad = best_sad;
                    }
                }
            }
        }
        if (lane == 0) st_store<FOLD>(u_right, depth, st_dist, og, out_u, out_z, out_sad);
    }
    }
    // ---- median cut (src/Frame.cc:737-750) by the pair's last workgroup: every workgroup's results went out as write-through
    // stores; once they are acknowledged the workgroup counts itself in, and the one that completes the count sees them all
    if (!FOLD) return;
    __shared__ int s_last;
    publish_drain();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nwg = (n_l + 4 * kpw - 1) / (4 * kpw);
        const int old = publish_arrive(&arrive[p]);
        s_last = old == nwg - 1;
        if (old == nwg - 1) __hip_atomic_store(&arrive[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    stereo_cut(n_l, (long long)p * cap, u_right, depth, st_dist);
}

__global__ __launch_bounds__(256) void k_stereo_cut(const int *__restrict__ nL, int cap, Published<float> u_right,
                                                    Published<float> depth, Published<int> st_dist)
{
    stereo_cut(nL[blockIdx.x], (long long)blockIdx.x * cap, u_right, depth, st_dist);
}

static int same_geometry(const orbx_extractor *L, const orbx_extractor *R)
{
    if (L->nlevels != R->nlevels || L->geom.w != R->geom.w || L->geom.h != R->geom.h || L->geom.w == 0) return 0;
    for (int i = 0; i < L->nlevels; i++) if (L->sf[i] != R->sf[i]) return 0;
    return 1;
}

extern "C" int orbx_stereo_row_table_available(const orbx_extractor *R, const void *d_kR, int img_r0, int batch, int cap)
{
    return R && d_kR && img_r0 >= 0 && batch >= 1 && R->d_rt_off && R->rt_kps && cap == R->rt_cap && img_r0 + batch <= R->rt_batch &&
           (const char *)d_kR == (const char *)R->rt_kps + (size_t)img_r0 * cap * sizeof(orbx_keypoint);
}

extern "C" int orbx_stereo_match_batch_device(orbx_extractor *L, int img_l0, orbx_extractor *R, int img_r0, int batch,
                                              const void *d_kL, const void *d_dL, const void *d_nL,
                                              const void *d_kR, const void *d_dR, const void *d_nR, int cap,
                                              float bf, float min_z, void *d_u_right, void *d_depth, int row_table, void *stream)
{
    if (row_table != ORBX_ROWTAB_FROM_KEYPOINTS && row_table != ORBX_ROWTAB_OF_EXTRACTION) { orbx_set_error("row_table: unknown source %d", row_table); return ORBX_E_INVALID; }
    if (!L || !R || !d_kL || !d_dL || !d_nL || !d_kR || !d_dR || !d_nR || !d_u_right || !d_depth || batch < 1 || cap < 1 ||
        img_l0 < 0 || img_r0 < 0 || img_l0 + batch > L->last_batch || img_r0 + batch > R->last_batch || !(min_z > 0)) {
        orbx_set_error("orbx_stereo_match_batch_device: invalid argument");
        return ORBX_E_INVALID;
    }
    if (L->device != R->device || !same_geometry(L, R)) {
        orbx_set_error("left and right extractors differ in device, image size or scale tables");
        return ORBX_E_INVALID;
    }
    if (cap >= 65536) { orbx_set_error("cap must be < 65536"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(L->device));
    hipStream_t s = stream ? (hipStream_t)stream : L->stream;
    const size_t need = (size_t)batch * cap * sizeof(int);
    if (need > L->st_cap || !L->d_st_dist) {
        ORBX_HIP(hipStreamSynchronize(s));
        { const int qrc = orbx_quiesce(L); if (qrc) return qrc; }
        if (L->d_st_dist) ORBX_HIP(hipFree(L->d_st_dist));
        L->d_st_dist = nullptr;
        ORBX_HIP(hipMalloc((void **)&L->d_st_dist, need));
        L->st_cap = need;
    }
    StereoTabs tabs;
    tabs.reach_pk = 0;
    for (int i = 0; i < ORBX_MAX_LEVELS; i++) {
        tabs.sf[i] = L->sf[i]; tabs.isf[i] = L->isf[i];
        // |centre row - row| <= ceil(r) + 1 for a band of radius r (see k_stereo_prep); r = 2 * scale[octave], octave <= i + 1
        const int rl = (int)ceilf(2.0f * L->sf[std::min(i + 1, L->nlevels - 1)]) + 1;
        tabs.reach_pk |= (unsigned long long)std::min(rl, 15) << (4 * i);
    }
    // row table: a right keypoint spans at most 2*r+3 rows, r = 2*scale[top level]
    const int rows = L->geom.lv[0].h;
    const int ent_cap = cap;            // one entry per right keypoint
    const int reach = (int)ceilf(2.0f * L->sf[L->nlevels - 1]) + 1;   // centre rows whose band can hold a given row: |c - row| <= ceil(r) + 1
    void *d_row_off, *d_entries;
    int row_ent_cap = ent_cap;
    // the caller states that the right extractor's last launch wrote exactly these keypoints: its by-product row table serves (desc_rowtab).
    // The source is an argument, never inferred from addresses; the address check below only refuses an impossible claim.
    const bool by_product = row_table == ORBX_ROWTAB_OF_EXTRACTION;
    if (by_product && !orbx_stereo_row_table_available(R, d_kR, img_r0, batch, cap)) {
        orbx_set_error("ORBX_ROWTAB_OF_EXTRACTION: the right handle's last extraction left no row table for this keypoint buffer / capacity / image range");
        return ORBX_E_INVALID;
    }
    if (by_product) {
        d_row_off = R->d_rt_off + (size_t)img_r0 * (rows + 1);
        d_entries = R->d_rt_entries + (size_t)img_r0 * R->rt_ent_cap * sizeof(uint4);
        row_ent_cap = R->rt_ent_cap;
    } else {
        int rc;
        if ((rc = orbx_scratch(L, 7, (size_t)batch * (rows + 1) * sizeof(int), &d_row_off))) return rc;
        const size_t need_e = (size_t)batch * ent_cap * sizeof(uint4);
        if (need_e > L->st_ent_cap || !L->d_st_entries) {
            ORBX_HIP(hipStreamSynchronize(s));
            { const int qrc = orbx_quiesce(L); if (qrc) return qrc; }
            if (L->d_st_entries) ORBX_HIP(hipFree(L->d_st_entries));
            L->d_st_entries = nullptr;
            ORBX_HIP(hipMalloc((void **)&L->d_st_entries, need_e));
            L->st_ent_cap = need_e;
        }
        d_entries = L->d_st_entries;
    }
    PyrRef pl, pr;
    pl.img0 = L->last_img0; pl.img0_stride = (long long)L->last_img_stride; pl.img0_pitch = (int)L->last_pitch;
    pl.pyr = L->d_pyr; pl.pyr_stride = L->geom.pyr_bytes;
    pr.img0 = R->last_img0; pr.img0_stride = (long long)R->last_img_stride; pr.img0_pitch = (int)R->last_pitch;
    pr.pyr = R->d_pyr; pr.pyr_stride = R->geom.pyr_bytes;
    { int urc = orbx_use_stream(L, s); if (urc) return urc; if (R != L && (urc = orbx_use_stream(R, s))) return urc; }
    const float max_d = bf / min_z; // src/Frame.cc:609
    if (!L->d_st_arrive) {       // arrival counters of k_stereo's pairs: zero between launches (the last workgroup of a pair resets its own)
        ORBX_HIP(hipMalloc((void **)&L->d_st_arrive, (size_t)L->max_batch * sizeof(int)));
        ORBX_HIP(hipMemsetAsync(L->d_st_arrive, 0, (size_t)L->max_batch * sizeof(int), s));
    }
    orbx_prof_begin(L, ORBX_STAGE_STEREO, s);
    if (!by_product)
        hipLaunchKernelGGL(k_stereo_prep, dim3(batch), dim3(256), (size_t)2 * ((rows + 4) & ~3) * sizeof(int), s, L->d_geom,
                           (const orbx_keypoint *)d_kR, (const int *)d_nR, cap, tabs, (int *)d_row_off, (uint4 *)d_entries, ent_cap);
    const int kpw_forced = L->stereo_kpw_forced;                          // tests: both forms on the same input (ORBX_STEREO_KPW, read when the handle is created)
    const int kpw = kpw_forced ? kpw_forced : (long long)batch * cap <= 2560 ? 1 : 4;    // a frame or two: one keypoint per wave (see k_stereo; 4 frames 111 vs 115 us, 8 frames 139 vs 154 with four)
    const bool fold = kpw == 1;
    const int gx = (cap + 4 * kpw - 1) / (4 * kpw);
#ifdef STX_NOXMAP
    const bool xmap = false;
#else
    const bool xmap = !fold && batch >= 16 && (long long)gx * ((batch + 8 * STX_G - 1) / (8 * STX_G)) * 8 * STX_G < (1ll << 30);   // batches: an XCD owns whole pairs (see the kernel)
#endif
    const dim3 grid = xmap ? dim3((unsigned)(gx * ((batch + 8 * STX_G - 1) / (8 * STX_G)) * 8 * STX_G)) : dim3(gx, batch);
    hipLaunchKernelGGL((fold ? k_stereo<true> : k_stereo<false>), grid, dim3(256), 0, s, L->d_geom, pl, pr, img_l0, img_r0,
                       (const orbx_keypoint *)d_kL, (const uint32_t *)d_dL, (const int *)d_nL,
                       (const orbx_keypoint *)d_kR, (const uint32_t *)d_dR, (const int *)d_nR, cap, bf, max_d, tabs,
                       Published<float>((float *)d_u_right), Published<float>((float *)d_depth), Published<int>(L->d_st_dist), (const int *)d_row_off, (const uint4 *)d_entries, row_ent_cap, reach,
                       L->d_st_arrive, kpw, xmap ? gx : 0, batch);
    if (!fold) hipLaunchKernelGGL(k_stereo_cut, dim3(batch), dim3(256), 0, s, (const int *)d_nL, cap, Published<float>((float *)d_u_right), Published<float>((float *)d_depth), Published<int>(L->d_st_dist));
    orbx_prof_end(L, s);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbx_stereo_match(orbx_extractor *L, orbx_extractor *R,
                                 const orbx_keypoint *kL, const uint8_t *dL, int nL,
                                 const orbx_keypoint *kR, const uint8_t *dR, int nR,
                                 float bf, float min_z, float *u_right, float *depth)
{
    if (!L || !R || nL < 0 || nR < 0 || (nL && (!kL || !dL || !u_right || !depth)) || (nR && (!kR || !dR))) {
        orbx_set_error("orbx_stereo_match: invalid argument");
        return ORBX_E_INVALID;
    }
    if (nL == 0) return ORBX_OK;
    for (int i = 0; i < nL; i++)
        if (kL[i].octave < 0 || kL[i].octave >= L->nlevels) { orbx_set_error("left keypoint %d: octave %d out of range", i, kL[i].octave); return ORBX_E_INVALID; }
    for (int i = 0; i < nR; i++)
        if (kR[i].octave < 0 || kR[i].octave >= L->nlevels) { orbx_set_error("right keypoint %d: octave %d out of range", i, kR[i].octave); return ORBX_E_INVALID; }
    if (!L->last_img0 || !R->last_img0) { orbx_set_error("run orbx_extract on both eyes first"); return ORBX_E_INVALID; }
    ORBX_HIP(orbx_use_device(L->device));
    ORBX_HIP(hipStreamSynchronize(R->stream)); // the right pyramid was produced on R's stream
    const int cap = nL > nR ? nL : (nR > 0 ? nR : 1);
    void *bkL, *bdL, *bkR, *bdR, *bn, *bu, *bz;
    int rc;
    if ((rc = orbx_scratch(L, 0, sizeof(orbx_keypoint) * cap, &bkL)) || (rc = orbx_scratch(L, 1, 32 * (size_t)cap, &bdL)) ||
        (rc = orbx_scratch(L, 2, sizeof(orbx_keypoint) * cap, &bkR)) || (rc = orbx_scratch(L, 3, 32 * (size_t)cap, &bdR)) ||
        (rc = orbx_scratch(L, 4, 2 * sizeof(int), &bn)) || (rc = orbx_scratch(L, 5, 4 * (size_t)cap, &bu)) ||
        (rc = orbx_scratch(L, 6, 4 * (size_t)cap, &bz)))
        return rc;
    const int counts[2] = { nL, nR };
    hipStream_t s = L->stream;
    ORBX_HIP(hipMemcpyAsync(bkL, kL, sizeof(orbx_keypoint) * nL, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(bdL, dL, 32 * (size_t)nL, hipMemcpyHostToDevice, s));
    if (nR) {
        ORBX_HIP(hipMemcpyAsync(bkR, kR, sizeof(orbx_keypoint) * nR, hipMemcpyHostToDevice, s));
        ORBX_HIP(hipMemcpyAsync(bdR, dR, 32 * (size_t)nR, hipMemcpyHostToDevice, s));
    }
    ORBX_HIP(hipMemcpyAsync(bn, counts, sizeof counts, hipMemcpyHostToDevice, s));
    L->prof_chain = false;
    rc = orbx_stereo_match_batch_device(L, 0, R, 0, 1, bkL, bdL, bn, bkR, bdR, (int *)bn + 1, cap, bf, min_z, bu, bz, ORBX_ROWTAB_FROM_KEYPOINTS, s);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(u_right, bu, 4 * (size_t)nL, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(depth, bz, 4 * (size_t)nL, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}
