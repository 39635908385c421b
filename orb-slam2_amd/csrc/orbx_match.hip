// orbx_match.hip — the per-call ("latency") forms of the three north-star matchers for gfx950:
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&)      reference src/ORBmatcher.cc:171-303
//   ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*)   :568-702
//   ORBmatcher::SearchForTriangulation              :704-871 (+ CheckDistEpipolarLine :147-164, ComputeThreeMaxima :1687-1728)
// called the way ORB-SLAM2 calls them: one keyframe pair per call (Tracking::TrackReferenceKeyFrame), or the current
// keyframe against its 10-20 neighbours / loop candidates (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:241-309;
// LoopClosing::ComputeSim3, src/LoopClosing.cc:293-323) as one batched call.
//
// Decomposition.  A search is independent across shared vocabulary nodes (a feature lives in exactly one node); only the
// rotation histogram is global.  The HOST intersects the two FeatureVectors (it holds both CSRs; a 100 + 100 entry merge)
// and emits one 64-byte work item per shared node; the launch has one single-wave workgroup per item, so a 1000 x 1000
// pair spreads over ~100 CUs instead of one:
//   * SearchForTriangulation (rows independent, vbMatched2 is never written in the reference): the node's rows and
//     columns are staged once into LDS tiles of 64 x 64 (descriptors at a 9-dword pitch, epipolar line per row), lanes
//     enumerate (row, column) pairs, the per-row minimum of dist<<20 | (0xFFFFF - column) is an LDS atomic;
//   * SearchByBoW (greedy inside a node: a second-side feature taken by an earlier row is skipped): lane = column with
//     its descriptor in registers and its "taken" bits in a 64-bit mask (columns lane, lane + 64, ...), rows broadcast
//     from LDS, two DPP wave minima per row.
// The last item of a pair to arrive (device-scope counter) filters the pair by the three dominant rotation bins, counts,
// and (triangulation) compacts the (idx1, idx2) list in order.
//
// Transport (tools/ubench/call_latency.hip, profiles/r03_call_latency.txt): a synchronous call built from
// hipMemcpyAsync + kernel + hipMemcpyAsync + hipStreamSynchronize costs 25 us before any work; the same bytes read by
// the kernel straight from coherent pinned memory and written straight back, with completion signalled through a flag
// the host polls, cost 12 us.  So a call here is ONE launch: the blob (items, pair records, participation bytes and,
// for host-pointer sides, the list-order descriptors) lives in mapped pinned memory, results land in mapped pinned
// memory, no copy commands, no stream synchronisation on the fast path.  Keyframes can also be made resident in HBM
// (orbx_kf_*): descriptors, FeatureVector and keypoint attributes are immutable once a keyframe exists; only the
// map-point flags change, and those travel with each call.
#include "orbx_device.h"
#include <atomic>
#include <chrono>
#include <string.h>
#include <vector>

#define M_TH_LOW 50
#define M_TILE 64                 // rows / columns per LDS tile
#define M_DPITCH 66               // u16 pitch of the distance table: 33 dwords, odd, so a column read by 64 row-lanes is conflict-free
#define M_GREEDY_MAX_COLS 4096    // greedy forms: "taken" bits of a node's columns live in one 64-bit mask per lane
#define M_SPLIT_PAIRS 4096        // triangulation: a node with more (row, column) pairs is split by rows over several items

// One work item = one shared vocabulary node (or a row range of it) of one pair.  Everything a wave needs is in its item, so
// its only dependent fetch from host memory is the item itself (a PCIe read costs ~1.3 us; pair records and participation
// bytes behind a second and third read cost the first version of this kernel 5 us per call and, batched, three times the
// read transactions the link sustains).
struct MItem {
    const uint8_t *a_base, *b_base;        // list-order side blocks
    const uint8_t *a_part, *b_part;        // participation bytes, used when a side of the item has more than 64 entries (NULL: all take part)
    unsigned long long a_mask, b_mask;     // participation of rows a_off + r / columns b_off + c when the count is <= 64
    int a_mp, b_mp;                        // padded list lengths = segment strides inside the blocks
    int a_off, a_cnt, b_off, b_cnt;        // rows / columns of this item in list positions
    int pair, nitems, nslots, tmp_off, out_off, cap, cnt_idx, pad0;   // the pair: arrival target, result row, scratch row, count slot
    int pad1[6];
};
static_assert(sizeof(MItem) == 128, "MItem layout");

struct MTri { float F[9], ex, ey, pad; };  // F12 row-major, epipole of camera 1 in image 2
static_assert(sizeof(MTri) == 48, "MTri layout");

struct MArgs {
    const MItem *items;
    const MTri *tri;                  // [npairs] (triangulation, calls with several pairs)
    Published<int32_t> tmp;           // device scratch rows, -1 outside a call; relaxed publish protocol (orbx_device.h): no plain access compiles
    unsigned *cnt;                    // [0] pairs finalised; pair p: [32 (p + 1)] items arrived, [32 (p + 1) + 1 + bin] rotation histogram; 0 outside a call
    int32_t *out;                     // result block (mapped pinned memory)
    unsigned *flag; unsigned ticket;  // completion flag (mapped pinned memory)
    int npairs, npairs_live;
    float nnratio; int check_ori;
    MTri tri0;                        // the pair of a single-pair call: no fetch
    float te[16], sg[16];             // per octave of the second keyframes: 100 * mvScaleFactors, mvLevelSigma2
};

#ifdef ORBX_DIAG
// diagnostic build: s_memrealtime (100 MHz) stamps of the wave that finalises pair 0: [0] entry, [1] item read,
// [2] last tile staged, [3] node done, [4] arrival atomic returned, [5] histogram known, [6] results issued,
// [7] system fence done, [8] ticket published; tools/diag_match_stamps.py
__device__ unsigned long long g_match_stamp[16];
extern "C" int orbx_diag_match_stamps(unsigned long long *out)
{
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_stamp), sizeof(unsigned long long) * 16));
    return ORBX_OK;
}
#define M_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); m_st[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define M_STAMP_ARG , unsigned long long *m_st
#define M_STAMP_PASS , m_st
#else
#define M_STAMP(i) do { } while (0)
#define M_STAMP_ARG
#define M_STAMP_PASS
#endif

// side block: descriptors [mp][8] u32 | index words [mp] u32 | angles [mp] f32 | positions [mp] float2
// index word: bits 0-19 feature index, 20-27 octave, 31 stereo (u_right >= 0)
__device__ __forceinline__ const uint32_t *side_desc(const uint8_t *b) { return reinterpret_cast<const uint32_t *>(b); }
__device__ __forceinline__ const uint32_t *side_idx(const uint8_t *b, int mp) { return reinterpret_cast<const uint32_t *>(b + 32ll * mp); }
__device__ __forceinline__ const float *side_ang(const uint8_t *b, int mp) { return reinterpret_cast<const float *>(b + 36ll * mp); }
__device__ __forceinline__ const float2 *side_xy(const uint8_t *b, int mp) { return reinterpret_cast<const float2 *>(b + 40ll * mp); }

// rotation bin of src/ORBmatcher.cc:253-258 (factor = 1/30 with HISTO_LENGTH = 30: upstream quirk kept)
__device__ __forceinline__ int m_rot_bin(float a1, float a2)
{
    const float factor = 1.0f / 30;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == 30) bin = 0;
    return (unsigned)bin < 30 ? bin : 0;
}

__device__ __forceinline__ void m_stage_desc(const uint32_t *src, long long p, uint32_t *dst)
{
    const uint4 d0 = reinterpret_cast<const uint4 *>(src + 8 * p)[0], d1 = reinterpret_cast<const uint4 *>(src + 8 * p)[1];
    dst[0] = d0.x; dst[1] = d0.y; dst[2] = d0.z; dst[3] = d0.w; dst[4] = d1.x; dst[5] = d1.y; dst[6] = d1.z; dst[7] = d1.w;
}

// does list position `off + k` (k = position inside the item's rows / columns) take part?
__device__ __forceinline__ bool m_live(int cnt, unsigned long long mask, const uint8_t *part, int off, int k)
{
    if (cnt <= 64) return (mask >> k) & 1ull;
    return part ? part[off + k] != 0 : true;
}

// A match found by an item wave goes into the pair's scratch row together with its rotation bin.  The store is an agent-scope
// atomic store (write-through, sc1): the arrival protocol below then needs no L2 write-back on the producer side -- a
// __threadfence() per item wave (buffer_wbl2) cost 17-27 us of a 2000-item launch, profiles/r03_match_stamps.txt.
__device__ __forceinline__ void m_record(const MArgs &g, const MItem &it, int slot, int value, float ang1, float ang2)
{
    const int bin = m_rot_bin(ang1, ang2);
    g.tmp.put(it.tmp_off + slot, (bin << 20) | value);
}

// The last item of a pair: ComputeThreeMaxima (:1687-1728) on the histogram the items accumulated, clearing of the other bins
// (:282-300), the count, and for SearchForTriangulation the ordered (idx1, idx2) list (:863-868).  One wave; the scratch row
// goes back to -1 and the counters to 0.
template <int MODE>
__device__ void m_finalize(const MArgs &g, const MItem &it, int lane, unsigned *s_hist M_STAMP_ARG)
{
    const int nslots = it.nslots, cap = it.cap;
    const Published<int32_t> tmp = g.tmp.at(it.tmp_off);
    int32_t *out = g.out + it.out_off;
    int v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int s = u * 64 + lane; v[u] = s < nslots ? tmp.get(s) : -1; }
    int k1 = -1, k2 = -1, k3 = -1;
    if (g.check_ori) {
        s_hist[lane & 31] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u++) if (v[u] != -1) atomicAdd(&s_hist[v[u] >> 20], 1u);
        for (int base = 16 * 64; base < nslots; base += 64) { const int s = base + lane; const int w = s < nslots ? tmp.get(s) : -1; if (w != -1) atomicAdd(&s_hist[w >> 20], 1u); }
        __syncthreads();
        // the sequential scan of the reference keeps, with strict comparisons, the three largest bins in the order (count
        // descending, bin ascending) among bins with count > 0: three wave maxima of count << 8 | (255 - bin)
        const unsigned hv = lane < 30 ? s_hist[lane] : 0u;
        unsigned key = hv ? (hv << 8) | (255u - (unsigned)lane) : 0u;
        const unsigned m1 = wave_max_u32(key);
        if (m1) k1 = 255 - (int)(m1 & 255u);
        if (lane == k1) key = 0;
        const unsigned m2 = wave_max_u32(key);
        if (m2) k2 = 255 - (int)(m2 & 255u);
        if (lane == k2) key = 0;
        const unsigned m3 = wave_max_u32(key);
        if (m3) k3 = 255 - (int)(m3 & 255u);
        const int max1 = (int)(m1 >> 8), max2 = (int)(m2 >> 8), max3 = (int)(m3 >> 8);
        if ((float)max2 < 0.1f * (float)max1) { k2 = -1; k3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { k3 = -1; }
    }
    M_STAMP(5);
    int cnt_keep = 0, run = 0;
    for (int base = 0; base < nslots; base += 16 * 64) {
        if (base) {
#pragma unroll
            for (int u = 0; u < 16; u++) { const int s = base + u * 64 + lane; v[u] = s < nslots ? tmp.get(s) : -1; }
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int s = base + u * 64 + lane;
            if (base + u * 64 >= nslots) continue;                     // wave-uniform
            const int bin = v[u] >> 20;
            const bool keep = v[u] != -1 && (!g.check_ori || bin == k1 || bin == k2 || bin == k3);
            if (MODE == 2) {
                const unsigned long long bal = __ballot(keep);
                const int pos = run + __popcll(bal & ((1ull << lane) - 1ull));
                if (keep && pos < cap) { out[2 * pos] = s; out[2 * pos + 1] = v[u] & 0xFFFFF; }
                run += __popcll(bal);
            } else {
                if (s < nslots) out[s] = keep ? (v[u] & 0xFFFFF) : -1;
                cnt_keep += keep ? 1 : 0;
            }
            if (v[u] != -1) tmp.put(s, -1);
        }
    }
    if (MODE != 2) run = wave_sum(cnt_keep);
    if (lane == 0) g.out[it.cnt_idx] = run;
    M_STAMP(6);
    __threadfence_system();
    M_STAMP(7);
    if (lane == 0) {
        bool last = true;
        if (g.npairs_live > 1) {
            last = atomicAdd(g.cnt, 1u) == (unsigned)g.npairs_live - 1;
            if (last) { g.cnt[0] = 0; __threadfence_system(); }
        }
        if (last) *reinterpret_cast<volatile unsigned *>(g.flag) = g.ticket;
    }
#ifdef ORBX_DIAG
    M_STAMP(8);
    if (lane == 0 && it.pair == 0) for (int i = 0; i < 9; i++) g_match_stamp[i] = m_st[i];
#endif
}

// MODE 0: SearchByBoW(KF, F)  — result row indexed by the F (second-side) feature, value = KF feature
// MODE 1: SearchByBoW(KF, KF) — result row indexed by the KF1 (first-side) feature, value = KF2 feature
// MODE 2: SearchForTriangulation — intermediate row indexed by the KF1 feature, value = KF2 feature
template <int MODE>
__device__ __forceinline__ void m_body(const MArgs &g, const MItem &it, const int lane M_STAMP_ARG)
{
    __shared__ uint32_t s_rd[M_TILE * 9], s_cd[M_TILE * 9];          // row / column descriptors, 9-dword pitch
    __shared__ float s_rl[MODE == 2 ? M_TILE * 4 : 4];               // epipolar line of a row: a, b, c, a^2 + b^2
    __shared__ uint32_t s_ri[M_TILE], s_ci[M_TILE], s_key[M_TILE];   // index words (bit 30: takes no part); per-row best key / per-column owner
    __shared__ float s_ca[M_TILE];
    __shared__ float2 s_cxy[MODE == 2 ? M_TILE : 1];
    __shared__ float s_te[16], s_sg[16];
    __shared__ uint16_t s_D[MODE == 2 ? 2 : M_TILE * M_DPITCH];      // distance table of a node (greedy forms)
    __shared__ unsigned s_hist[32];
    const uint32_t *a_desc = side_desc(it.a_base), *a_idx = side_idx(it.a_base, it.a_mp);
    const float *a_ang = side_ang(it.a_base, it.a_mp);
    const uint32_t *b_desc = side_desc(it.b_base), *b_idx = side_idx(it.b_base, it.b_mp);
    const float *b_ang = side_ang(it.b_base, it.b_mp);
    const int ac = it.a_cnt, bc = it.b_cnt;

    if (MODE == 2) {
        const float2 *a_xy = side_xy(it.a_base, it.a_mp), *b_xy = side_xy(it.b_base, it.b_mp);
        float F[9], ex, ey;
        if (g.npairs == 1) {
#pragma unroll
            for (int i = 0; i < 9; i++) F[i] = g.tri0.F[i];
            ex = g.tri0.ex; ey = g.tri0.ey;
        } else {
            const MTri *T = g.tri + it.pair;
#pragma unroll
            for (int i = 0; i < 9; i++) F[i] = T->F[i];
            ex = T->ex; ey = T->ey;
        }
        if (lane < 16) { s_te[lane] = g.te[lane]; s_sg[lane] = g.sg[lane]; }
        for (int r0 = 0; r0 < ac; r0 += M_TILE) {
            const int nr = min(M_TILE, ac - r0);
            float ang1 = 0.f;
            uint32_t iw1 = 0x40000000u;                               // bit 30: the row takes no part
            if (lane < nr && m_live(ac, it.a_mask, it.a_part, it.a_off, r0 + lane)) {
                // stage the row: descriptor, index word, epipolar line of the keypoint in image 2 (:150-152)
                const int p = it.a_off + r0 + lane;
                m_stage_desc(a_desc, p, s_rd + lane * 9);
                iw1 = a_idx[p];
                const float2 xy = a_xy[p];
                ang1 = a_ang[p];
                const float la = xy.x * F[0] + xy.y * F[3] + F[6];
                const float lb = xy.x * F[1] + xy.y * F[4] + F[7];
                const float lc = xy.x * F[2] + xy.y * F[5] + F[8];
                s_rl[lane * 4 + 0] = la; s_rl[lane * 4 + 1] = lb; s_rl[lane * 4 + 2] = lc; s_rl[lane * 4 + 3] = la * la + lb * lb;
            }
            s_ri[lane] = iw1;
            s_key[lane] = 0xFFFFFFFFu;
            for (int c0 = 0; c0 < bc; c0 += M_TILE) {
                const int nc = min(M_TILE, bc - c0);
                __syncthreads();                      // the previous tile's readers are done
                {
                    uint32_t w = 0x40000000u;
                    if (lane < nc && m_live(bc, it.b_mask, it.b_part, it.b_off, c0 + lane)) {
                        const int p = it.b_off + c0 + lane;
                        m_stage_desc(b_desc, p, s_cd + lane * 9);
                        w = b_idx[p];
                        s_cxy[lane] = b_xy[p];
                    }
                    s_ci[lane] = w;
                }
                __syncthreads();
                M_STAMP(2);
                const int np = nr * nc;
                const FastDiv dv(nc);
                for (int p = lane; p < np; p += 64) {
                    const int i = dv.div(p), j = p - i * nc;
                    const uint32_t wi = s_ri[i], wj = s_ci[j];
                    if ((wi | wj) & 0x40000000u) continue;
                    int dist = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) dist += __popc(s_rd[i * 9 + k] ^ s_cd[j * 9 + k]);
                    if (dist > M_TH_LOW) continue;
                    const float2 xy2 = s_cxy[j];
                    const int oct2 = (int)((wj >> 20) & 15u);
                    if (!(wi >> 31) && !(wj >> 31)) {                 // both monocular: not too close to the epipole (:789-796)
                        const float distex = ex - xy2.x, distey = ey - xy2.y;
                        if (distex * distex + distey * distey < s_te[oct2]) continue;
                    }
                    const float num = s_rl[i * 4 + 0] * xy2.x + s_rl[i * 4 + 1] * xy2.y + s_rl[i * 4 + 2];
                    const float den = s_rl[i * 4 + 3];
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (!((double)dsqr < 3.84 * (double)s_sg[oct2])) continue;
                    // smallest distance, LAST column on ties (dist <= bestDist replaces, :786-800)
                    atomicMin(&s_key[i], ((unsigned)dist << 20) | (0xFFFFFu - (unsigned)(c0 + j)));
                }
            }
            __syncthreads();
            if (lane < nr) {
                const unsigned key = s_key[lane];
                if (key != 0xFFFFFFFFu) {
                    const int pb = it.b_off + (int)(0xFFFFFu - (key & 0xFFFFFu));
                    const uint32_t w2 = b_idx[pb];
                    const float ang2 = b_ang[pb];
                    m_record(g, it, (int)(iw1 & 0xFFFFFu), (int)(w2 & 0xFFFFFu), ang1, ang2);
                }
            }
            __syncthreads();
        }
    } else if (ac <= M_TILE && bc <= M_TILE) {
        // ---- a node that fits one tile (every node of a 1000-feature frame over ORB-SLAM2's 100 vocabulary nodes): all distances
        // at once, lane = (row, column) pair, into an LDS table; then the reference's sequential walk over the rows (a column
        // taken by an earlier row is skipped) as a fixpoint with lane = row: choice(r) = best column not chosen by an earlier
        // row, runner-up over the same columns, iterated until no row changes (row r only depends on rows before it, so
        // the iteration reproduces the walk: a round per link of the longest chain of displaced rows, 2-3 rounds on real data).
        float ang1 = 0.f;
        uint32_t iw1 = 0x40000000u;
        if (lane < ac && m_live(ac, it.a_mask, it.a_part, it.a_off, lane)) {
            const int p = it.a_off + lane;
            m_stage_desc(a_desc, p, s_rd + lane * 9);
            iw1 = a_idx[p]; ang1 = a_ang[p];
        }
        s_ri[lane] = iw1;
        {
            uint32_t w = 0x40000000u;
            if (lane < bc && m_live(bc, it.b_mask, it.b_part, it.b_off, lane)) {
                const int p = it.b_off + lane;
                m_stage_desc(b_desc, p, s_cd + lane * 9);
                w = b_idx[p]; s_ca[lane] = b_ang[p];
            }
            s_ci[lane] = w;
        }
        __syncthreads();
        M_STAMP(2);
        const int np = ac * bc;
        const FastDiv dv(bc);
        for (int p = lane; p < np; p += 64) {
            const int i = dv.div(p), j = p - i * bc;
            unsigned d = 0xFFFFu;                                     // the column takes no part
            if (!(s_ci[j] & 0x40000000u)) {
                d = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) d += __popc(s_rd[i * 9 + k] ^ s_cd[j * 9 + k]);
            }
            s_D[i * M_DPITCH + j] = (uint16_t)d;
        }
        __syncthreads();
        // lane = row from here on: its table row goes to registers as keys dist << 8 | column (0xFFFFFFFF: the column takes no part)
        const bool row_on = !(iw1 & 0x40000000u);
        unsigned kcol[M_TILE];
#pragma unroll
        for (int gq = 0; gq < M_TILE / 8; gq++) {
            if (gq * 8 < bc) {                                         // wave-uniform
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = gq * 8 + q;
                    const unsigned d = j < bc ? s_D[lane * M_DPITCH + j] : 0xFFFFu;
                    kcol[j] = d == 0xFFFFu ? 0xFFFFFFFFu : (d << 8) | (unsigned)j;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) kcol[gq * 8 + q] = 0xFFFFFFFFu;
            }
        }
        int choice = -1;
        for (int round = 0; round <= ac; round++) {
            // columns held by EARLIER rows: exclusive prefix OR over the lanes of 1 << choice
            unsigned c_lo = choice >= 0 && choice < 32 ? 1u << choice : 0u, c_hi = choice >= 32 ? 1u << (choice - 32) : 0u;
            // (one lane up on the DPP path, wave_shr:1 with lane 0 reading 0: __shfl_up is an LDS permute round trip, twice per round on the
            // chain every single-pair call waits for)
            c_lo = (unsigned)ORBX_DPP((int)c_lo, 0, 0x138, 0xf, 0xf); c_hi = (unsigned)ORBX_DPP((int)c_hi, 0, 0x138, 0xf, 0xf);
            const unsigned t_lo = wave_incl_scan_or(c_lo), t_hi = bc > 32 ? wave_incl_scan_or(c_hi) : 0u;
            unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;               // smallest and second smallest key among the free columns
#pragma unroll
            for (int gq = 0; gq < M_TILE / 8; gq++) {
                if (gq * 8 < bc) {                                     // wave-uniform
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int j = gq * 8 + q;
                        const unsigned tw = j < 32 ? t_lo : t_hi;
                        const unsigned k = (tw & (1u << (j & 31))) ? 0xFFFFFFFFu : kcol[j];
                        const unsigned lo = min(k1, k), hi = max(k1, k);
                        k1 = lo; k2 = min(k2, hi);
                    }
                }
            }
            int nc = -1;
            if (row_on && k1 != 0xFFFFFFFFu) {
                // first column wins ties (strict <, :229-239): the key carries the column below the distance
                const int best1 = (int)(k1 >> 8), best2 = k2 == 0xFFFFFFFFu ? 256 : (int)(k2 >> 8);
                const bool ok_dist = MODE == 0 ? best1 <= M_TH_LOW : best1 < M_TH_LOW;
                if (ok_dist && (float)best1 < g.nnratio * (float)best2) nc = (int)(k1 & 0xFFu);
            }
            const bool changed = nc != choice;
            choice = nc;
            if (!__any(changed)) break;
        }
        if (choice >= 0) {
            const int idx1 = (int)(iw1 & 0xFFFFFu), idx2 = (int)(s_ci[choice] & 0xFFFFFu);
            if (MODE == 0) m_record(g, it, idx2, idx1, ang1, s_ca[choice]);
            else m_record(g, it, idx1, idx2, ang1, s_ca[choice]);
        }
    } else {
        // ---- larger nodes: lane = column; column chunk 0 in registers; "taken" bit t of a lane = column lane + 64 t (columns that
        // take no part start taken); rows broadcast from LDS, two DPP wave minima per row
        unsigned long long taken = 0;
        uint32_t db[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        uint32_t bw0 = 0; float bang0 = 0.f;
        for (int t = 0, j = lane; j < bc; j += 64, t++) {
            if (!m_live(bc, it.b_mask, it.b_part, it.b_off, j)) taken |= 1ull << t;
            else if (t == 0) {
                const int p = it.b_off + j;
                m_stage_desc(b_desc, p, db);
                bw0 = b_idx[p]; bang0 = b_ang[p];
            }
        }
        if (lane >= bc) taken |= 1ull;
        for (int r0 = 0; r0 < ac; r0 += M_TILE) {
            const int nr = min(M_TILE, ac - r0);
            __syncthreads();
            {
                uint32_t w = 0x40000000u;
                if (lane < nr && m_live(ac, it.a_mask, it.a_part, it.a_off, r0 + lane)) {
                    const int p = it.a_off + r0 + lane;
                    m_stage_desc(a_desc, p, s_rd + lane * 9);
                    w = a_idx[p];
                    s_ca[lane] = a_ang[p];
                }
                s_ri[lane] = w;
            }
            __syncthreads();
            M_STAMP(2);
            for (int i = 0; i < nr; i++) {
                const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ri[i]);
                if (wi & 0x40000000u) continue;                      // no (good) map point (:205-210 / :606-613)
                uint32_t da[8];
#pragma unroll
                for (int k = 0; k < 8; k++) da[k] = s_rd[i * 9 + k];
                unsigned key1 = 0xFFFFFFFFu;                         // dist << 20 | column: first column wins ties (strict <, :229-239)
                int l1 = 256, l2 = 256;
                if (!(taken & 1ull)) { l1 = hamming256(da, db); key1 = ((unsigned)l1 << 20) | (unsigned)lane; }
                for (int t = 1, j = lane + 64; j < bc; j += 64, t++) {
                    if ((taken >> t) & 1ull) continue;
                    uint32_t dj[8];
                    m_stage_desc(b_desc, it.b_off + j, dj);
                    const int dist = hamming256(da, dj);
                    if (dist < l1) { l2 = l1; l1 = dist; key1 = ((unsigned)dist << 20) | (unsigned)j; }
                    else if (dist < l2) l2 = dist;
                }
                const unsigned kbest = wave_min_u32(key1);
                if (kbest == 0xFFFFFFFFu) continue;
                const int best1 = (int)(kbest >> 20);
                // second smallest of the multiset: the owner of the winner contributes its own runner-up
                const int best2 = (int)wave_min_u32((unsigned)(key1 == kbest ? l2 : l1));
                const bool ok_dist = MODE == 0 ? best1 <= M_TH_LOW : best1 < M_TH_LOW;
                if (ok_dist && (float)best1 < g.nnratio * (float)best2) {
                    const int jw = (int)(kbest & 0xFFFFFu);
                    if (lane == (jw & 63)) {
                        taken |= 1ull << (jw >> 6);
                        uint32_t w2 = bw0; float ang2 = bang0;
                        if (jw >= 64) { w2 = b_idx[it.b_off + jw]; ang2 = b_ang[it.b_off + jw]; }
                        const int idx1 = (int)(wi & 0xFFFFFu), idx2 = (int)(w2 & 0xFFFFFu);
                        if (MODE == 0) m_record(g, it, idx2, idx1, s_ca[i], ang2);
                        else m_record(g, it, idx1, idx2, s_ca[i], ang2);
                    }
                }
            }
        }
    }
    // ---- arrival: the last item of the pair finalises it
    M_STAMP(3);
    if (it.nitems > 1) {
        // release: every store of this wave to the scratch row was a write-through agent-scope atomic; once they are acknowledged
        // (vmcnt 0) they are visible device-wide, so the counter may be bumped without an L2 write-back
        publish_drain();
        unsigned prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(&g.cnt[32 * (it.pair + 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
        if (prev != (unsigned)it.nitems - 1) return;
        if (lane == 0) g.cnt[32 * (it.pair + 1)] = 0;
    }
    __threadfence();              // acquire: the other items' stores (other XCDs, other L2s) before the scratch row is read
    M_STAMP(4);
    m_finalize<MODE>(g, it, lane, s_hist M_STAMP_PASS);
}

template <int MODE>
__global__ __launch_bounds__(64) void k_match(MArgs g)
{
    const int lane = threadIdx.x;
#ifdef ORBX_DIAG
    unsigned long long m_st[9];
#endif
    M_STAMP(0);
    const MItem it = g.items[blockIdx.x];
    M_STAMP(1);
    m_body<MODE>(g, it, lane M_STAMP_PASS);
}

// A call with ONE pair (what ORB-SLAM2's own loops issue) and at most M_BYVAL_MAX items carries its work items BY VALUE in the
// kernel-argument segment: what every item of a pair shares once (MPairV), what differs per item in 24 bytes.  The wave's item then
// arrives with the argument fetch it makes anyway, instead of behind a second dependent read of host memory (~1.3 us over PCIe).
struct MItemC { unsigned long long a_mask, b_mask; uint16_t a_off, a_cnt, b_off, b_cnt; };
static_assert(sizeof(MItemC) == 24, "MItemC layout");
#define M_BYVAL_MAX 144
struct MItemsV { MItemC it[M_BYVAL_MAX]; };
struct MPairV {
    const uint8_t *a_base, *b_base, *a_part, *b_part;
    int a_mp, b_mp, nitems, nslots, cap, cnt_idx;
};
static_assert(sizeof(MArgs) + sizeof(MPairV) + sizeof(MItemsV) <= 4096, "kernel-argument segment");

template <int MODE>
__global__ __launch_bounds__(64) void k_match_v(MArgs g, const MPairV pv, const MItemsV iv)
{
    const int lane = threadIdx.x;
#ifdef ORBX_DIAG
    unsigned long long m_st[9];
#endif
    M_STAMP(0);
    const MItemC c = iv.it[blockIdx.x];
    MItem it;
    it.a_base = pv.a_base; it.b_base = pv.b_base; it.a_part = pv.a_part; it.b_part = pv.b_part;
    it.a_mask = c.a_mask; it.b_mask = c.b_mask;
    it.a_mp = pv.a_mp; it.b_mp = pv.b_mp;
    it.a_off = c.a_off; it.a_cnt = c.a_cnt; it.b_off = c.b_off; it.b_cnt = c.b_cnt;
    it.pair = 0; it.nitems = pv.nitems; it.nslots = pv.nslots; it.tmp_off = 0; it.out_off = 0; it.cap = pv.cap; it.cnt_idx = pv.cnt_idx; it.pad0 = 0;
    M_STAMP(1);
    m_body<MODE>(g, it, lane M_STAMP_PASS);
}

// ---------------------------------------------------------------- host side

struct MatchCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t *h_blob = nullptr, *d_blob = nullptr; size_t blob_cap = 0;      // coherent mapped pinned memory: host view / device view
    int32_t *h_out = nullptr, *d_out = nullptr; size_t out_cap = 0;         // result block, likewise
    unsigned *h_flag = nullptr, *d_flag = nullptr;
    int32_t *d_tmp = nullptr; size_t tmp_cap = 0;
    unsigned *d_cnt = nullptr; size_t cnt_cap = 0;                          // 32 counters per pair + 32 global (see MArgs::cnt)
    unsigned ticket = 0;
    std::vector<int> pair_first;                                            // first item of each pair (+ end)
    std::vector<size_t> pair_out;                                           // first int of each pair's results in the result block
    std::vector<unsigned long long> bits;                                   // participation bits of the call's sides, list order
    bool poisoned = false;      // a call failed after its launch: d_tmp / d_cnt may not be back at -1 / 0 (the kernels rely on that); restored before the next call
    void release()              // stream, pinned blobs and device scratch of this thread on this device
    {
        if (device < 0) return;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && device < ndev && hipSetDevice(device) == hipSuccess) {   // (a runtime that is already shutting down: leave it alone)
            if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
            if (h_flag) (void)hipHostFree(h_flag);
            if (h_blob) (void)hipHostFree(h_blob);
            if (h_out) (void)hipHostFree(h_out);
            if (d_tmp) (void)hipFree(d_tmp);
            if (d_cnt) (void)hipFree(d_cnt);
        }
        stream = nullptr; h_flag = d_flag = nullptr; h_blob = d_blob = nullptr; h_out = d_out = nullptr; d_tmp = nullptr; d_cnt = nullptr;
        blob_cap = out_cap = tmp_cap = cnt_cap = 0; device = -1; poisoned = false;
    }
    ~MatchCtx() { release(); }  // a thread that called a matcher gives its resources back when it ends (thread pools, short-lived workers)
};
static thread_local MatchCtx g_mctx[16];
// the calling thread's matcher resources on every device, now (they are also released when the thread ends)
extern "C" void orbx_thread_release(void) { for (MatchCtx &c : g_mctx) c.release(); }
// test hook: 1 = a single-pair call reads its work items from the blob like a batch does (the k_match kernels), 0 = by value (k_match_v)
static std::atomic<int> g_items_in_memory{0};
extern "C" int orbx_debug_set_match_items(int in_memory) { g_items_in_memory.store(in_memory ? 1 : 0, std::memory_order_relaxed); return ORBX_OK; }
static inline bool orbx_match_items_in_memory() { return g_items_in_memory.load(std::memory_order_relaxed) != 0; }
static thread_local double g_mtime[4];   // host phases of this thread's most recent call, microseconds: prepare, launch, wait, copy-out
static inline double m_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
extern "C" int orbx_debug_match_timing(double *out4) { if (!out4) return ORBX_E_INVALID; for (int i = 0; i < 4; i++) out4[i] = g_mtime[i]; return ORBX_OK; }

static int mctx_get(int device, MatchCtx **out)
{
    MatchCtx *c = device >= 0 && device < 16 ? &g_mctx[device] : nullptr;
    if (c && c->stream) {                 // steady state: this thread has used the device before
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) ORBX_HIP(hipSetDevice(device));
        if (c->poisoned) {                // an earlier call failed after its launch: scratch rows back to -1, counters to 0, before anything reads them
            ORBX_HIP(hipStreamSynchronize(c->stream));
            if (c->d_tmp) ORBX_HIP(hipMemset(c->d_tmp, 0xFF, c->tmp_cap * sizeof(int32_t)));
            if (c->d_cnt) ORBX_HIP(hipMemset(c->d_cnt, 0, c->cnt_cap * sizeof(unsigned)));
            ORBX_HIP(hipDeviceSynchronize());
            *c->h_flag = 0; c->ticket = 0;
            c->poisoned = false;
        }
        *out = c;
        return ORBX_OK;
    }
    int ndev = 0;
    if (!c || hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    // the context counts as set up (c->stream != 0 is the steady-state test above) only when every piece exists
    unsigned *hf = nullptr, *df = nullptr;
    hipStream_t st = nullptr;
    ORBX_HIP(hipHostMalloc((void **)&hf, 64, hipHostMallocCoherent | hipHostMallocMapped));
    if (hipHostGetDevicePointer((void **)&df, hf, 0) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipHostFree(hf);
        orbx_set_error("matcher context: stream / mapped flag could not be created on device %d", device);
        return ORBX_E_HIP;
    }
    *hf = 0;
    c->h_flag = hf; c->d_flag = df; c->device = device; c->stream = st;
    *out = c;
    return ORBX_OK;
}

static int mctx_reserve(MatchCtx *c, size_t blob, size_t out_ints, size_t tmp_ints, size_t npairs)
{
    if (blob > c->blob_cap) {
        ORBX_HIP(hipStreamSynchronize(c->stream));
        if (c->h_blob) ORBX_HIP(hipHostFree(c->h_blob));
        c->h_blob = nullptr; c->blob_cap = 0;
        ORBX_HIP(hipHostMalloc((void **)&c->h_blob, blob * 2, hipHostMallocCoherent | hipHostMallocMapped));   // non-coherent pinned memory: no faster
        ORBX_HIP(hipHostGetDevicePointer((void **)&c->d_blob, c->h_blob, 0));
        c->blob_cap = blob * 2;
    }
    if (out_ints > c->out_cap) {
        ORBX_HIP(hipStreamSynchronize(c->stream));
        if (c->h_out) ORBX_HIP(hipHostFree(c->h_out));
        c->h_out = nullptr; c->out_cap = 0;
        ORBX_HIP(hipHostMalloc((void **)&c->h_out, out_ints * 2 * sizeof(int32_t), hipHostMallocCoherent | hipHostMallocMapped));
        ORBX_HIP(hipHostGetDevicePointer((void **)&c->d_out, c->h_out, 0));
        c->out_cap = out_ints * 2;
    }
    if (tmp_ints > c->tmp_cap) {
        ORBX_HIP(hipStreamSynchronize(c->stream));
        if (c->d_tmp) ORBX_HIP(hipFree(c->d_tmp));
        c->d_tmp = nullptr; c->tmp_cap = 0;
        ORBX_HIP(hipMalloc((void **)&c->d_tmp, tmp_ints * 2 * sizeof(int32_t)));
        ORBX_HIP(hipMemset(c->d_tmp, 0xFF, tmp_ints * 2 * sizeof(int32_t)));   // -1: every call leaves the rows it used at -1 again
        c->tmp_cap = tmp_ints * 2;
    }
    if (32 * (npairs + 1) > c->cnt_cap) {
        ORBX_HIP(hipStreamSynchronize(c->stream));
        if (c->d_cnt) ORBX_HIP(hipFree(c->d_cnt));
        c->d_cnt = nullptr; c->cnt_cap = 0;
        ORBX_HIP(hipMalloc((void **)&c->d_cnt, 64 * (npairs + 1) * sizeof(unsigned)));
        ORBX_HIP(hipMemset(c->d_cnt, 0, 64 * (npairs + 1) * sizeof(unsigned)));
        c->cnt_cap = 64 * (npairs + 1);
    }
    return ORBX_OK;
}

static inline size_t m_a16(size_t v) { return (v + 15) & ~(size_t)15; }
static inline int m_pad4(int m) { return (m + 3) & ~3; }

// one side of a call as the host sees it
struct HSide {
    int n = 0, nnodes = 0, m = 0, mp = 0;
    const uint32_t *node_id = nullptr; const int32_t *node_off = nullptr; const uint32_t *feat = nullptr;
    const uint8_t *flag = nullptr;        // caller's per-feature flags (NULL: none given)
    const float *u_right = nullptr;       // per feature (host-pointer sides)
    const uint8_t *stereo_l = nullptr;    // per list position (resident sides)
    const orbx_featset *pack = nullptr;   // host-pointer side: packed into the blob by the call
    const uint8_t *dev_base = nullptr;    // resident side: its block in HBM
    int role = 0;                         // 0 first side, 1 second side
    size_t blob_off = 0, part_off = 0; bool has_part = false;
    size_t bits_off = 0;                  // first word of its participation bits in MatchCtx::bits
};

// list-order block of a feature set (layout: see side_desc .. side_xy)
static void side_pack(const orbx_featset *f, bool geom, uint8_t *dst, int m, int mp)
{
    uint32_t *sd = reinterpret_cast<uint32_t *>(dst);
    uint32_t *si = reinterpret_cast<uint32_t *>(dst + 32 * (size_t)mp);
    float *sa = reinterpret_cast<float *>(dst + 36 * (size_t)mp);
    float *sxy = reinterpret_cast<float *>(dst + 40 * (size_t)mp);
    for (int k = 0; k < m; k++) {
        const uint32_t fi = f->feat[k];
        memcpy(sd + 8 * (size_t)k, f->desc + 32 * (size_t)fi, 32);
        uint32_t w = fi;
        if (geom) {
            w |= ((uint32_t)f->octave[fi] & 0xFFu) << 20;
            if (f->u_right[fi] >= 0) w |= 0x80000000u;
            sxy[2 * k] = f->x[fi]; sxy[2 * k + 1] = f->y[fi];
        }
        si[k] = w;
        sa[k] = f->angle[fi];
    }
}
static size_t side_bytes(int mp, bool geom) { return (size_t)mp * (geom ? 48 : 40); }

// device-resident keyframe / frame
struct orbx_kf {
    int device, n, nnodes, m, mp, geom, max_octave;
    uint8_t *d_block;
    std::vector<uint32_t> node_id, feat;
    std::vector<int32_t> node_off;
    std::vector<uint8_t> stereo_l;
};

extern "C" int orbx_kf_create(int device, const orbx_featset *fs, orbx_kf **out)
{
    if (!out || !fs) { orbx_set_error("orbx_kf_create: null argument"); return ORBX_E_INVALID; }
    *out = nullptr;
    const bool geom = fs->x && fs->y && fs->octave && fs->u_right;
    if (!orbx_feat_validate(fs, geom ? 1 : 0, 0)) { orbx_set_error("orbx_kf_create: malformed feature set"); return ORBX_E_INVALID; }
    MatchCtx *c;
    int rc = mctx_get(device, &c);
    if (rc) return rc;
    orbx_kf *k = new orbx_kf();
    k->device = device; k->n = fs->n; k->nnodes = fs->nnodes; k->geom = geom ? 1 : 0;
    k->m = fs->nnodes ? fs->node_off[fs->nnodes] : 0;
    k->mp = m_pad4(k->m);
    k->d_block = nullptr;
    k->max_octave = 0;
    if (geom) for (int i = 0; i < fs->n; i++) {
        if (fs->octave[i] < 0 || fs->octave[i] >= ORBX_MAX_LEVELS) { delete k; orbx_set_error("orbx_kf_create: octave out of range"); return ORBX_E_INVALID; }
        if (fs->octave[i] > k->max_octave) k->max_octave = fs->octave[i];
    }
    k->node_id.assign(fs->node_id, fs->node_id + fs->nnodes);
    if (fs->nnodes) k->node_off.assign(fs->node_off, fs->node_off + fs->nnodes + 1); else k->node_off.assign(1, 0);
    k->feat.assign(fs->feat, fs->feat + k->m);
    k->stereo_l.resize((size_t)k->m);
    for (int i = 0; i < k->m; i++) k->stereo_l[i] = geom && fs->u_right[fs->feat[i]] >= 0 ? 1 : 0;
    const size_t bytes = side_bytes(k->mp, true);
    if (bytes) {
        std::vector<uint8_t> h(bytes, 0);
        side_pack(fs, geom, h.data(), k->m, k->mp);
        if (hipMalloc((void **)&k->d_block, bytes) != hipSuccess || hipMemcpy(k->d_block, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
            if (k->d_block) hipFree(k->d_block);
            delete k;
            orbx_set_error("orbx_kf_create: HIP allocation / upload failed");
            return ORBX_E_HIP;
        }
    }
    *out = k;
    return ORBX_OK;
}

extern "C" void orbx_kf_destroy(orbx_kf *k)
{
    if (!k) return;
    hipSetDevice(k->device);
    if (k->d_block) hipFree(k->d_block);
    delete k;
}

extern "C" int orbx_kf_size(const orbx_kf *k) { return k ? k->n : ORBX_E_INVALID; }

static void hside_from_set(HSide *s, const orbx_featset *f, int role)
{
    s->n = f->n; s->nnodes = f->nnodes; s->m = f->nnodes ? f->node_off[f->nnodes] : 0; s->mp = m_pad4(s->m);
    s->node_id = f->node_id; s->node_off = f->node_off; s->feat = f->feat;
    s->flag = f->flag; s->u_right = f->u_right; s->pack = f; s->role = role;
}
static void hside_from_kf(HSide *s, const orbx_kf *k, const uint8_t *flag, int role)
{
    s->n = k->n; s->nnodes = k->nnodes; s->m = k->m; s->mp = k->mp;
    s->node_id = k->node_id.data(); s->node_off = k->node_off.data(); s->feat = k->feat.data();
    s->flag = flag; s->stereo_l = k->stereo_l.data(); s->dev_base = k->d_block; s->role = role;
}

// Return codes of match_call beyond ORBX_*: 1 = not applicable (a node with more than M_GREEDY_MAX_COLS columns in a greedy
// search): the caller takes the legacy kernels.
#define M_NOT_APPLICABLE 1

// Which list positions of a side take part in a search of `mode` (one bit each, formed in match_call).  SearchByBoW: first side = features
// with a (good) map point (:205-210), (KF, KF) both sides (:606-613, :627-631); triangulation: features WITHOUT a map point, only stereo
// ones under bOnlyStereo (:750-763, :776-789)

// rows[p * row_stride ..]: MODE 0 / 1 the match row (nslots ints), MODE 2 the (idx1, idx2) list (2 * cap ints); counts[p]
static int match_call(int mode, int device, HSide *sides, int nsides, const int *pa, const int *pb, int npairs,
                      const float *F12s, const float *eps, const float *sf2, const float *sig2, int nlv,
                      float nnratio, int check_ori, int only_stereo, int cap,
                      int32_t *rows, size_t row_stride, int *counts)
{
    const double t_begin = m_now_us();
    double t_prep = t_begin, t_launch = t_begin, t_wait = t_begin;
    MatchCtx *c;
    int rc = mctx_get(device, &c);
    if (rc) return rc;
    {   // who takes part, once per side and call: one bit per list position (a side shared by 20 pairs is walked once).  Word at a time,
        // without branches in the walk: this loop and the item loop below are most of what a batched call costs on the host
        size_t words = 0;
        for (int s = 0; s < nsides; s++) {
            HSide &S = sides[s];
            S.has_part = mode == 2 ? (S.flag != nullptr || only_stereo) : (S.role == 0 || mode == 1);
            S.bits_off = words;
            if (S.has_part) words += ((size_t)S.m >> 6) + 2;
        }
        if (c->bits.size() < words) c->bits.resize(words);
        for (int s = 0; s < nsides; s++) {
            const HSide &S = sides[s];
            if (!S.has_part) continue;
            unsigned long long *bw = c->bits.data() + S.bits_off;
            const uint32_t *feat = S.feat;
            const uint8_t *flag = S.flag, *st_l = S.stereo_l;
            const float *ur = S.u_right;
            const int m = S.m, nw = (m >> 6) + 2;
            for (int w = 0; w < nw; w++) {
                const int k0 = w << 6, n = m - k0 < 64 ? (m - k0 > 0 ? m - k0 : 0) : 64;
                unsigned long long v = 0;
                if (mode != 2) {                       // SearchByBoW: features with a (good) map point
                    if (flag) for (int j = 0; j < n; j++) v |= (unsigned long long)(flag[feat[k0 + j]] != 0) << j;
                } else {                               // triangulation: features WITHOUT a map point, only stereo ones under bOnlyStereo
                    if (flag) for (int j = 0; j < n; j++) v |= (unsigned long long)(flag[feat[k0 + j]] == 0) << j;
                    else v = n == 64 ? ~0ull : (1ull << n) - 1ull;
                    if (only_stereo) {
                        unsigned long long sv = 0;
                        if (st_l) for (int j = 0; j < n; j++) sv |= (unsigned long long)(st_l[k0 + j] != 0) << j;
                        else for (int j = 0; j < n; j++) sv |= (unsigned long long)(ur[feat[k0 + j]] >= 0) << j;
                        v &= sv;
                    }
                }
                bw[w] = v;
            }
        }
    }
    // the (at most 64) participation bits of list positions [off, off + cnt) of a side
    auto mask_of = [&](const HSide &S, int off, int cnt) -> unsigned long long {
        if (!S.has_part) return ~0ull;
        const unsigned long long *bw = c->bits.data() + S.bits_off;
        const int w = off >> 6, sh = off & 63;
        unsigned long long v = bw[w] >> sh;
        if (sh) v |= bw[w + 1] << (64 - sh);
        return cnt >= 64 ? v : v & ((1ull << cnt) - 1ull);
    };
    // ---- everything whose size is known before the node intersection: result rows, scratch rows and the blob
    //      [per-pair F12 / epipole | participation bytes | host-pointer sides | work items], so that the items can be written where the
    //      kernel reads them, final (no staging vector, no pointer pass, no copy)
    const bool geom = mode == 2;
    if (nsides >= 65536) { orbx_set_error("matcher call too large"); return ORBX_E_INVALID; }
    c->pair_first.assign((size_t)npairs + 1, 0);
    c->pair_out.resize((size_t)npairs);
    size_t tmp_total = 0, out_total = 0, item_bound = 0;
    for (int p = 0; p < npairs; p++) {
        const HSide &A = sides[pa[p]], &B = sides[pb[p]];
        const int nslots = mode == 0 ? B.n : A.n;
        c->pair_out[(size_t)p] = out_total;
        tmp_total += (size_t)((nslots + 3) & ~3);
        out_total += mode == 2 ? 2 * (size_t)cap : (size_t)nslots;
        // items of a pair: one per shared node, plus (triangulation) the row splits of nodes above M_SPLIT_PAIRS (row, column) pairs:
        // ceil(a / floor(S / b)) <= 2 a b / S + 1 per node, and never more than one item per row
        const size_t split = 2 * (size_t)A.m * (size_t)B.m / M_SPLIT_PAIRS;
        item_bound += (size_t)(A.nnodes < B.nnodes ? A.nnodes : B.nnodes) + (mode == 2 ? (split < (size_t)A.m ? split : (size_t)A.m) : 0);
    }
    const size_t cnt_base = out_total;
    out_total += (size_t)npairs;
    if (tmp_total >= (1ull << 31) || out_total >= (1ull << 31) || item_bound >= (1ull << 24)) { orbx_set_error("matcher call too large"); return ORBX_E_INVALID; }
    size_t off = 0;
    const size_t off_tri = off;
    if (mode == 2 && npairs > 1) off += m_a16((size_t)npairs * sizeof(MTri));
    for (int s = 0; s < nsides; s++) {
        HSide &S = sides[s];
        S.part_off = 0;
        if (S.has_part) { S.part_off = off; off += m_a16((size_t)S.mp + 1); }
    }
    for (int s = 0; s < nsides; s++)
        if (sides[s].pack) { sides[s].blob_off = off; off += m_a16(side_bytes(sides[s].mp, geom)); }
    off = (off + 127) & ~(size_t)127;
    const size_t off_items = off;
    off += item_bound * sizeof(MItem);
    if ((rc = mctx_reserve(c, off, out_total, tmp_total, (size_t)npairs))) return rc;
    // ---- work items: the merge join of the two FeatureVectors (same node set as :193-302 / :588-673 / :737-838)
    MItem *const items = reinterpret_cast<MItem *>(c->h_blob + off_items);
    size_t n_items = 0, tmp_off = 0;
    int live = 0;
    bool need_part_bytes = false;
    for (int p = 0; p < npairs; p++) {
        const HSide &A = sides[pa[p]], &B = sides[pb[p]];
        const int nslots = mode == 0 ? B.n : A.n;
        const size_t first = n_items;
        c->pair_first[(size_t)p] = (int)first;
        const uint8_t *a_base = A.pack ? c->d_blob + A.blob_off : A.dev_base, *b_base = B.pack ? c->d_blob + B.blob_off : B.dev_base;
        const uint8_t *a_part = A.has_part ? c->d_blob + A.part_off : nullptr, *b_part = B.has_part ? c->d_blob + B.part_off : nullptr;
        int ia = 0, ib = 0;
        while (ia < A.nnodes && ib < B.nnodes) {
            const uint32_t na = A.node_id[ia], nb = B.node_id[ib];
            if (na < nb) { ia++; continue; }
            if (nb < na) { ib++; continue; }
            const int ao = A.node_off[ia], acn = A.node_off[ia + 1] - ao, bo = B.node_off[ib], bcn = B.node_off[ib + 1] - bo;
            ia++; ib++;
            if (acn <= 0 || bcn <= 0) continue;
            if (mode != 2 && bcn > M_GREEDY_MAX_COLS) return M_NOT_APPLICABLE;
            unsigned long long bmask = ~0ull;
            if (bcn <= 64 && B.has_part) {
                bmask = mask_of(B, bo, bcn);
                if (!bmask) continue;                                  // no column takes part: nothing can match in this node
            }
            if (bcn > 64 && B.has_part) need_part_bytes = true;
            int rows_per = acn;
            if (mode == 2 && (long long)acn * bcn > M_SPLIT_PAIRS) { rows_per = M_SPLIT_PAIRS / bcn; if (rows_per < 1) rows_per = 1; }
            for (int r = 0; r < acn; r += rows_per) {
                const int a_off = ao + r, a_cnt = acn - r < rows_per ? acn - r : rows_per;
                unsigned long long amask = ~0ull;
                if (a_cnt <= 64 && A.has_part) {
                    amask = mask_of(A, a_off, a_cnt);
                    if (!amask) continue;                              // no row takes part
                }
                if (a_cnt > 64 && A.has_part) need_part_bytes = true;
                if (n_items >= item_bound) { orbx_set_error("matcher call: work item bound exceeded"); return ORBX_E_INVALID; }   // (cannot happen: see the bound)
                MItem &it = items[n_items++];
                it.a_base = a_base; it.b_base = b_base; it.a_part = a_part; it.b_part = b_part;
                it.a_mask = amask; it.b_mask = bmask;
                it.a_mp = A.mp; it.b_mp = B.mp;
                it.a_off = a_off; it.a_cnt = a_cnt; it.b_off = bo; it.b_cnt = bcn;
                it.pair = p; it.nslots = nslots; it.tmp_off = (int)tmp_off; it.out_off = (int)c->pair_out[(size_t)p]; it.cap = cap;
                it.cnt_idx = (int)(cnt_base + (size_t)p); it.pad0 = 0;
            }
        }
        const int nitems = (int)(n_items - first);
        for (size_t i = first; i < n_items; i++) items[i].nitems = nitems;
        if (nitems) { tmp_off += (size_t)((nslots + 3) & ~3); live++; }
    }
    c->pair_first[(size_t)npairs] = (int)n_items;
    for (int s = 0; s < nsides; s++) {
        HSide &S = sides[s];
        if (S.has_part && need_part_bytes) {        // (the kernel looks at these bytes only for a side of an item with more than 64 entries)
            uint8_t *pt = c->h_blob + S.part_off;
            const unsigned long long *bw = c->bits.data() + S.bits_off;
            for (int k = 0; k < S.m; k++) pt[k] = (uint8_t)((bw[k >> 6] >> (k & 63)) & 1ull);
        }
        if (S.pack) side_pack(S.pack, geom, c->h_blob + S.blob_off, S.m, S.mp);
    }
    if (n_items) {
        bool by_value = npairs == 1 && n_items <= M_BYVAL_MAX && !orbx_match_items_in_memory();
        if (by_value) for (size_t i = 0; i < n_items; i++) { const MItem &it = items[i]; by_value = by_value && it.a_off < 65536 && it.a_cnt < 65536 && it.b_off < 65536 && it.b_cnt < 65536; }
        MArgs g;
        memset(&g, 0, sizeof g);
        g.items = reinterpret_cast<const MItem *>(c->d_blob + off_items);
        g.tri = reinterpret_cast<const MTri *>(c->d_blob + off_tri);
        g.tmp = Published<int32_t>(c->d_tmp); g.cnt = c->d_cnt;
        g.out = c->d_out; g.flag = c->d_flag; g.ticket = ++c->ticket; g.npairs = npairs; g.npairs_live = live;
        g.nnratio = nnratio; g.check_ori = check_ori;
        if (mode == 2) {
            MTri *ht = reinterpret_cast<MTri *>(c->h_blob + off_tri);
            for (int p = 0; p < npairs; p++) {
                MTri t;
                for (int i = 0; i < 9; i++) t.F[i] = F12s[9 * (size_t)p + i];
                t.ex = eps[2 * (size_t)p]; t.ey = eps[2 * (size_t)p + 1]; t.pad = 0.f;
                if (npairs > 1) ht[p] = t; else g.tri0 = t;
            }
            for (int i = 0; i < 16; i++) { g.te[i] = i < nlv ? 100 * sf2[i] : 0.f; g.sg[i] = i < nlv ? sig2[i] : 0.f; }
        }
        t_prep = m_now_us();
        c->poisoned = true;         // until this call has its results: any error return below leaves scratch rows / counters in an unknown state
        const dim3 grid((unsigned)n_items), block(64);
        if (by_value) {
            const MItem &f = items[0];    // one pair: tmp_off = out_off = 0, everything but masks and ranges is the same in every item
            MPairV pv;
            pv.a_base = f.a_base; pv.b_base = f.b_base; pv.a_part = f.a_part; pv.b_part = f.b_part;
            pv.a_mp = f.a_mp; pv.b_mp = f.b_mp; pv.nitems = f.nitems; pv.nslots = f.nslots; pv.cap = f.cap; pv.cnt_idx = f.cnt_idx;
            MItemsV iv;
            for (size_t i = 0; i < n_items; i++) {
                const MItem &it = items[i];
                iv.it[i].a_mask = it.a_mask; iv.it[i].b_mask = it.b_mask;
                iv.it[i].a_off = (uint16_t)it.a_off; iv.it[i].a_cnt = (uint16_t)it.a_cnt; iv.it[i].b_off = (uint16_t)it.b_off; iv.it[i].b_cnt = (uint16_t)it.b_cnt;
            }
            g.items = nullptr;
            if (mode == 0) hipLaunchKernelGGL(k_match_v<0>, grid, block, 0, c->stream, g, pv, iv);
            else if (mode == 1) hipLaunchKernelGGL(k_match_v<1>, grid, block, 0, c->stream, g, pv, iv);
            else hipLaunchKernelGGL(k_match_v<2>, grid, block, 0, c->stream, g, pv, iv);
        }
        else if (mode == 0) hipLaunchKernelGGL(k_match<0>, grid, block, 0, c->stream, g);
        else if (mode == 1) hipLaunchKernelGGL(k_match<1>, grid, block, 0, c->stream, g);
        else hipLaunchKernelGGL(k_match<2>, grid, block, 0, c->stream, g);
        ORBX_HIP(hipGetLastError());
        t_launch = m_now_us();
        // completion: the kernel's last finaliser publishes the ticket after a system-scope fence; poll it (hipStreamSynchronize
        // costs 5 us more per call, profiles/r03_call_latency.txt).  A stream that finishes without the ticket is an error.
        const volatile unsigned *flag = c->h_flag;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != g.ticket) {
            if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                ORBX_HIP(hipStreamSynchronize(c->stream));
                if (*flag != g.ticket) { orbx_set_error("matcher kernel finished without publishing its results"); return ORBX_E_HIP; }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        c->poisoned = false;        // the last finaliser has run: every scratch row is -1 again, every counter 0
        t_wait = m_now_us();
    }
    for (int p = 0; p < npairs; p++) {
        int32_t *row = rows + (size_t)p * row_stride;
        const int first = c->pair_first[(size_t)p];
        if (first == c->pair_first[(size_t)p + 1]) {
            if (mode != 2) { const int nslots = mode == 0 ? sides[pb[p]].n : sides[pa[p]].n; for (int i = 0; i < nslots; i++) row[i] = -1; }
            counts[p] = 0;
            continue;
        }
        const int cnt = c->h_out[cnt_base + (size_t)p];
        counts[p] = cnt;
        if (mode == 2) memcpy(row, c->h_out + c->pair_out[(size_t)p], sizeof(int32_t) * 2 * (size_t)(cnt < cap ? cnt : cap));
        else memcpy(row, c->h_out + c->pair_out[(size_t)p], sizeof(int32_t) * (size_t)(mode == 0 ? sides[pb[p]].n : sides[pa[p]].n));
    }
    g_mtime[0] = t_prep - t_begin; g_mtime[1] = t_launch - t_prep; g_mtime[2] = t_wait - t_launch; g_mtime[3] = m_now_us() - t_wait;
    return ORBX_OK;
}

// ---------------------------------------------------------------- entry points (host pointers)

static int bow_entry(int mode, int device, const orbx_featset *firsts, int nfirst, const orbx_featset *seconds, int nsecond,
                     float nnratio, int check_ori, int32_t *match, int *nmatches, const char *who)
{
    // nfirst x 1 (keyframes against one frame) or 1 x nsecond (one keyframe against candidates)
    if (!firsts || !seconds || nfirst < 1 || nsecond < 1 || (nfirst > 1 && nsecond > 1) || !match || !nmatches) {
        orbx_set_error("%s: invalid argument", who);
        return ORBX_E_INVALID;
    }
    for (int i = 0; i < nfirst; i++) if (!orbx_feat_validate(&firsts[i], 0, 1)) { orbx_set_error("%s: malformed feature set %d", who, i); return ORBX_E_INVALID; }
    for (int i = 0; i < nsecond; i++) if (!orbx_feat_validate(&seconds[i], 0, 1)) { orbx_set_error("%s: malformed feature set", who); return ORBX_E_INVALID; }
    const int npairs = nfirst * nsecond;
    size_t stride = 0;
    if (mode == 0) stride = (size_t)seconds[0].n;
    else for (int i = 0; i < nfirst; i++) if ((size_t)firsts[i].n > stride) stride = (size_t)firsts[i].n;
    const int forced = orbx_bow_forced_form();
    if (!forced) {
        std::vector<HSide> sides((size_t)(nfirst + nsecond));
        std::vector<int> pa((size_t)npairs), pb((size_t)npairs);
        for (int i = 0; i < nfirst; i++) hside_from_set(&sides[(size_t)i], &firsts[i], 0);
        for (int i = 0; i < nsecond; i++) hside_from_set(&sides[(size_t)(nfirst + i)], &seconds[i], 1);
        for (int p = 0; p < npairs; p++) { pa[(size_t)p] = nfirst > 1 ? p : 0; pb[(size_t)p] = nfirst + (nsecond > 1 ? p : 0); }
        const int rc = match_call(mode, device, sides.data(), nfirst + nsecond, pa.data(), pb.data(), npairs, nullptr, nullptr, nullptr, nullptr, 0,
                                  nnratio, check_ori, 0, 0, match, stride, nmatches);
        if (rc != M_NOT_APPLICABLE) return rc;
    }
    // legacy kernels (orbx_bow.hip): forced by the test hook, or a vocabulary node too large for the register form
    if (nsecond == 1) return orbx_bow_run_legacy(mode, device, firsts, nfirst, &seconds[0], nnratio, check_ori, match, nmatches);
    for (int i = 0; i < nsecond; i++) {
        const int rc = orbx_bow_run_legacy(mode, device, firsts, 1, &seconds[i], nnratio, check_ori, match + (size_t)i * stride, nmatches + i);
        if (rc) return rc;
    }
    return ORBX_OK;
}

extern "C" int orbx_search_by_bow_kf_f(int device, const orbx_featset *kf, const orbx_featset *f,
                                       float nnratio, int check_orientation, int32_t *match_f, int *nmatches)
{
    return bow_entry(0, device, kf, 1, f, 1, nnratio, check_orientation, match_f, nmatches, "orbx_search_by_bow_kf_f");
}

extern "C" int orbx_search_by_bow_kf_f_batch(int device, const orbx_featset *kfs, int nkf, const orbx_featset *f,
                                             float nnratio, int check_orientation, int32_t *match_f, int *nmatches)
{
    return bow_entry(0, device, kfs, nkf, f, 1, nnratio, check_orientation, match_f, nmatches, "orbx_search_by_bow_kf_f_batch");
}

extern "C" int orbx_search_by_bow_kf_kf(int device, const orbx_featset *k1, const orbx_featset *k2,
                                        float nnratio, int check_orientation, int32_t *match12, int *nmatches)
{
    return bow_entry(1, device, k1, 1, k2, 1, nnratio, check_orientation, match12, nmatches, "orbx_search_by_bow_kf_kf");
}

extern "C" int orbx_search_by_bow_kf_kf_batch(int device, const orbx_featset *k1, const orbx_featset *k2s, int n2,
                                              float nnratio, int check_orientation, int32_t *match12, int *nmatches)
{
    return bow_entry(1, device, k1, 1, k2s, n2, nnratio, check_orientation, match12, nmatches, "orbx_search_by_bow_kf_kf_batch");
}

static int tri_check_tables(const float *sf2, const float *sig2, int nlevels2, const char *who)
{
    if (!sf2 || !sig2 || nlevels2 < 1 || nlevels2 > ORBX_MAX_LEVELS) { orbx_set_error("%s: invalid scale tables", who); return 0; }
    return 1;
}

extern "C" int orbx_search_for_triangulation_batch(int device, const orbx_featset *k1, const orbx_featset *k2s, int n2,
                                                   const float *F12s, const float *epipoles,
                                                   const float *sf2, const float *sig2, int nlevels2,
                                                   int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs)
{
    const char *who = "orbx_search_for_triangulation";
    if (!k1 || !k2s || n2 < 1 || !F12s || !epipoles || !pairs || !npairs || cap < 0) { orbx_set_error("%s: invalid argument", who); return ORBX_E_INVALID; }
    if (!tri_check_tables(sf2, sig2, nlevels2, who)) return ORBX_E_INVALID;
    if (!orbx_feat_validate(k1, 1, 1)) { orbx_set_error("%s: malformed feature set", who); return ORBX_E_INVALID; }
    for (int i = 0; i < n2; i++) {
        if (!orbx_feat_validate(&k2s[i], 1, 1)) { orbx_set_error("%s: malformed feature set %d", who, i); return ORBX_E_INVALID; }
        for (int j = 0; j < k2s[i].n; j++)
            if (k2s[i].octave[j] < 0 || k2s[i].octave[j] >= nlevels2) { orbx_set_error("octave out of range"); return ORBX_E_INVALID; }
    }
    std::vector<HSide> sides((size_t)n2 + 1);
    std::vector<int> pa((size_t)n2, 0), pb((size_t)n2);
    hside_from_set(&sides[0], k1, 0);
    for (int i = 0; i < n2; i++) { hside_from_set(&sides[(size_t)i + 1], &k2s[i], 1); pb[(size_t)i] = i + 1; }
    const int rc = match_call(2, device, sides.data(), n2 + 1, pa.data(), pb.data(), n2, F12s, epipoles, sf2, sig2, nlevels2, 0.f, check_orientation,
                              only_stereo, cap, pairs, 2 * (size_t)cap, npairs);
    if (rc) return rc;
    for (int i = 0; i < n2; i++)
        if (npairs[i] > cap) { orbx_set_error("pair capacity %d < %d matches", cap, npairs[i]); return ORBX_E_CAPACITY; }
    return ORBX_OK;
}

extern "C" int orbx_search_for_triangulation(int device, const orbx_featset *k1, const orbx_featset *k2,
                                             const float F12[9], float ex, float ey,
                                             const float *sf2, const float *sig2, int nlevels2,
                                             int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs)
{
    const float ep[2] = { ex, ey };
    return orbx_search_for_triangulation_batch(device, k1, k2, 1, F12, ep, sf2, sig2, nlevels2, only_stereo, check_orientation, pairs, cap, npairs);
}

// ---------------------------------------------------------------- entry points (resident keyframes)

static int kf_sides(const orbx_kf *first, const uint8_t *flag1, const orbx_kf *const *seconds, const uint8_t *const *flags2, int n2,
                    std::vector<HSide> &sides, std::vector<int> &pa, std::vector<int> &pb, const char *who)
{
    if (!first || !seconds || n2 < 1) { orbx_set_error("%s: invalid argument", who); return ORBX_E_INVALID; }
    sides.assign((size_t)n2 + 1, HSide());
    pa.assign((size_t)n2, 0); pb.resize((size_t)n2);
    hside_from_kf(&sides[0], first, flag1, 0);
    for (int i = 0; i < n2; i++) {
        if (!seconds[i] || seconds[i]->device != first->device) { orbx_set_error("%s: keyframe %d is null or on another device", who, i); return ORBX_E_INVALID; }
        hside_from_kf(&sides[(size_t)i + 1], seconds[i], flags2 ? flags2[i] : nullptr, 1);
        pb[(size_t)i] = i + 1;
    }
    return ORBX_OK;
}

extern "C" int orbx_kf_search_by_bow_kf_f(const orbx_kf *kf, const uint8_t *kf_flag, const orbx_kf *f,
                                          float nnratio, int check_orientation, int32_t *match_f, int *nmatches)
{
    const char *who = "orbx_kf_search_by_bow_kf_f";
    if (!kf_flag || !match_f || !nmatches) { orbx_set_error("%s: null argument", who); return ORBX_E_INVALID; }
    std::vector<HSide> sides; std::vector<int> pa, pb;
    int rc = kf_sides(kf, kf_flag, &f, nullptr, 1, sides, pa, pb, who);
    if (rc) return rc;
    rc = match_call(0, kf->device, sides.data(), 2, pa.data(), pb.data(), 1, nullptr, nullptr, nullptr, nullptr, 0, nnratio, check_orientation, 0, 0,
                    match_f, (size_t)f->n, nmatches);
    if (rc == M_NOT_APPLICABLE) { orbx_set_error("%s: a vocabulary node holds more than %d features", who, M_GREEDY_MAX_COLS); return ORBX_E_INVALID; }
    return rc;
}

// Tracking::Relocalization (src/Tracking.cc:1661-1682): SearchByBoW(pKF, mCurrentFrame) for every candidate keyframe, one launch.  The keyframes are
// resident; the frame lives one frame time and comes as host pointers (packed into the call's blob and read by the kernel over PCIe: an upload of
// its own would cost more than the whole call)
extern "C" int orbx_kf_search_by_bow_kfs_f(const orbx_kf *const *kfs, const uint8_t *const *kf_flags, int nkf, const orbx_featset *f,
                                           float nnratio, int check_orientation, int32_t *match_f, int *nmatches)
{
    const char *who = "orbx_kf_search_by_bow_kfs_f";
    if (!kfs || !kf_flags || nkf < 1 || !f || !match_f || !nmatches) { orbx_set_error("%s: invalid argument", who); return ORBX_E_INVALID; }
    if (!orbx_feat_validate(f, 0, 1)) { orbx_set_error("%s: malformed frame feature set", who); return ORBX_E_INVALID; }
    std::vector<HSide> sides((size_t)nkf + 1);
    std::vector<int> pa((size_t)nkf), pb((size_t)nkf, nkf);
    for (int i = 0; i < nkf; i++) {
        if (!kfs[i] || !kf_flags[i] || kfs[i]->device != kfs[0]->device) { orbx_set_error("%s: keyframe %d is null, has no flags or is on another device", who, i); return ORBX_E_INVALID; }
        hside_from_kf(&sides[(size_t)i], kfs[i], kf_flags[i], 0);
        pa[(size_t)i] = i;
    }
    hside_from_set(&sides[(size_t)nkf], f, 1);
    const int rc = match_call(0, kfs[0]->device, sides.data(), nkf + 1, pa.data(), pb.data(), nkf, nullptr, nullptr, nullptr, nullptr, 0, nnratio, check_orientation, 0, 0,
                              match_f, (size_t)f->n, nmatches);
    if (rc == M_NOT_APPLICABLE) { orbx_set_error("%s: a vocabulary node holds more than %d features", who, M_GREEDY_MAX_COLS); return ORBX_E_INVALID; }
    return rc;
}

extern "C" int orbx_kf_search_by_bow_kf_kf(const orbx_kf *k1, const uint8_t *flag1, const orbx_kf *const *k2s, const uint8_t *const *flags2, int n2,
                                           float nnratio, int check_orientation, int32_t *match12, int *nmatches)
{
    const char *who = "orbx_kf_search_by_bow_kf_kf";
    if (!flag1 || !flags2 || !match12 || !nmatches) { orbx_set_error("%s: null argument", who); return ORBX_E_INVALID; }
    for (int i = 0; i < n2; i++) if (!flags2[i]) { orbx_set_error("%s: null flags %d", who, i); return ORBX_E_INVALID; }
    std::vector<HSide> sides; std::vector<int> pa, pb;
    int rc = kf_sides(k1, flag1, k2s, flags2, n2, sides, pa, pb, who);
    if (rc) return rc;
    rc = match_call(1, k1->device, sides.data(), n2 + 1, pa.data(), pb.data(), n2, nullptr, nullptr, nullptr, nullptr, 0, nnratio, check_orientation, 0, 0,
                    match12, (size_t)k1->n, nmatches);
    if (rc == M_NOT_APPLICABLE) { orbx_set_error("%s: a vocabulary node holds more than %d features", who, M_GREEDY_MAX_COLS); return ORBX_E_INVALID; }
    return rc;
}

extern "C" int orbx_kf_search_for_triangulation(const orbx_kf *k1, const uint8_t *flag1, const orbx_kf *const *k2s, const uint8_t *const *flags2, int n2,
                                                const float *F12s, const float *epipoles, const float *sf2, const float *sig2, int nlevels2,
                                                int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs)
{
    const char *who = "orbx_kf_search_for_triangulation";
    if (!F12s || !epipoles || !pairs || !npairs || cap < 0) { orbx_set_error("%s: invalid argument", who); return ORBX_E_INVALID; }
    if (!tri_check_tables(sf2, sig2, nlevels2, who)) return ORBX_E_INVALID;
    std::vector<HSide> sides; std::vector<int> pa, pb;
    int rc = kf_sides(k1, flag1, k2s, flags2, n2, sides, pa, pb, who);
    if (rc) return rc;
    if (!k1->geom) { orbx_set_error("%s: keyframe made without positions / octaves / u_right", who); return ORBX_E_INVALID; }
    for (int i = 0; i < n2; i++) {
        if (!k2s[i]->geom) { orbx_set_error("%s: keyframe %d made without positions / octaves / u_right", who, i); return ORBX_E_INVALID; }
        if (k2s[i]->max_octave >= nlevels2) { orbx_set_error("%s: keyframe %d has octave %d, tables have %d levels", who, i, k2s[i]->max_octave, nlevels2); return ORBX_E_INVALID; }
    }
    rc = match_call(2, k1->device, sides.data(), n2 + 1, pa.data(), pb.data(), n2, F12s, epipoles, sf2, sig2, nlevels2, 0.f, check_orientation, only_stereo,
                    cap, pairs, 2 * (size_t)cap, npairs);
    if (rc) return rc;
    for (int i = 0; i < n2; i++)
        if (npairs[i] > cap) { orbx_set_error("pair capacity %d < %d matches", cap, npairs[i]); return ORBX_E_CAPACITY; }
    return ORBX_OK;
}
