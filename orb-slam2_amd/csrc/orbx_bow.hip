// orbx_bow.hip — ORBmatcher::SearchByBoW (KF,F) / (KF,KF) for gfx950, the THROUGHPUT forms (reference src/ORBmatcher.cc:171-303,
// :568-702, :1687-1728): thousands of (keyframe, frame) pairs per launch, FeatureVectors intersected on the device (the batched,
// device-resident relocalisation chain of BASELINE config 3).  The per-call forms, SearchForTriangulation and the resident
// keyframe handles live in orbx_match.hip.
//
// One 256-thread workgroup per keyframe pair.  The two FeatureVectors (CSR, node ids ascending)
// are intersected by binary search (same node set as the reference's merge join); every shared
// vocabulary node is owned by one wave, which walks the first side's features sequentially (the
// greedy "already claimed" state of :222 / :622 is per node because a feature lives in exactly one
// node) and spreads the second side's features over its 64 lanes: 256-bit Hamming by popcount,
// wave min-reductions for best / second best.  Rotation histogram + ComputeThreeMaxima run once
// per pair in LDS.
#include "orbx_device.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define BOW_TH_LOW 50
#define BOW_HISTO 30

// rotation bin of src/ORBmatcher.cc:253-258 (factor = 1/30 with HISTO_LENGTH = 30: upstream quirk kept)
__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / BOW_HISTO;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == BOW_HISTO) bin = 0;
    return (unsigned)bin < BOW_HISTO ? bin : 0; // the reference asserts the range; NaN / huge angles must not index out of the histogram
}

__device__ __forceinline__ int find_node(const uint32_t *ids, int n, uint32_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < n && ids[lo] == key) ? lo : -1;
}

__device__ __forceinline__ void load_desc(const uint32_t *d, long long idx, uint32_t *out)
{
    const uint4 *s = reinterpret_cast<const uint4 *>(d + idx * 8);
    const uint4 v0 = s[0], v1 = s[1];
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w;
    out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
}

// ComputeThreeMaxima (:1687-1728) on per-bin counts + clearing of the other bins (:282-300).
// bins[i] = bin of match slot i or 255; match[] entries outside the 3 dominant bins become -1.
// Returns (in *out_n, by thread 0) the number of surviving matches.
__device__ void histogram_filter(int32_t *match, const uint8_t *bins, int nslots, int check_ori, int *hist, int *keep3, int *s_cnt, int *out_n)
{
    const int tid = threadIdx.x;
    if (tid < BOW_HISTO) hist[tid] = 0;
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    if (check_ori) {
        for (int i = tid; i < nslots; i += blockDim.x)
            if (bins[i] != 255) atomicAdd(&hist[bins[i]], 1);
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < BOW_HISTO; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            keep3[0] = i1; keep3[1] = i2; keep3[2] = i3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < nslots; i += blockDim.x) {
        const int b = bins[i];
        if (b == 255) continue;
        if (check_ori && b != keep3[0] && b != keep3[1] && b != keep3[2]) match[i] = -1;
        else local++;
    }
    if (local) atomicAdd(s_cnt, local);
    __syncthreads();
    if (tid == 0) *out_n = *s_cnt;
}

extern __shared__ __align__(16) unsigned char bow_smem[];

// MODE 0: SearchByBoW(KF, F)  — match[nB] indexed by the F feature, value = KF feature
// MODE 1: SearchByBoW(KF, KF) — match[nA] indexed by the KF1 feature, value = KF2 feature
//
// Work decomposition.  Inside one shared vocabulary node the reference is sequential (a feature of the second side
// that an earlier first-side feature has taken is skipped, :222 / :622), across nodes it is independent, and a
// typical node holds ~10 x 10 features.  Walking the first side of a node with one wave (64 lanes over ~10 candidates,
// two wave reductions per step) left most lanes idle, so the kernel is split in two phases per batch of nodes:
//   1. all Hamming distances of all shared nodes, one (node, a, b) triple per thread, into an LDS table (u16;
//      0xFE = the first-side feature takes no part, 0xFF = the second-side feature takes no part);
//   2. the greedy walk of a node by ONE thread on that table (pure LDS reads, no descriptor traffic), 64 nodes per
//      wave side by side.
// Nodes are packed into passes of at most BOW_MATCAP table entries; a single node larger than that is walked by a
// wave straight from global memory (node_greedy_wave, the former kernel body).
// Table entries are BYTES: a distance is clamped to 253 (0xFE = the first-side feature takes no part, 0xFF = the second-side one).
// An accepted match has best1 <= TH_LOW = 50, and the ratio test best1 < nnratio * best2 reads the same for best2 = 253 as for any
// larger best2 as long as nnratio * 253 > 50, i.e. nnratio >= 0.2 (bow_launch sends smaller ratios -- ORB-SLAM2 uses 0.6 to 0.9 --
// to the wave form).  8192 one-byte entries cost the LDS that 4096 two-byte ones did and halve the passes of a typical
// 1000 x 1000 pair (14 700 distances: 4.2 passes -> 2.3).
#ifndef BOW_MATCAP
#define BOW_MATCAP 8192
#endif
#define BOW_DCLAMP 253
#ifndef BOW_P1_UNROLL
#define BOW_P1_UNROLL 4      // second-side descriptors in flight per thread in the distance phase
#endif
#define BOW_CHUNK 256

template <int MODE>
__device__ void node_greedy_wave(const DevFeat &A, const DevFeat &B, int a0, int a1, int b0, int b1, uint8_t *claimed,
                                 uint8_t *bins, int32_t *match, float nnratio, int lane)
{
    for (int i1 = a0; i1 < a1; i1++) {
        if (!A.sflag[i1]) continue;
        const int idx1 = (int)A.feat[i1];
        uint32_t da[8];
        load_desc(A.sdesc, i1, da);
        unsigned k1 = 0xFFFFFFFFu; // dist<<20 | position: first index wins ties (strict <, :229-239)
        int l1 = 256, l2 = 256;
        for (int j = b0 + lane; j < b1; j += 64) {
            const int idx2 = (int)B.feat[j];
            if (claimed[idx2]) continue;
            if (MODE == 1 && !B.sflag[j]) continue;
            uint32_t db[8];
            load_desc(B.sdesc, j, db);
            const int dist = hamming256(da, db);
            if (dist < l1) { l2 = l1; l1 = dist; k1 = ((unsigned)dist << 20) | (unsigned)(j - b0); }
            else if (dist < l2) l2 = dist;
        }
        const unsigned kbest = wave_min_u32(k1);
        if (kbest == 0xFFFFFFFFu) continue;
        const int best1 = (int)(kbest >> 20);
        // second smallest of the multiset: the owner of the winner contributes its own runner-up
        const unsigned second = wave_min_u32((unsigned)(k1 == kbest ? l2 : l1));
        const int best2 = (int)second;
        const bool ok_dist = MODE == 0 ? best1 <= BOW_TH_LOW : best1 < BOW_TH_LOW;
        if (ok_dist && (float)best1 < nnratio * (float)best2) {
            const int idx2 = (int)B.feat[b0 + (int)(kbest & 0xFFFFF)];
            if (lane == 0) {
                claimed[idx2] = 1;
                const int bin = rot_bin(A.angle[idx1], B.angle[idx2]);
                if (MODE == 0) { match[idx2] = idx1; bins[idx2] = (uint8_t)bin; }
                else { match[idx1] = idx2; bins[idx1] = (uint8_t)bin; }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
}

// Latency form: one 1024-thread workgroup per pair, the shared nodes dealt over its 16 waves, each node walked by a
// wave straight from global memory (node_greedy_wave).  Used when a call has too few pairs to fill the GPU (a single
// query against a keyframe set: 123 us for 500 keyframes); the table kernel below is the throughput form.
template <int MODE>
__global__ __launch_bounds__(1024) void k_bow_wave(const DevFeat *__restrict__ sides_a, const DevFeat *__restrict__ sides_b,
                                                  int b_shared, float nnratio, int check_ori, int32_t *__restrict__ match_out,
                                                  int match_stride, int *__restrict__ nmatches)
{
    __shared__ int hist[BOW_HISTO];
    __shared__ int keep3[3];
    __shared__ int s_cnt;
    const int pair = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DevFeat A = sides_a[pair];
    const DevFeat B = sides_b[b_shared ? frame : pair];
    const int nslots = MODE == 0 ? B.n : A.n;
    uint8_t *claimed = bow_smem;                 // [B.n]
    uint8_t *bins = bow_smem + ((B.n + 15) & ~15); // [nslots]
    int32_t *match = match_out + ((long long)frame * gridDim.x + pair) * match_stride;
    nmatches += (long long)frame * gridDim.x;
    const int nwaves = blockDim.x >> 6;
    for (int i = tid; i < B.n; i += blockDim.x) claimed[i] = 0;
    for (int i = tid; i < nslots; i += blockDim.x) { bins[i] = 255; match[i] = -1; }
    __syncthreads();
    for (int ia = wv; ia < A.nnodes; ia += nwaves) {
        const int ib = find_node(B.node_id, B.nnodes, A.node_id[ia]);
        if (ib < 0) continue;
        node_greedy_wave<MODE>(A, B, A.node_off[ia], A.node_off[ia + 1], B.node_off[ib], B.node_off[ib + 1], claimed, bins, match,
                               nnratio, lane);
    }
    __syncthreads();
    histogram_filter(match, bins, nslots, check_ori, hist, keep3, &s_cnt, nmatches + pair);
}

#ifndef BOW_ROWCAP
#define BOW_ROWCAP 512
#endif

#ifdef ORBX_DIAG
__device__ unsigned long long g_bow_stat[8]; // diagnostic build: [0] table entries, [1] fixpoint rounds, [2] passes, [3] wave-fallback nodes, [4] workgroups, [5] rows, [6] phase-1 cycles, [7] phase-2 cycles (thread 0)
extern "C" int orbx_diag_bow_stats(unsigned long long *out, int reset)
{
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bow_stat), sizeof(unsigned long long) * 8));
    if (reset) { unsigned long long z[8] = { 0 }; ORBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bow_stat), z, sizeof z)); }
    return ORBX_OK;
}
#define BOW_STAT(i, v) do { if (tid == 0) atomicAdd(&g_bow_stat[i], (unsigned long long)(v)); } while (0)
#else
#define BOW_STAT(i, v) do { } while (0)
#endif

// ---------------------------------------------------------------- k_bow2: the throughput form without a distance table (round 4)
// Counters of k_bow on config 3 (profiles/r04_sq_bow_before.txt): 6100 VALU wave-instructions per wave, 71 % VALU busy -- NOT latency-bound as
// rounds 2-3 believed: a third of them in the fixpoint's walks over whole table rows (one byte and one `own` word per column, per row, per
// round), the rest in the distance phase and its bookkeeping.  What the greedy walk of a node needs from a first-side row is its best and
// second-best column among those no EARLIER row holds; claims only ever REMOVE columns, so
//   * a row whose smallest distance fails the TH_LOW test can never match: it leaves the fixpoint at once;
//   * the three smallest keys (distance << 16 | column, first column wins ties as the reference's strict `<` does) decide the row whenever
//     at least two of them are free -- and even with one free, when best1 < nnratio * d3 already holds (best2 >= d3);
//   * only the rest (two of the top three taken AND the ratio test open) walks the node's columns again, straight from global memory.
// So the distance phase keeps three keys per row in REGISTERS (thread t owns rows t, t + 256, ...), there is no table, hence no passes over
// a pair (one pass per 1024 first-side rows instead of one per 8192 table entries: 2.1 -> 1.0 passes per 1000 x 1000 pair), no clamped byte
// distances (exact for every nnratio) and 19 KB of LDS per workgroup instead of 25.  The match row lives in LDS until the pair is done and
// leaves either as the dense row (coalesced, written once: it used to be initialised, scattered into and filtered in global memory) or as
// the compact list of (slot, value) pairs in slot order that Tracking::Relocalization hands to its PnP solver next (src/Tracking.cc:1682-1693).
#ifndef BOW2_ROWS
#define BOW2_ROWS 1024                          // first-side rows per pass: BOW2_RPT per thread
#endif
#define BOW2_RPT (BOW2_ROWS / 256)
#ifndef BOW2_U
#define BOW2_U 4                                // second-side descriptors in flight per thread in the distance phase
#endif
#define BOW2_SENT 0x0100FFFFu                   // "no column": distance 256, column 0xFFFF -- above every real key

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the rare full walk of a row: best / second best among the columns no earlier row of the node holds (own: tag << 16 | earliest row)
template <int MODE>
__device__ __forceinline__ void bow2_row_scan(const DevFeat &A, const DevFeat &B, long long pa, long long pb0, int bcn, const unsigned *own, int boff,
                                           unsigned r, unsigned tag, int *best1, int *best2, int *bj)
{
    uint32_t da[8];
    load_desc(A.sdesc, pa, da);
    int b1 = 256, b2 = 256, j1 = -1;
    for (int j = 0; j < bcn; j++) {
        if (MODE == 1 && !B.sflag[pb0 + j]) continue;
        const unsigned v = own[boff + j];
        if ((v >> 16) == tag && (v & 0xFFFFu) < r) continue;
        uint32_t db[8];
        load_desc(B.sdesc, pb0 + j, db);
        const int d = hamming256(da, db);
        if (d < b1) { b2 = b1; b1 = d; j1 = j; } else if (d < b2) b2 = d;
    }
    *best1 = b1; *best2 = b2; *bj = j1;
}

// OUT 0: match_out = dense rows [frame][pair][match_stride]; OUT 1: match_out = compact lists [frame][pair][2 * match_stride] of (slot, value)
// in slot order, the first min(count, match_stride) of them; nmatches[frame][pair] = count either way
#ifndef BOW2_WPE
#define BOW2_WPE 6      // registers capped at 80: six workgroups per CU.  4 (103 registers): 0.76 ms per 500 x 32 launch; 5: 0.64, no spills, 107 MB of
                        // traffic; 6: 0.60 ms, three spilled dwords per thread (194 MB, 53 of them spill writes); 7 / 8: 0.60 / 0.59 ms with 8 / 17 spilled
                        // dwords (340 / 580 MB): the time no longer moves, the scratch traffic does
#endif
template <int MODE, int OUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BOW2_WPE, 8))) void k_bow2(const DevFeat *__restrict__ sides_a, const DevFeat *__restrict__ sides_b,
                                             int b_shared, float nnratio, int check_ori, int32_t *__restrict__ match_out,
                                             int match_stride, int *__restrict__ nmatches, int xmap_npairs, int xmap_nframes)
{
    __shared__ int hist[BOW_HISTO];
    __shared__ int keep3[3];
    __shared__ int s_cnt;
    __shared__ int s_w[4];
    __shared__ int s_npass, s_nfall, s_changed[2];
    __shared__ int c_aoff[BOW_CHUNK], c_boff[BOW_CHUNK], c_roff[BOW_CHUNK];      // node tables of the chunk (shared nodes only, node order)
    __shared__ unsigned c_cnt[BOW_CHUNK];                                         // a_cnt | b_cnt << 16
    __shared__ uint16_t pass_first[BOW_CHUNK + 1], pass_rows[BOW_CHUNK];
    __shared__ uint16_t fall_list[BOW_CHUNK];
    __shared__ uint16_t choice[BOW2_ROWS];         // per first-side row of the pass: chosen column of its node or 0xFFFF
    __shared__ uint8_t row_node[BOW2_ROWS];
    __shared__ uint8_t node_dirty[2][BOW_CHUNK];
    // Grid: (pairs, frames), or -- a keyframe set against a batch of frames -- ONE dimension in which an XCD owns whole keyframes: workgroups are
    // dealt round-robin over the 8 XCDs in dispatch order, so workgroup L runs on XCD L % 8; XCD x takes the keyframes x, x + 8, ... and each
    // of them with all its frames in consecutive workgroups.  A keyframe's 41 KB then come over the fabric once (its 32 uses hit the XCD's
    // L2 while they are fresh) instead of once per few frames, and the batch's frames (1.3 MB) stay in every L2.
    int pair, frame, npairs_g;
    if (xmap_nframes > 0) {
#ifndef BOW2_XG
#define BOW2_XG 16
#endif
        const unsigned L = blockIdx.x, slot = L >> 3, per = (unsigned)xmap_nframes * BOW2_XG, grp = slot / per, rem = slot - grp * per;
        frame = (int)(rem / BOW2_XG); pair = (int)((grp * BOW2_XG + rem % BOW2_XG) * 8u + (L & 7u)); npairs_g = xmap_npairs;
        if (pair >= npairs_g) return;
    } else { pair = blockIdx.x; frame = blockIdx.y; npairs_g = gridDim.x; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DevFeat A = sides_a[pair];
    const DevFeat B = sides_b[b_shared ? frame : pair];
    const int nslots = MODE == 0 ? B.n : A.n;
    const int mB = B.nnodes ? B.node_off[B.nnodes] : 0;
    uint8_t *claimed = bow_smem;                                                   // [B.n]   (wave fallback only)
    uint8_t *bins = bow_smem + ((B.n + 15) & ~15);                                 // [nslots]
    unsigned *own = reinterpret_cast<unsigned *>(bins + ((nslots + 15) & ~15));    // [B.n] per list position: round tag << 16 | earliest row choosing it
    int32_t *match = reinterpret_cast<int32_t *>(own + ((B.n + 3) & ~3));          // [nslots] the pair's match row, in LDS until the end
    for (int i = tid; i < B.n; i += 256) claimed[i] = 0;
    for (int i = tid; i < nslots; i += 256) { bins[i] = 255; match[i] = -1; }
    __syncthreads();
    for (int base = 0; base < A.nnodes; base += BOW_CHUNK) {
        // ---- shared nodes of this chunk, compacted in node order
        const int ia = base + tid;
        int ao = 0, ac = 0, bo = 0, bc = 0;
        if (ia < A.nnodes) {
            const int ib = find_node(B.node_id, B.nnodes, A.node_id[ia]);
            if (ib >= 0) {
                ao = A.node_off[ia]; ac = A.node_off[ia + 1] - ao;
                bo = B.node_off[ib]; bc = B.node_off[ib + 1] - bo;
            }
        }
        const bool shared = ac > 0 && bc > 0;
        const bool big = shared && (ac > BOW2_ROWS || bc >= 0xFFFF);
        const bool tab = shared && !big;
        int ncomp, nbig, rows_tot;
        const int k = block_excl_scan256(tab ? 1 : 0, &ncomp, s_w);
        const int kb = block_excl_scan256(big ? 1 : 0, &nbig, s_w);
        const int ro = block_excl_scan256(tab ? ac : 0, &rows_tot, s_w);
        if (tab) { c_aoff[k] = ao; c_boff[k] = bo; c_cnt[k] = (unsigned)ac | ((unsigned)bc << 16); c_roff[k] = ro; }
        if (big) fall_list[kb] = (uint16_t)tid;
        __syncthreads();
        if (rows_tot <= BOW2_ROWS) {               // the usual case: the chunk is one pass
            if (tid == 0) { s_npass = ncomp ? 1 : 0; pass_first[0] = 0; pass_first[1] = (uint16_t)ncomp; pass_rows[0] = (uint16_t)rows_tot; s_nfall = nbig; }
        } else if (tid == 0) {                     // more than BOW2_ROWS listed first-side rows in 256 nodes: pack the nodes, in order, into passes
            int np = 0, roff = 0;
            pass_first[0] = 0;
            for (int q = 0; q < ncomp; q++) {
                const int an = (int)(c_cnt[q] & 0xFFFF);
                if (roff + an > BOW2_ROWS) { pass_rows[np] = (uint16_t)roff; np++; pass_first[np] = (uint16_t)q; roff = 0; }
                c_roff[q] = roff;
                roff += an;
            }
            pass_rows[np] = (uint16_t)roff; np++; pass_first[np] = (uint16_t)ncomp;
            s_npass = np; s_nfall = nbig;
        }
        __syncthreads();
        const int npass = s_npass, nfall = s_nfall;
        BOW_STAT(2, npass); BOW_STAT(3, nfall); BOW_STAT(4, base == 0);
        for (int ps = 0; ps < npass; ps++) {
            const int q0 = pass_first[ps], q1 = pass_first[ps + 1], rows = pass_rows[ps];
            for (int q = q0 + wv; q < q1; q += 4) { // the node of every row
                const int acn = (int)(c_cnt[q] & 0xFFFF), rb = c_roff[q];
                for (int i1 = lane; i1 < acn; i1 += 64) row_node[rb + i1] = (uint8_t)q;
            }
            {   // (an opaque copy of tid: &own[tid] is loop-invariant, and kept across the passes it was the register that went to scratch --
                // 33 MB of spill stores per launch of 16 000 workgroups)
                int t_ = tid;
                asm volatile("" : "+v"(t_));
                for (int i = t_; i < mB; i += 256) own[i] = 0xFFFFFFFFu;
            }
            node_dirty[0][tid] = 1; node_dirty[1][tid] = 0;
            __syncthreads();
            // ---- distances: thread t owns rows t, t + 256, ...: its descriptor once, the node's columns four at a time, the three smallest
            // keys kept in registers.  alive = the row can still match (flagged, and its smallest distance passes the TH_LOW test)
            unsigned K1[BOW2_RPT], K2[BOW2_RPT], K3[BOW2_RPT];
            unsigned alive = 0, complete = 0;
#pragma unroll
            for (int u4 = 0; u4 < BOW2_RPT; u4++) {
                K1[u4] = K2[u4] = K3[u4] = BOW2_SENT;
                const int r = tid + 256 * u4;
                if (r >= rows) continue;
                choice[r] = 0xFFFF;
                const int lo = row_node[r];
                const int bcn = (int)(c_cnt[lo] >> 16), i1 = r - c_roff[lo];
                const long long pa = c_aoff[lo] + i1, pb0 = c_boff[lo];
                if (!A.sflag[pa]) continue;                       // A.flag (:205-210 / :606-613)
                uint32_t da[8];
                load_desc(A.sdesc, pa, da);
                unsigned k1 = BOW2_SENT, k2 = BOW2_SENT, k3 = BOW2_SENT;
                int nv = 0;
                for (int j0 = 0; j0 < bcn; j0 += BOW2_U) {
                    unsigned key[BOW2_U];
#pragma unroll
                    for (int u = 0; u < BOW2_U; u++) {
                        const int j = min(j0 + u, bcn - 1);
                        uint32_t db[8];
                        load_desc(B.sdesc, pb0 + j, db);
                        const bool col = j0 + u < bcn && (MODE == 1 ? B.sflag[pb0 + j] != 0 : true);
                        key[u] = col ? ((unsigned)hamming256(da, db) << 16) | (unsigned)j : BOW2_SENT;
                        nv += col ? 1 : 0;
                    }
#pragma unroll
                    for (int u = 0; u < BOW2_U; u++) {             // sorted insertion into (k1 <= k2 <= k3): a minimum and two medians (v_med3_u32)
                        const unsigned o1 = k1, o2 = k2;
                        k1 = min(o1, key[u]);
                        k2 = umed3(o1, key[u], o2);
                        k3 = umed3(o2, key[u], k3);
                    }
                }
                const int d1 = (int)(k1 >> 16);
                const bool can = MODE == 0 ? d1 <= BOW_TH_LOW : d1 < BOW_TH_LOW;     // claims only remove columns: best1 never gets smaller than this
                if (can) alive |= 1u << u4;
                if (nv <= 3) complete |= 1u << u4;
                K1[u4] = k1; K2[u4] = k2; K3[u4] = k3;
            }
            BOW_STAT(5, rows);
            if (tid < 2) s_changed[tid] = 0;
            __syncthreads();
            // ---- the greedy walk as a fixpoint over rows (see k_bow): choice(r) = best column no EARLIER row of the node currently holds.
            // own[] carries the round in its high half, so it is filled once per round and never cleared.
            for (int round = 0; round <= rows; round++) {
                const unsigned tag = 0xFFFEu - (unsigned)round;
#pragma unroll
                for (int u4 = 0; u4 < BOW2_RPT; u4++) {
                    const int r = tid + 256 * u4;
                    if (!((alive >> u4) & 1u) || r >= rows) continue;
                    const unsigned c = choice[r];
                    if (c != 0xFFFF) atomicMin(&own[c_boff[row_node[r]] + (int)c], (tag << 16) | (unsigned)r);
                }
                __syncthreads();
                int changed = 0;
                unsigned need_scan = 0;
                const uint8_t *was = node_dirty[round & 1];
                uint8_t *now = node_dirty[(round & 1) ^ 1];
#pragma unroll
                for (int u4 = 0; u4 < BOW2_RPT; u4++) {
                    const int r = tid + 256 * u4;
                    if (!((alive >> u4) & 1u) || r >= rows) continue;
                    const int lo = row_node[r];
                    if (!was[lo]) continue;          // a node none of whose rows changed in the previous round is settled
                    const int boff = c_boff[lo];
                    const unsigned k1 = K1[u4], k2 = K2[u4], k3 = K3[u4];
                    auto taken = [&](unsigned key) -> bool {
                        if (key == BOW2_SENT) return false;
                        const unsigned v = own[boff + (int)(key & 0xFFFFu)];
                        return (v >> 16) == tag && (v & 0xFFFFu) < (unsigned)r;
                    };
                    const bool t1 = taken(k1), t2 = taken(k2), t3 = taken(k3);
                    // the first two free keys of the top three; `open` = the second one lies beyond them
                    unsigned b1, b2;
                    bool have1 = true, open = false;
                    if (!t1) { b1 = k1; if (!t2) b2 = k2; else if (!t3) b2 = k3; else { b2 = k3; open = true; } }
                    else if (!t2) { b1 = k2; if (!t3) b2 = k3; else { b2 = k3; open = true; } }
                    else if (!t3) { b1 = k3; b2 = k3; open = true; }
                    else { b1 = b2 = BOW2_SENT; have1 = false; open = true; }
                    const bool all_known = (complete >> u4) & 1u;      // the node has at most three columns: beyond the top three there is nothing
                    int best1 = (int)(b1 >> 16), best2 = open ? (all_known ? 256 : (int)(k3 >> 16)) : (int)(b2 >> 16), bj = (int)(b1 & 0xFFFFu);
                    if (!have1 && all_known) { best1 = 256; bj = -1; }
                    bool decided = all_known || !open;
                    if (!decided && have1) {
                        // best2 >= d3: the ratio test already holds with d3, or the distance test already fails -> no need to know best2
                        const bool okd = MODE == 0 ? best1 <= BOW_TH_LOW : best1 < BOW_TH_LOW;
                        if (!okd || (float)best1 < nnratio * (float)(k3 >> 16)) { decided = true; best2 = okd ? 256 : best2; }
                    }
                    if (!decided) { need_scan |= 1u << u4; continue; }      // (walked below, outside the unrolled code: keeps this path's registers low)
                    unsigned nc = 0xFFFF;
                    const bool ok_dist = MODE == 0 ? best1 <= BOW_TH_LOW : best1 < BOW_TH_LOW;
                    if (bj >= 0 && best1 < 256 && ok_dist && (float)best1 < nnratio * (float)best2) nc = (unsigned)bj;
                    if (nc != choice[r]) { choice[r] = (uint16_t)nc; changed = 1; now[lo] = 1; }
                }
                if (need_scan) {
#pragma unroll 1
                    for (int u4 = 0; u4 < BOW2_RPT; u4++) {
                        if (!((need_scan >> u4) & 1u)) continue;
                        const int r = tid + 256 * u4, lo = row_node[r], boff = c_boff[lo], i1 = r - c_roff[lo];
                        int best1, best2, bj;
                        bow2_row_scan<MODE>(A, B, (long long)c_aoff[lo] + i1, (long long)boff, (int)(c_cnt[lo] >> 16), own, boff, (unsigned)r, tag, &best1, &best2, &bj);
                        unsigned nc = 0xFFFF;
                        const bool ok_dist = MODE == 0 ? best1 <= BOW_TH_LOW : best1 < BOW_TH_LOW;
                        if (bj >= 0 && best1 < 256 && ok_dist && (float)best1 < nnratio * (float)best2) nc = (unsigned)bj;
                        if (nc != choice[r]) { choice[r] = (uint16_t)nc; changed = 1; now[lo] = 1; }
                    }
                }
                BOW_STAT(1, 1);
                // "did any row change?" through a flag per round parity (set by whoever changed, read after the barrier, the other parity's
                // flag cleared for the next round): __syncthreads_or costs two barriers, an LDS reduction and the y / z work-item ids,
                // which went to scratch in the (KF, KF) form
                if (changed) s_changed[round & 1] = 1;
                __syncthreads();
                if (!s_changed[round & 1]) break;
                if (tid == 0) s_changed[(round & 1) ^ 1] = 0;
                node_dirty[round & 1][tid] = 0;
                __syncthreads();
            }
            // ---- results of the pass
            for (int r = tid; r < rows; r += 256) {
                const unsigned c = choice[r];
                if (c == 0xFFFF) continue;
                const int lo = row_node[r];
                const int idx1 = (int)A.feat[c_aoff[lo] + (r - c_roff[lo])], idx2 = (int)B.feat[c_boff[lo] + (int)c];
                const int bin = rot_bin(A.angle[idx1], B.angle[idx2]);
                if (MODE == 0) { match[idx2] = idx1; bins[idx2] = (uint8_t)bin; }
                else { match[idx1] = idx2; bins[idx1] = (uint8_t)bin; }
            }
            __syncthreads();
        }
        // ---- nodes with more rows than a pass holds (or 65535+ columns): one wave each, straight from global memory
        for (int f = wv; f < nfall; f += 4) {
            const int ja = base + fall_list[f];
            const int ib = find_node(B.node_id, B.nnodes, A.node_id[ja]);
            node_greedy_wave<MODE>(A, B, A.node_off[ja], A.node_off[ja + 1], B.node_off[ib], B.node_off[ib + 1], claimed, bins, match,
                                   nnratio, lane);
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const long long pi = (long long)frame * npairs_g + pair;
    histogram_filter(match, bins, nslots, check_ori, hist, keep3, &s_cnt, nmatches + pi);
    __syncthreads();
    if (OUT == 0) {
        int32_t *row = match_out + pi * match_stride;
        for (int i = tid; i < nslots; i += 256) row[i] = match[i];
    } else {
        int32_t *lst = match_out + pi * 2 * (long long)match_stride;
        int run = 0;
        for (int i0 = 0; i0 < nslots; i0 += 256) {           // (slot, value) in slot order
            const int i = i0 + tid;
            const int v = i < nslots ? match[i] : -1;
            int tot;
            const int pos = run + block_excl_scan256(v >= 0 ? 1 : 0, &tot, s_w);
            if (v >= 0 && pos < match_stride) { lst[2 * pos] = i; lst[2 * pos + 1] = v; }
            run += tot;
            __syncthreads();
        }
    }
}

// 0 = choose by size, 1 = always the wave form, 2 = always the table form (explicit debug entry point, no environment lookup)
static std::atomic<int> g_bow_form{0};
extern "C" int orbx_debug_set_bow_form(int form)
{
    if (form < 0 || form > 2) { orbx_set_error("orbx_debug_set_bow_form: 0 (auto), 1 (wave) or 2 (table)"); return ORBX_E_INVALID; }
    g_bow_form.store(form, std::memory_order_relaxed);
    return ORBX_OK;
}

int orbx_bow_forced_form() { return g_bow_form.load(std::memory_order_relaxed); }

// pairs < BOW_TABLE_MIN_PAIRS: too few workgroups to fill 256 CUs, the 16-wave latency form is faster per call
#define BOW_TABLE_MIN_PAIRS 4096
// compact = 1: d_match receives (slot, value) lists of capacity `stride` pairs per (frame, pair) instead of dense rows of `stride` ints (k_bow2 only)
template <int MODE>
static int bow_launch(int npairs_x, int nframes_y, int max_b, int max_slots, hipStream_t st, const DevFeat *dA, const DevFeat *dB,
                      int b_shared, float nnratio, int check_ori, int32_t *d_match, int stride, int *d_n, int compact = 0)
{
    const size_t base = (size_t)((max_b + 15) & ~15) + (size_t)((max_slots + 15) & ~15) + 16;
    bool table = (long long)npairs_x * nframes_y >= BOW_TABLE_MIN_PAIRS || compact;
    const int forced = g_bow_form.load(std::memory_order_relaxed); // orbx_debug_set_bow_form: the parity tests run both forms on small inputs
    if (forced && !compact) table = forced == 2;
    const size_t lds = table ? base + 4 * (size_t)((max_b + 3) & ~3) + 4 * (size_t)((max_slots + 3) & ~3) : base;
    if (lds > 120 * 1024) { orbx_set_error("feature sets too large for LDS"); return ORBX_E_INVALID; }
    if (table) {
        void (*kern)(const DevFeat *, const DevFeat *, int, float, int, int32_t *, int, int *, int, int) = compact ? k_bow2<MODE, 1> : k_bow2<MODE, 0>;
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef BOW2_NOXMAP
        const bool xmap = false;
#else
        const bool xmap = b_shared && nframes_y >= 2 && npairs_x >= 8 && (long long)((npairs_x + 7) / 8) * 8 * nframes_y < (1ll << 30);
#endif
        if (xmap)    // keyframes against a batch of frames: an XCD owns whole keyframes (see the kernel)
            hipLaunchKernelGGL(kern, dim3((unsigned)(((npairs_x + 8 * BOW2_XG - 1) / (8 * BOW2_XG)) * 8 * BOW2_XG * nframes_y)), dim3(256), lds, st, dA, dB, b_shared, nnratio, check_ori, d_match, stride, d_n,
                               npairs_x, nframes_y);
        else
            hipLaunchKernelGGL(kern, dim3(npairs_x, nframes_y), dim3(256), lds, st, dA, dB, b_shared, nnratio, check_ori, d_match, stride, d_n, 0, 0);
    } else {
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_wave<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_bow_wave<MODE>, dim3(npairs_x, nframes_y), dim3(1024), lds, st, dA, dB, b_shared, nnratio, check_ori, d_match, stride, d_n);
    }
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ---------------------------------------------------------------- host side

struct BowCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t *h_blob = nullptr; size_t h_cap = 0;   // pinned staging
    uint8_t *d_blob = nullptr; size_t d_cap = 0;
    int32_t *d_out = nullptr; size_t out_cap = 0;  // match / pairs / counts
    int32_t *h_out = nullptr; size_t h_out_cap = 0;
};
static thread_local BowCtx g_bow[16];

static int bow_ctx(int device, BowCtx **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 16) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    BowCtx *c = &g_bow[device];
    if (!c->stream) { ORBX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->device = device; }
    *out = c;
    return ORBX_OK;
}

static int bow_reserve(BowCtx *c, size_t blob, size_t out_ints)
{
    if (blob > c->h_cap) {
        if (c->h_blob) ORBX_HIP(hipHostFree(c->h_blob));
        c->h_blob = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&c->h_blob, blob * 2, hipHostMallocDefault));
        c->h_cap = blob * 2;
    }
    if (blob > c->d_cap) {
        if (c->d_blob) ORBX_HIP(hipFree(c->d_blob));
        c->d_blob = nullptr;
        ORBX_HIP(hipMalloc((void **)&c->d_blob, blob * 2));
        c->d_cap = blob * 2;
    }
    if (out_ints > c->out_cap) {
        if (c->d_out) ORBX_HIP(hipFree(c->d_out));
        c->d_out = nullptr;
        ORBX_HIP(hipMalloc((void **)&c->d_out, out_ints * 2 * sizeof(int32_t)));
        c->out_cap = out_ints * 2;
    }
    if (out_ints > c->h_out_cap) {
        if (c->h_out) ORBX_HIP(hipHostFree(c->h_out));
        c->h_out = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&c->h_out, out_ints * 2 * sizeof(int32_t), hipHostMallocDefault));
        c->h_out_cap = out_ints * 2;
    }
    return ORBX_OK;
}

static size_t a16(size_t v) { return (v + 15) & ~(size_t)15; }

int orbx_feat_validate(const orbx_featset *f, int need_geom, int need_flag)
{
    if (!f || f->n < 0 || f->nnodes < 0) return 0;
    if (f->n >= (1 << 20)) return 0;
    if (f->n && (!f->desc || (need_flag && !f->flag) || !f->angle)) return 0;
    if (f->nnodes && (!f->node_id || !f->node_off || !f->feat)) return 0;
    if (need_geom && f->n && (!f->x || !f->y || !f->octave || !f->u_right)) return 0;
    if (f->nnodes) {
        if (f->node_off[0] != 0) return 0;
        for (int i = 0; i < f->nnodes; i++) {
            if (f->node_off[i + 1] < f->node_off[i]) return 0;
            if (i && f->node_id[i] <= f->node_id[i - 1]) return 0;
        }
        const int m = f->node_off[f->nnodes];
        if (m > f->n) return 0; // a FeatureVector lists every feature at most once (the kernels size per-position tables by n)
        for (int i = 0; i < m; i++) if (f->feat[i] >= (uint32_t)f->n) return 0; // indices must be in range: the kernel trusts them
    }
    return 1;
}

static size_t feat_bytes(const orbx_featset *f, int geom)
{
    const size_t m = f->nnodes ? (size_t)f->node_off[f->nnodes] : 0;
    size_t s = a16((size_t)f->n * 32) + a16((size_t)f->nnodes * 4) + a16(((size_t)f->nnodes + 1) * 4) + a16(m * 4) +
               a16((size_t)f->n) + a16((size_t)f->n * 4) + a16(m * 32) + a16(m);
    if (geom) s += 4 * a16((size_t)f->n * 4);
    return s;
}

// copy one featset into the staging blob at *off; pointers in `d` refer to the device blob.
// drop_unflagged: the node lists keep only features with flag != 0.  In the SearchByBoW searches a first-side feature
// without flag is skipped before anything else (src/ORBmatcher.cc:205-210, :606-613) and, for (KF, KF), so is a
// second-side one (:627-631): leaving them out of the lists changes no result and spares the kernels the distance
// evaluations of rows / columns that can never match (40 % of a keyframe's features on the bench data).
static void feat_pack(const orbx_featset *f, int geom, uint8_t *h, const uint8_t *dbase, size_t *off, DevFeat *d, bool drop_unflagged = false)
{
    auto put = [&](const void *src, size_t bytes) -> const uint8_t * {
        const uint8_t *dp = dbase + *off;
        if (bytes && src) memcpy(h + *off, src, bytes);
        *off += a16(bytes);
        return dp;
    };
    const size_t m = f->nnodes ? (size_t)f->node_off[f->nnodes] : 0;
    static const int32_t zero_off[1] = { 0 };
    d->n = f->n; d->nnodes = f->nnodes;
    d->desc = (const uint32_t *)put(f->desc, (size_t)f->n * 32);
    d->node_id = (const uint32_t *)put(f->node_id, (size_t)f->nnodes * 4);
    if (drop_unflagged && f->nnodes) {
        int32_t *no = (int32_t *)(h + *off);
        d->node_off = (const int32_t *)put(nullptr, ((size_t)f->nnodes + 1) * 4);
        uint32_t *fo = (uint32_t *)(h + *off);
        d->feat = (const uint32_t *)put(nullptr, m * 4);
        int w = 0;
        for (int i = 0; i < f->nnodes; i++) {
            no[i] = w;
            for (int j = f->node_off[i]; j < f->node_off[i + 1]; j++)
                if (f->flag[f->feat[j]]) fo[w++] = f->feat[j];
        }
        no[f->nnodes] = w;
    } else {
        d->node_off = (const int32_t *)put(f->nnodes ? f->node_off : zero_off, ((size_t)f->nnodes + 1) * 4);
        d->feat = (const uint32_t *)put(f->feat, m * 4);
    }
    d->flag = (const uint8_t *)put(f->flag, (size_t)f->n);
    d->angle = (const float *)put(f->angle, (size_t)f->n * 4);
    {   // descriptors and flags once more, in list order (see k_bow phase 1)
        const uint32_t *list = (const uint32_t *)(h + ((const uint8_t *)d->feat - dbase));
        const int32_t *noff = (const int32_t *)(h + ((const uint8_t *)d->node_off - dbase));
        const size_t ml = f->nnodes ? (size_t)noff[f->nnodes] : 0;
        uint8_t *sd = h + *off;
        d->sdesc = (const uint32_t *)put(nullptr, m * 32);
        uint8_t *sf = h + *off;
        d->sflag = (const uint8_t *)put(nullptr, m);
        for (size_t k = 0; k < ml; k++) {
            memcpy(sd + 32 * k, f->desc + 32 * (size_t)list[k], 32);
            sf[k] = f->flag[list[k]];
        }
    }
    d->x = d->y = d->u_right = nullptr; d->octave = nullptr;
    if (geom) {
        d->x = (const float *)put(f->x, (size_t)f->n * 4);
        d->y = (const float *)put(f->y, (size_t)f->n * 4);
        d->u_right = (const float *)put(f->u_right, (size_t)f->n * 4);
        d->octave = (const int32_t *)put(f->octave, (size_t)f->n * 4);
    }
}

// The SearchByBoW kernels of this file behind the host-pointer entry points of orbx_match.hip: taken when the test hook forces a
// form, or when a vocabulary node is too large for the register form there.
int orbx_bow_run_legacy(int mode, int device, const orbx_featset *as, int na, const orbx_featset *b,
                        float nnratio, int check_ori, int32_t *match, int *nmatches)
{
    const int b_shared = 1;
    if (!as || !b || na < 1 || !match || !nmatches) { orbx_set_error("bow search: null argument"); return ORBX_E_INVALID; }
    size_t blob = a16(sizeof(DevFeat) * (size_t)(na + 1));
    int max_b = 0, max_slots = 0;
    for (int i = 0; i < na; i++) {
        if (!orbx_feat_validate(&as[i], 0, 1)) { orbx_set_error("bow search: malformed feature set %d", i); return ORBX_E_INVALID; }
        blob += feat_bytes(&as[i], 0);
        if (mode == 1 && as[i].n > max_slots) max_slots = as[i].n;
    }
    if (!orbx_feat_validate(b, 0, 1)) { orbx_set_error("bow search: malformed feature set"); return ORBX_E_INVALID; }
    blob += feat_bytes(b, 0);
    max_b = b->n;
    const int stride = mode == 0 ? b->n : max_slots;
    if (mode == 0) max_slots = b->n;
    BowCtx *c;
    int rc = bow_ctx(device, &c);
    if (rc) return rc;
    const size_t out_ints = (size_t)na * (stride > 0 ? stride : 1) + na;
    if ((rc = bow_reserve(c, blob, out_ints))) return rc;
    DevFeat *hd = (DevFeat *)c->h_blob;
    size_t off = a16(sizeof(DevFeat) * (size_t)(na + 1));
    for (int i = 0; i < na; i++) feat_pack(&as[i], 0, c->h_blob, c->d_blob, &off, &hd[i], true);
    feat_pack(b, 0, c->h_blob, c->d_blob, &off, &hd[na], mode == 1);
    ORBX_HIP(hipMemcpyAsync(c->d_blob, c->h_blob, off, hipMemcpyHostToDevice, c->stream));
    const DevFeat *dA = (const DevFeat *)c->d_blob, *dB = dA + na;
    int32_t *d_match = c->d_out;
    int *d_n = c->d_out + (size_t)na * (stride > 0 ? stride : 1);
    rc = mode == 0 ? bow_launch<0>(na, 1, max_b, max_slots, c->stream, dA, dB, b_shared, nnratio, check_ori, d_match, stride, d_n)
                   : bow_launch<1>(na, 1, max_b, max_slots, c->stream, dA, dB, b_shared, nnratio, check_ori, d_match, stride, d_n);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(c->h_out, c->d_out, out_ints * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    ORBX_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < na; i++) {
        const int cnt = mode == 0 ? b->n : as[i].n;
        memcpy(match + (size_t)i * stride, c->h_out + (size_t)i * stride, sizeof(int32_t) * cnt);
        nmatches[i] = c->h_out[(size_t)na * (stride > 0 ? stride : 1) + i];
    }
    return ORBX_OK;
}

// ---- device-resident keyframe set
struct orbx_bowdb {
    int device, nkf, max_n;
    uint8_t *d_blob;       // DevFeat[nkf] followed by the packed arrays
    hipStream_t stream;
    uint8_t *h_f; size_t h_f_cap;   // pinned staging of the frame side
    uint8_t *d_f; size_t d_f_cap;
    int32_t *d_out; int32_t *h_out; size_t out_cap;
};

extern "C" int orbx_bowdb_create(int device, const orbx_featset *kfs, int nkf, orbx_bowdb **out)
{
    if (!out || !kfs || nkf < 1) { orbx_set_error("orbx_bowdb_create: invalid argument"); return ORBX_E_INVALID; }
    *out = nullptr;
    size_t blob = a16(sizeof(DevFeat) * (size_t)nkf);
    for (int i = 0; i < nkf; i++) {
        if (!orbx_feat_validate(&kfs[i], 0, 1)) { orbx_set_error("orbx_bowdb_create: malformed feature set %d", i); return ORBX_E_INVALID; }
        blob += feat_bytes(&kfs[i], 0);
    }
    BowCtx *c;
    int rc = bow_ctx(device, &c); // validates the device
    if (rc) return rc;
    orbx_bowdb *db = new orbx_bowdb();
    memset(db, 0, sizeof *db);
    db->device = device; db->nkf = nkf;
    std::vector<uint8_t> h(blob);
    hipError_t e1 = hipMalloc((void **)&db->d_blob, blob);
    hipError_t e2 = hipStreamCreateWithFlags(&db->stream, hipStreamNonBlocking);
    if (e1 != hipSuccess || e2 != hipSuccess) { orbx_set_error("orbx_bowdb_create: HIP allocation failed"); orbx_bowdb_destroy(db); return ORBX_E_HIP; }
    DevFeat *hd = (DevFeat *)h.data();
    size_t off = a16(sizeof(DevFeat) * (size_t)nkf);
    for (int i = 0; i < nkf; i++) {
        feat_pack(&kfs[i], 0, h.data(), db->d_blob, &off, &hd[i], true);
        if (kfs[i].n > db->max_n) db->max_n = kfs[i].n;
    }
    if (hipMemcpy(db->d_blob, h.data(), off, hipMemcpyHostToDevice) != hipSuccess) { orbx_set_error("upload failed"); orbx_bowdb_destroy(db); return ORBX_E_HIP; }
    *out = db;
    return ORBX_OK;
}

extern "C" int orbx_bowdb_size(const orbx_bowdb *db) { return db ? db->nkf : ORBX_E_INVALID; }

extern "C" void orbx_bowdb_destroy(orbx_bowdb *db)
{
    if (!db) return;
    hipSetDevice(db->device);
    if (db->stream) { hipStreamSynchronize(db->stream); hipStreamDestroy(db->stream); }
    if (db->d_blob) hipFree(db->d_blob);
    if (db->d_f) hipFree(db->d_f);
    if (db->h_f) hipHostFree(db->h_f);
    if (db->d_out) hipFree(db->d_out);
    if (db->h_out) hipHostFree(db->h_out);
    delete db;
}

extern "C" int orbx_bowdb_search(orbx_bowdb *db, const orbx_featset *f, float nnratio, int check_orientation,
                                 int32_t *match_f, int *nmatches)
{
    if (!db || !f || !match_f || !nmatches) { orbx_set_error("orbx_bowdb_search: null argument"); return ORBX_E_INVALID; }
    if (!orbx_feat_validate(f, 0, 1)) { orbx_set_error("orbx_bowdb_search: malformed feature set"); return ORBX_E_INVALID; }
    ORBX_HIP(hipSetDevice(db->device));
    const size_t fb = a16(sizeof(DevFeat)) + feat_bytes(f, 0);
    if (fb > db->h_f_cap) {
        if (db->h_f) ORBX_HIP(hipHostFree(db->h_f));
        if (db->d_f) ORBX_HIP(hipFree(db->d_f));
        db->h_f = nullptr; db->d_f = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&db->h_f, fb * 2, hipHostMallocDefault));
        ORBX_HIP(hipMalloc((void **)&db->d_f, fb * 2));
        db->h_f_cap = db->d_f_cap = fb * 2;
    }
    const int stride = f->n > 0 ? f->n : 1;
    const size_t out_ints = (size_t)db->nkf * stride + db->nkf;
    if (out_ints > db->out_cap) {
        if (db->d_out) ORBX_HIP(hipFree(db->d_out));
        if (db->h_out) ORBX_HIP(hipHostFree(db->h_out));
        db->d_out = nullptr; db->h_out = nullptr;
        ORBX_HIP(hipMalloc((void **)&db->d_out, out_ints * 2 * sizeof(int32_t)));
        ORBX_HIP(hipHostMalloc((void **)&db->h_out, out_ints * 2 * sizeof(int32_t), hipHostMallocDefault));
        db->out_cap = out_ints * 2;
    }
    DevFeat *hd = (DevFeat *)db->h_f;
    size_t off = a16(sizeof(DevFeat));
    feat_pack(f, 0, db->h_f, db->d_f, &off, hd);
    ORBX_HIP(hipMemcpyAsync(db->d_f, db->h_f, off, hipMemcpyHostToDevice, db->stream));
    int32_t *d_match = db->d_out;
    int *d_n = db->d_out + (size_t)db->nkf * stride;
    {
        const int rc = bow_launch<0>(db->nkf, 1, f->n, f->n, db->stream, (const DevFeat *)db->d_blob, (const DevFeat *)db->d_f, 1, nnratio,
                                     check_orientation, d_match, f->n, d_n);
        if (rc) return rc;
    }
    ORBX_HIP(hipMemcpyAsync(db->h_out, db->d_out, out_ints * sizeof(int32_t), hipMemcpyDeviceToHost, db->stream));
    ORBX_HIP(hipStreamSynchronize(db->stream));
    if (f->n) memcpy(match_f, db->h_out, sizeof(int32_t) * (size_t)db->nkf * f->n);
    memcpy(nmatches, db->h_out + (size_t)db->nkf * stride, sizeof(int) * db->nkf);
    return ORBX_OK;
}

// Batched, device-resident relocalisation search: every keyframe of the set against every frame of an
// orbx_bow_frames batch (the output of orbx_bow_transform_batch_device) in one launch, nothing crosses PCIe.
extern "C" int orbx_bowdb_search_batch_device(orbx_bowdb *db, const orbx_bow_frames *fr, int batch, float nnratio,
                                              int check_orientation, void *d_match, void *d_nmatches, void *stream)
{
    if (!db || !fr || !d_match || !d_nmatches || batch < 1 || batch > fr->batch || fr->device != db->device) {
        orbx_set_error("orbx_bowdb_search_batch_device: invalid argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(hipSetDevice(db->device));
    const int rc = bow_launch<0>(db->nkf, batch, fr->cap, fr->cap, stream ? (hipStream_t)stream : fr->last_stream, (const DevFeat *)db->d_blob,
                                 (const DevFeat *)fr->d_feats, 1, nnratio, check_orientation, (int32_t *)d_match, fr->cap, (int *)d_nmatches);
    if (rc) return rc;
    return ORBX_OK;
}

// The same search with the result in the form its consumer reads: Tracking::Relocalization hands every candidate's matches to a PnP solver
// that walks the non-null entries of vvpMapPointMatches[i] (src/Tracking.cc:1682-1693).  Per (frame, keyframe): the number of matches and
// the first min(count, cap_pairs) (frame feature, keyframe feature) pairs in frame-feature order -- 0.8 KB written per pair where the dense
// row is 4 KB (500 keyframes x 32 frames: 13 MB instead of 66 MB).
extern "C" int orbx_bowdb_search_batch_device_compact(orbx_bowdb *db, const orbx_bow_frames *fr, int batch, float nnratio, int check_orientation,
                                                      void *d_pairs, int cap_pairs, void *d_nmatches, void *stream)
{
    if (!db || !fr || !d_pairs || !d_nmatches || batch < 1 || batch > fr->batch || fr->device != db->device || cap_pairs < 1) {
        orbx_set_error("orbx_bowdb_search_batch_device_compact: invalid argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(hipSetDevice(db->device));
    return bow_launch<0>(db->nkf, batch, fr->cap, fr->cap, stream ? (hipStream_t)stream : fr->last_stream, (const DevFeat *)db->d_blob,
                         (const DevFeat *)fr->d_feats, 1, nnratio, check_orientation, (int32_t *)d_pairs, cap_pairs, (int *)d_nmatches, 1);
}

// ---------------------------------------------------------------- MapPoint::ComputeDistinctiveDescriptors (f3)
// One wave per map point (src/MapPoint.cc:266-340): the N x N Hamming matrix goes to LDS (u16), every
// lane then takes rows and finds the row median vDists[int(0.5*(N-1))] by bisection on the value (the k-th
// smallest of a row is the smallest v with #(d <= v) > k), and a wave min of (median<<16 | row) gives the
// first row with the smallest median.
__global__ __launch_bounds__(64) void k_distinctive(const uint32_t *__restrict__ desc, const int32_t *__restrict__ off,
                                                    int32_t *__restrict__ best_idx)
{
    extern __shared__ __align__(16) unsigned char dd_smem[];
    uint16_t *dm = reinterpret_cast<uint16_t *>(dd_smem);
    const int p = blockIdx.x, lane = threadIdx.x;
    const int o = off[p], n = off[p + 1] - o;
    if (n <= 0) { if (lane == 0) best_idx[p] = -1; return; }
    for (int i = lane; i < n; i += 64) {
        uint32_t a[8];
        load_desc(desc, o + i, a);
        for (int j = 0; j < n; j++) {
            uint32_t b[8];
            load_desc(desc, o + j, b);
            dm[i * n + j] = (uint16_t)hamming256(a, b);
        }
    }
    __syncthreads();
    const int k = (int)(0.5 * (n - 1));
    unsigned best = 0xFFFFFFFFu;
    for (int i = lane; i < n; i += 64) {
        int lo = 0, hi = 256; // smallest v with count(d <= v) > k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < n; j++) c += dm[i * n + j] <= mid;
            if (c > k) hi = mid; else lo = mid + 1;
        }
        const unsigned key = ((unsigned)lo << 16) | (unsigned)i;
        best = key < best ? key : best;
    }
    best = wave_min_u32(best);
    if (lane == 0) best_idx[p] = (int)(best & 0xFFFF);
}

extern "C" int orbx_distinctive_descriptors(int device, const uint8_t *desc, const int32_t *off, int npoints, int32_t *best_idx)
{
    if (!off || !best_idx || npoints < 0 || (npoints && off[npoints] > 0 && !desc)) { orbx_set_error("orbx_distinctive_descriptors: invalid argument"); return ORBX_E_INVALID; }
    if (npoints == 0) return ORBX_OK;
    int max_n = 0;
    if (off[0] != 0) { orbx_set_error("off[0] must be 0"); return ORBX_E_INVALID; }
    for (int p = 0; p < npoints; p++) {
        const int n = off[p + 1] - off[p];
        if (n < 0) { orbx_set_error("offsets must be non-decreasing"); return ORBX_E_INVALID; }
        if (n > 256) { orbx_set_error("map point %d has %d observations (limit 256)", p, n); return ORBX_E_INVALID; }
        if (n > max_n) max_n = n;
    }
    BowCtx *c;
    int rc = bow_ctx(device, &c);
    if (rc) return rc;
    const size_t total = (size_t)off[npoints];
    const size_t b_desc = a16(total * 32), b_off = a16(sizeof(int32_t) * ((size_t)npoints + 1));
    if ((rc = bow_reserve(c, b_desc + b_off, (size_t)npoints + 4))) return rc;
    if (total) memcpy(c->h_blob, desc, total * 32);
    memcpy(c->h_blob + b_desc, off, sizeof(int32_t) * ((size_t)npoints + 1));
    ORBX_HIP(hipMemcpyAsync(c->d_blob, c->h_blob, b_desc + b_off, hipMemcpyHostToDevice, c->stream));
    const size_t lds = (size_t)max_n * max_n * 2 + 16;
    ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_distinctive), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_distinctive, dim3(npoints), dim3(64), lds, c->stream, (const uint32_t *)c->d_blob,
                       (const int32_t *)(c->d_blob + b_desc), c->d_out);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(c->h_out, c->d_out, sizeof(int32_t) * (size_t)npoints, hipMemcpyDeviceToHost, c->stream));
    ORBX_HIP(hipStreamSynchronize(c->stream));
    memcpy(best_idx, c->h_out, sizeof(int32_t) * (size_t)npoints);
    return ORBX_OK;
}
