// orbx_proj.hip — the two per-frame projection-guided matchers of ORB-SLAM2's tracking thread
// (SURVEY.md 8f row f1):
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)        reference src/ORBmatcher.cc:1396-1553
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   reference src/ORBmatcher.cc:48-129
// with Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (src/Frame.cc:261-279, :386-457).
//
// The reference walks the map points sequentially and lets every accepted match claim its feature
// (later points skip features whose holder has Observations() > 0), so the result depends on the order.
// Exact parallel form: the per-point candidate lists (window, level, uRight tests and Hamming distances) do
// not depend on the claims and are built in parallel, in GetFeaturesInArea's traversal order; the claims are
// then the unique fixpoint of
//     choice(i) = best candidate f of i with  !occupied(f)  and no j < i, has_obs(j), choice(j) == f,
// a recursion on i, reached by iterating all points in parallel until nothing changes (at most one round
// per link of the longest conflict chain).  Projections (u, v, 1/z) come from the adaptor, which has the
// poses: restating cv::gemm's accumulation is not needed.
#include "orbx_device.h"
#include <string.h>
#include <vector>

#define PG_COLS 64 // FRAME_GRID_COLS, include/Frame.h:38
#define PG_ROWS 48 // FRAME_GRID_ROWS, include/Frame.h:37
#define PG_CELLS (PG_COLS * PG_ROWS)

struct DevFrame {
    int n;
    const float *x, *y, *angle, *u_right;
    const int32_t *octave;
    const uint32_t *desc;
    const uint8_t *occupied;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
};
struct DevPoints {
    int n;
    const float *u, *v, *aux, *angle, *view_cos;
    const int32_t *level;
    const uint32_t *desc;
    const uint8_t *valid, *has_obs;
};
// One parameter block describes every projection-type search of src/ORBmatcher.cc (which tests, which window, which
// claim rule); the extern "C" entry points below fill it per reference function.
struct ProjParams {
    int radius_mode;   // 0: th * sf[level]   1: RadiusByViewingCos(view_cos) [* th] * sf[level]   2: th (a fixed window)
    int bounds;        // 0: none  1: Frame bounds, inclusive (:1431-1434)  2: KeyFrame::IsInImage (src/KeyFrame.cc:649-652)
    int need_pos_aux;  // 1: reject aux (= invzc) < 0 (:1426)
    int lo_off, hi_off;// candidate levels [level + lo_off, level + hi_off] unless direction != 0
    int direction;     // last-frame search: 1 bForward, 2 bBackward
    int ur_mode;       // 0: none  1: |u - mbf*aux - uR| > r (:1467-1471)  2: |aux - uR| > r (:88-92)
    int chi2;          // Fuse: reprojection gate 5.99 / 7.8 with inv_sigma2[octave], aux = ur (:967-992)
    int max_dist;      // TH_HIGH, TH_LOW or ORBdist
    int ratio;         // 1: same-level ratio test of the map-point search (:117-121)
    int check_ori;
    int mark_cleared;  // 1: a feature whose match the rotation filter cleared reads -2 instead of -1 (the reference leaves NULL there, not the old holder)
    int init_search;   // 1: SearchForInitialization's sequential rule (k_init_resolve)
    int claims;        // 0: points are independent (Fuse, SearchBySim3)  1: an accepted match of a point with
                       // Observations() > 0 blocks its feature  2: every accepted match blocks it
    float th, mbf, nnratio;
    float sf[ORBX_MAX_LEVELS], inv_sigma2[ORBX_MAX_LEVELS];
};

#define PG_LDS_FEATS 8192
// ---- Frame::AssignFeaturesToGrid: CSR over the 64x48 cells, ascending feature index inside a cell
__global__ __launch_bounds__(256) void k_grid_build(DevFrame F, int *__restrict__ cell_off, int *__restrict__ cell_idx)
{
    __shared__ int cnt[PG_CELLS];
    __shared__ int cur[PG_CELLS];
    __shared__ int s_w[4];
    __shared__ uint16_t idx_l[PG_LDS_FEATS];     // the cell lists while they are being ordered (frames of up to PG_LDS_FEATS features)
    const int tid = threadIdx.x;
    const bool in_lds = F.n <= PG_LDS_FEATS;
    for (int c = tid; c < PG_CELLS; c += 256) { cnt[c] = 0; cur[c] = 0; }
    __syncthreads();
    for (int i = tid; i < F.n; i += 256) {
        const int px = (int)roundf((F.x[i] - F.min_x) * F.inv_w), py = (int)roundf((F.y[i] - F.min_y) * F.inv_h); // PosInGrid :444-457
        if (px >= 0 && px < PG_COLS && py >= 0 && py < PG_ROWS) atomicAdd(&cnt[px * PG_ROWS + py], 1);
    }
    __syncthreads();
    const int total = lds_excl_scan(cnt, PG_CELLS, s_w);
    for (int c = tid; c < PG_CELLS; c += 256) cell_off[c] = cnt[c];
    if (tid == 0) cell_off[PG_CELLS] = total;
    for (int i = tid; i < F.n; i += 256) {
        const int px = (int)roundf((F.x[i] - F.min_x) * F.inv_w), py = (int)roundf((F.y[i] - F.min_y) * F.inv_h);
        if (px >= 0 && px < PG_COLS && py >= 0 && py < PG_ROWS) {
            const int c = px * PG_ROWS + py;
            const int slot = cnt[c] + atomicAdd(&cur[c], 1);
            if (in_lds) idx_l[slot] = (uint16_t)i; else cell_idx[slot] = i;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (in_lds) {
        // push_back order = ascending i: insertion sort of the (short) cell lists in LDS, then ONE coalesced copy to memory (sorting them in
        // global memory was a chain of dependent loads and stores per cell with two entries or more: half of this kernel's 10.8 us)
        for (int c = tid; c < PG_CELLS; c += 256) {
            const int b = cnt[c], e = b + cur[c];
            for (int a = b + 1; a < e; a++) {
                const uint16_t v = idx_l[a];
                int j = a - 1;
                while (j >= b && idx_l[j] > v) { idx_l[j + 1] = idx_l[j]; j--; }
                idx_l[j + 1] = v;
            }
        }
        __syncthreads();
        for (int j = tid; j < total; j += 256) cell_idx[j] = idx_l[j];
        return;
    }
    for (int c = tid; c < PG_CELLS; c += 256) { // (larger frames: the same in global memory)
        const int b = cnt[c], e = b + cur[c];
        for (int a = b + 1; a < e; a++) {
            const int v = cell_idx[a];
            int j = a - 1;
            while (j >= b && cell_idx[j] > v) { cell_idx[j + 1] = cell_idx[j]; j--; }
            cell_idx[j + 1] = v;
        }
    }
}

// window of a point: radius, level range, cell range; false if the point takes no part
struct Win { float u, v, r; int min_l, max_l, cx0, cx1, cy0, cy1; };

__device__ __forceinline__ bool point_window(const DevFrame &F, const DevPoints &P, const ProjParams &pp, int i, Win *w)
{
    if (!P.valid[i]) return false;
    const float u = P.u[i], v = P.v[i];
    const int lvl = P.level[i] & (ORBX_MAX_LEVELS - 1);
    float r;
    int min_l, max_l;
    if (pp.need_pos_aux && P.aux[i] < 0) return false;                 // invzc < 0 (:1426)
    if (pp.bounds == 1) { if (u < F.min_x || u > F.max_x || v < F.min_y || v > F.max_y) return false; } // :1431-1434
    else if (pp.bounds == 2) { if (!(u >= F.min_x && u < F.max_x && v >= F.min_y && v < F.max_y)) return false; }
    if (pp.radius_mode == 0) {
        r = pp.th * pp.sf[lvl];                                        // :1439
    } else if (pp.radius_mode == 2) {
        r = pp.th;                                                     // SearchForInitialization's windowSize (:461)
    } else {
        float rr = (double)P.view_cos[i] > 0.998 ? 2.5f : 4.0f;        // RadiusByViewingCos (:131-137)
        if ((double)pp.th != 1.0) rr *= pp.th;                         // bFactor (:52, :66-67)
        r = rr * pp.sf[lvl];
    }
    if (pp.direction == 1) { min_l = lvl; max_l = -1; }                // bForward  (:1443)
    else if (pp.direction == 2) { min_l = 0; max_l = lvl; }            // bBackward (:1445)
    else { min_l = lvl + pp.lo_off; max_l = lvl + pp.hi_off; }
    // GetFeaturesInArea cell range (src/Frame.cc:391-406)
    const int a = (int)floorf((u - F.min_x - r) * F.inv_w);
    const int cx0 = a > 0 ? a : 0;
    if (cx0 >= PG_COLS) return false;
    int cx1 = (int)ceilf((u - F.min_x + r) * F.inv_w);
    cx1 = cx1 < PG_COLS - 1 ? cx1 : PG_COLS - 1;
    if (cx1 < 0) return false;
    const int b = (int)floorf((v - F.min_y - r) * F.inv_h);
    const int cy0 = b > 0 ? b : 0;
    if (cy0 >= PG_ROWS) return false;
    int cy1 = (int)ceilf((v - F.min_y + r) * F.inv_h);
    cy1 = cy1 < PG_ROWS - 1 ? cy1 : PG_ROWS - 1;
    if (cy1 < 0) return false;
    w->u = u; w->v = v; w->r = r; w->min_l = min_l; w->max_l = max_l;
    w->cx0 = cx0; w->cx1 = cx1; w->cy0 = cy0; w->cy1 = cy1;
    return true;
}

// claim-independent part of the candidate test: level range, box, right-image coordinate
__device__ __forceinline__ bool cand_ok(const DevFrame &F, const DevPoints &P, const ProjParams &pp, const Win &w, int i, int k)
{
    if (pp.claims && F.occupied[k]) return false;   // the feature holds an observed map point: skipped by every point (:80-82, :1453-1455); filtered here, once, not in every round of the resolve kernel
    const int oct = F.octave[k];
    if (w.min_l > 0 || w.max_l >= 0) { // bCheckLevels (:408)
        if (oct < w.min_l) return false;
        if (w.max_l >= 0 && oct > w.max_l) return false;
    }
    const float distx = F.x[k] - w.u, disty = F.y[k] - w.v;
    if (!(fabsf(distx) < w.r && fabsf(disty) < w.r)) return false; // :434
    const float ur_k = F.u_right[k];
    if (pp.ur_mode && ur_k > 0) {
        if (pp.ur_mode == 1) {
            const float ur = w.u - pp.mbf * P.aux[i];                  // :1467-1471
            if (fabsf(ur - ur_k) > w.r) return false;
        } else {
            if (fabsf(P.aux[i] - ur_k) > w.r) return false;            // :88-92 (r * scale == w.r)
        }
    }
    if (pp.chi2) {                                                     // Fuse, :961-992 (float products, double compare)
        const float inv = pp.inv_sigma2[oct & (ORBX_MAX_LEVELS - 1)];
        const float ex = -distx, ey = -disty;
        if (ur_k >= 0) {
            const float er = P.aux[i] - ur_k;
            const float e2 = ex * ex + ey * ey + er * er;
            if ((double)(e2 * inv) > 7.8) return false;
        } else {
            const float e2 = ex * ex + ey * ey;
            if ((double)(e2 * inv) > 5.99) return false;
        }
    }
    return true;
}

// Candidate list of every point, one wave per point.  For a fixed grid column ix the cells (ix, cy0..cy1) are
// consecutive in the CSR, so the window is a handful of contiguous index ranges whose concatenation is exactly
// GetFeaturesInArea's traversal order (ix, iy, position in cell): lanes stride over a range, test the candidate and
// compute the Hamming distance, and a ballot keeps the order when the survivors are appended
// (entry = feature | dist<<16 | octave<<25).  The wave reserves the window's total range length in the pool with
// one atomicAdd (an upper bound of its list); beg[i] / cnt[i] locate the list.  If the pool overflows the host
// grows it and repeats the call (pool_used = entries needed).
__global__ __launch_bounds__(256) void k_proj_lists(DevFrame F, DevPoints P, ProjParams pp, const int *__restrict__ cell_off,
                                                    const int *__restrict__ cell_idx, int *__restrict__ beg, int *__restrict__ cnt,
                                                    uint32_t *__restrict__ entries, int pool_cap, int *__restrict__ pool_used)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= P.n) return; // wave-uniform
    Win w;
    int n = 0, base = 0;
    if (point_window(F, P, pp, i, &w)) {
        // The window's cells are, per grid column, one contiguous run of the CSR (cells are stored column by column): lane q fetches the run of
        // column cx0 + q (the grid has 64 columns), a wave prefix sum flattens the runs, and the candidates are then taken 64 at a time ACROSS
        // columns -- in the order the column-by-column walk had (the list order decides ties downstream).  That walk was four dependent memory
        // round trips per column; this is four per point.
        const int ncol = w.cx1 - w.cx0 + 1;
        int c_lo = 0, c_n = 0;
        if (lane < ncol) {
            const int ix = w.cx0 + lane;
            c_lo = cell_off[ix * PG_ROWS + w.cy0];
            c_n = cell_off[ix * PG_ROWS + w.cy1 + 1] - c_lo;
        }
        const int c_incl = wave_incl_scan(c_n), c_excl = c_incl - c_n;
        const int upper = __builtin_amdgcn_readlane(c_incl, 63);
        if (upper) {
            if (lane == 0) base = atomicAdd(pool_used, upper);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        if (upper && base + upper <= pool_cap) {
            uint32_t d[8];
            const uint4 *s = reinterpret_cast<const uint4 *>(P.desc + (long long)i * 8);
            const uint4 q0 = s[0], q1 = s[1];
            d[0] = q0.x; d[1] = q0.y; d[2] = q0.z; d[3] = q0.w; d[4] = q1.x; d[5] = q1.y; d[6] = q1.z; d[7] = q1.w;
            {
                for (int jb = 0; jb < upper; jb += 64) {
                    const int jf = jb + lane;            // position in the flattened candidate sequence
                    int j = -1;
                    for (int q = 0; q < ncol; q++) {     // which column's run holds it (a handful of columns; their runs ride in lanes 0 .. ncol - 1)
                        const int e_ = __builtin_amdgcn_readlane(c_excl, q), n_ = __builtin_amdgcn_readlane(c_n, q), l_ = __builtin_amdgcn_readlane(c_lo, q);
                        if (jf >= e_ && jf < e_ + n_) j = l_ + (jf - e_);
                    }
                    bool ok = false;
                    uint32_t en = 0;
                    if (jf < upper) {
                        const int k = cell_idx[j];
                        if (cand_ok(F, P, pp, w, i, k)) {
                            ok = true;
                            const uint4 *t = reinterpret_cast<const uint4 *>(F.desc + (long long)k * 8);
                            const uint4 v0 = t[0], v1 = t[1];
                            const int dist = __popc(d[0] ^ v0.x) + __popc(d[1] ^ v0.y) + __popc(d[2] ^ v0.z) + __popc(d[3] ^ v0.w) +
                                             __popc(d[4] ^ v1.x) + __popc(d[5] ^ v1.y) + __popc(d[6] ^ v1.z) + __popc(d[7] ^ v1.w);
                            en = (uint32_t)k | ((uint32_t)dist << 16) | ((uint32_t)(F.octave[k] & 31) << 25);
                        }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) entries[base + n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = en;
                    n += __popcll(m);
                }
            }
        } else if (upper) {
            n = 0; // overflow: the host repeats the call with a larger pool
        }
    }
    if (lane == 0) { beg[i] = base; cnt[i] = n; }
}

// the claim fixpoint + output, one workgroup.  owner[] (the earliest observed point currently choosing each
// feature) lives in LDS; points below the first index that changed in a round are final and are not evaluated again.
extern __shared__ __align__(16) int resolve_smem[];

__global__ __launch_bounds__(1024) void k_proj_resolve(DevFrame F, DevPoints P, ProjParams pp, const int *__restrict__ beg,
                                                       const int *__restrict__ cnt, const uint32_t *__restrict__ entries,
                                                       int *__restrict__ choice_a, int *__restrict__ choice_b,
                                                       int32_t *__restrict__ match, int *__restrict__ out_n,
                                                       int32_t *__restrict__ pt_choice, int32_t *__restrict__ pt_dist)
{
    __shared__ int s_first, s_cnt;
    __shared__ int hist[30];
    __shared__ int keep3[3];
    int *owner = resolve_smem; // [F.n]
    const int tid = threadIdx.x, nt = 1024;
    int *cur = choice_a, *nxt = choice_b;
    for (int i = tid; i < P.n; i += nt) { cur[i] = -1; nxt[i] = -1; }
    __threadfence_block();
    __syncthreads();
    int stable = 0; // points [0, stable) are final
    for (int round = 0; round <= P.n; round++) {
        for (int f = tid; f < F.n; f += nt) owner[f] = 0x7FFFFFFF;
        if (tid == 0) s_first = 0x7FFFFFFF;
        __syncthreads();
        for (int i = tid; i < P.n; i += nt)
            if (cur[i] >= 0 && pp.claims && (pp.claims == 2 || P.has_obs[i])) atomicMin(&owner[cur[i]], i);
        __syncthreads();
        int first = 0x7FFFFFFF;
        for (int i = stable + tid; i < P.n; i += nt) {
            int b1 = 256, b2 = 256, l1 = -1, l2 = -1, bi = -1;
            const int e0 = beg[i], e1 = e0 + cnt[i];
            for (int eb = e0; eb < e1; eb += 8) {      // eight entries per trip: their loads travel together (one by one the walk was a memory round trip per entry and round)
                uint32_t en8[8];
#pragma unroll
                for (int q = 0; q < 8; q++) en8[q] = eb + q < e1 ? entries[eb + q] : 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t en = en8[q];
                    if (eb + q >= e1) break;
                    const int f = en & 0xFFFF, dist = (en >> 16) & 0x1FF, lv = en >> 25;
                    if (pp.claims && owner[f] < i) continue; // the feature is held by an earlier point (see ProjParams::claims; occupied features never enter the lists)
                    if (dist < b1) { b2 = b1; l2 = l1; b1 = dist; l1 = lv; bi = f; }
                    else if (dist < b2) { b2 = dist; l2 = lv; }
                }
            }
            int c = -1;
            if (b1 <= pp.max_dist) {
                if (!pp.ratio) c = bi;
                else if (!(l1 == l2 && (float)b1 > pp.nnratio * (float)b2)) c = bi; // :117-121
            }
            nxt[i] = c;
            pt_dist[i] = c >= 0 ? b1 : 256; // final once the point is stable: an unchanged choice keeps its distance
            if (c != cur[i] && i < first) first = i;
        }
        if (first != 0x7FFFFFFF) atomicMin(&s_first, first);
        for (int i = tid; i < stable; i += nt) nxt[i] = cur[i]; // the stable prefix keeps its choices in both buffers
        __threadfence_block();
        __syncthreads();
        const int fc = s_first;
        { int *t = cur; cur = nxt; nxt = t; }
        __syncthreads();
        if (fc == 0x7FFFFFFF) break;
        stable = fc; // nothing below the first change moved: those points depend only on earlier ones and are final
    }
    // ---- outputs: a feature ends up with the LAST point that chose it (:1488 overwrites); every choice counts
    for (int f = tid; f < F.n; f += nt) match[f] = -1;
    if (tid < 30) hist[tid] = 0;
    if (tid == 0) s_cnt = 0;
    __threadfence_block();
    __syncthreads();
    int local = 0;
    for (int i = tid; i < P.n; i += nt) {
        const int f = cur[i];
        pt_choice[i] = f;
        if (f < 0) continue;
        atomicMax(&match[f], i);
        local++;
        if (pp.check_ori) {
            float rot = P.angle[i] - F.angle[f];               // :1493-1500
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / 30));
            if (bin == 30) bin = 0;
            bin = (unsigned)bin < 30u ? bin : 0;
            atomicAdd(&hist[bin], 1);
            nxt[i] = bin;
        }
    }
    if (local) atomicAdd(&s_cnt, local);
    __threadfence_block();
    __syncthreads();
    if (pp.check_ori) {
        if (tid == 0) { // ComputeThreeMaxima (:1687-1728)
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; i++) {
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
        int removed = 0;
        for (int i = tid; i < P.n; i += nt) {
            const int f = cur[i];
            if (f < 0) continue;
            const int b = nxt[i];
            if (b != keep3[0] && b != keep3[1] && b != keep3[2]) { match[f] = pp.mark_cleared ? -2 : -1; pt_choice[i] = -1; removed++; } // :1518-1523, one decrement per entry
        }
        if (removed) atomicSub(&s_cnt, removed);
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) *out_n = s_cnt;
}

// ORBmatcher::SearchForInitialization's matching loop (src/ORBmatcher.cc:447-511) is sequential by construction: a
// candidate is skipped when the match it already holds is at least as good (vMatchedDistance, :470) and an accepted
// match steals the feature from its previous holder (:491-496).  The candidate lists with their distances come from
// k_proj_lists (all pairs in parallel); this kernel replays the decisions in order with ONE wave: per point a
// coalesced sweep over its list, the two smallest distances and the first position of the minimum by wave
// reductions, then the scalar bookkeeping.  md / m21 (vMatchedDistance, vnMatches21) live in LDS.
__global__ __launch_bounds__(64) void k_init_resolve(DevFrame F, DevPoints P, ProjParams pp, const int *__restrict__ beg,
                                                     const int *__restrict__ cnt, const uint32_t *__restrict__ entries,
                                                     int *__restrict__ bins, int32_t *__restrict__ pt_choice,
                                                     int32_t *__restrict__ pt_dist, int *__restrict__ out_n)
{
    uint16_t *md = reinterpret_cast<uint16_t *>(resolve_smem);          // [F.n] 0xFFFF = INT_MAX
    uint16_t *m21 = md + ((F.n + 7) & ~7);                               // [F.n] holder + 1, 0 = none
    float *fang = reinterpret_cast<float *>(m21 + ((F.n + 7) & ~7));     // [F.n] the frame's keypoint angles (read per accepted match)
    __shared__ int hist[30];
    const int lane = threadIdx.x;
    for (int f = lane; f < F.n; f += 64) { md[f] = 0xFFFF; m21[f] = 0; fang[f] = pp.check_ori ? F.angle[f] : 0.f; }
    for (int i = lane; i < P.n; i += 64) { pt_choice[i] = -1; pt_dist[i] = 256; bins[i] = -1; }
    if (lane < 30) hist[lane] = 0;
    __threadfence_block();
    __syncthreads();
    int nm = 0;
    // The walk is sequential, its memory traffic need not be: list heads (cnt, beg) are fetched 64 points at a time and handed out by
    // v_readlane, and while point j is decided the first two 64-entry chunks of the NEXT point with a list are already on their way
    // (a point's list is ~80 entries at 1000-2000 features and the reference's window of 100 px).  The per-point chain is then two wave
    // minima, the LDS look-ups and lane 0's bookkeeping -- not three dependent global round trips.  Only LDS state (md, m21, hist) orders
    // the points; the global result arrays are written as we go and read back after the walk, behind one fence.
    for (int base = 0; base < P.n; base += 64) {
        const int ip = base + lane;
        const int my_cnt = ip < P.n ? cnt[ip] : 0, my_beg = ip < P.n ? beg[ip] : 0;
        const float my_ang = ip < P.n && pp.check_ori ? P.angle[ip] : 0.f;
        unsigned long long todo = __ballot(my_cnt > 0);
        uint32_t nx0 = 0, nx1 = 0;                              // chunks 0 and 1 of the next point's list
        if (todo) {
            const int j = (int)__builtin_ctzll(todo);
            const int c = __builtin_amdgcn_readlane(my_cnt, j), e0 = __builtin_amdgcn_readlane(my_beg, j);
            if (lane < c) nx0 = entries[e0 + lane];
            if (64 + lane < c) nx1 = entries[e0 + 64 + lane];
        }
        while (todo) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int i1 = base + j;
            const int c = __builtin_amdgcn_readlane(my_cnt, j), e0 = __builtin_amdgcn_readlane(my_beg, j);
            const uint32_t en0 = nx0, en1 = nx1;
            if (todo) {                                         // the next point's chunks leave now
                const int jn = (int)__builtin_ctzll(todo);
                const int cn = __builtin_amdgcn_readlane(my_cnt, jn), en_ = __builtin_amdgcn_readlane(my_beg, jn);
                nx0 = lane < cn ? entries[en_ + lane] : 0u;
                nx1 = 64 + lane < cn ? entries[en_ + 64 + lane] : 0u;
            }
            unsigned k1 = 0xFFFFFFFFu, d2 = 0x7FFFFFFFu; // lane-local: smallest (dist<<16 | position), second smallest distance
            for (int eb = 0; eb < c; eb += 64) {
                const int e = eb + lane;
                if (e < c) {
                    const uint32_t en = eb == 0 ? en0 : eb == 64 ? en1 : entries[e0 + e];
                    const unsigned f = en & 0xFFFF, d = (en >> 16) & 0x1FF;
                    if (!(md[f] <= d)) {                                   // :470
                        const unsigned key = (d << 16) | (unsigned)e;     // lists are shorter than 65536 (host check)
                        if (key < k1) { d2 = k1 >> 16; k1 = key; }
                        else if (d < d2) d2 = d;
                    }
                }
            }
            if (k1 == 0xFFFFFFFFu) d2 = 0x7FFFFFFFu; else if (d2 == 0xFFFFu) d2 = 0x7FFFFFFFu;
            const unsigned best = wave_min_u32(k1);
            // second smallest over the wave: the winner lane contributes its own second, every other lane its first
            const unsigned mine = (k1 == best) ? d2 : (k1 == 0xFFFFFFFFu ? 0x7FFFFFFFu : (k1 >> 16));
            const unsigned second = wave_min_u32(mine);
            if (best == 0xFFFFFFFFu) continue;
            const int bd = (int)(best >> 16), be = (int)(best & 0xFFFF);
            if (bd <= 50 && (float)bd < (float)(int)second * pp.nnratio) {  // TH_LOW (:485), ratio (:487)
                // the winning entry sits in a register of lane be % 64 (chunks 0 / 1) or, beyond them, in memory
                const uint32_t wen = be < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)en0, be) : be < 128 ? (uint32_t)__builtin_amdgcn_readlane((int)en1, be - 64) : entries[e0 + be];
                const int f = (int)(wen & 0xFFFF);
                if (lane == 0) {
                    const int prev = (int)m21[f] - 1;
                    if (prev >= 0) pt_choice[prev] = -1;                   // :489-493
                    pt_choice[i1] = f; pt_dist[i1] = bd;
                    m21[f] = (uint16_t)(i1 + 1); md[f] = (uint16_t)bd;
                    if (pp.check_ori) {
                        float rot = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(my_ang), j)) - fang[f];   // :501-509
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * (1.0f / 30));
                        if (bin == 30) bin = 0;
                        bin = (unsigned)bin < 30u ? bin : 0;
                        hist[bin]++; bins[i1] = bin;
                    }
                }
                // lane 0's LDS writes before the next point's LDS reads: one wave, in-order LDS -- a compiler barrier is all it takes
                // (the workgroup fence that stood here also waited for the global stores: a memory round trip per accepted match)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // nmatches = matches still standing (every steal removed one, :492), then the orientation filter (:514-541)
    int i1k = -1, i2k = -1, i3k = -1;
    if (pp.check_ori) {
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int sv = hist[i];
            if (sv > max1) { max3 = max2; max2 = max1; max1 = sv; i3k = i2k; i2k = i1k; i1k = i; }
            else if (sv > max2) { max3 = max2; max2 = sv; i3k = i2k; i2k = i; }
            else if (sv > max3) { max3 = sv; i3k = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2k = -1; i3k = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { i3k = -1; }
    }
    int local = 0;
    for (int i = lane; i < P.n; i += 64) {
        if (pt_choice[i] < 0) continue;
        if (pp.check_ori) {
            const int b = bins[i];
            if (b != i1k && b != i2k && b != i3k) { pt_choice[i] = -1; continue; }
        }
        local++;
    }
    nm = wave_sum(local);
    if (lane == 0) *out_n = nm;
}

// ---------------------------------------------------------------- host side

struct ProjCtx {
    hipStream_t stream = nullptr;
    uint8_t *h_blob = nullptr, *d_blob = nullptr; size_t cap = 0;
    uint8_t *d_work = nullptr; size_t work_cap = 0;
    uint32_t *d_entries = nullptr; size_t ent_cap = 0;
    int32_t *h_out = nullptr; size_t out_cap = 0;
};
static thread_local ProjCtx g_proj[16];

static size_t pa16(size_t v) { return (v + 15) & ~(size_t)15; }

static int proj_run(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts, const float *sf, int nlevels,
                    const ProjParams &pp_in, int32_t *match_cur, int *nmatches, const float *inv_sigma2 = nullptr,
                    int32_t *pt_choice = nullptr, int32_t *pt_dist = nullptr)
{
    int nm_dummy = 0;
    std::vector<int32_t> mc_dummy;
    if (!nmatches) nmatches = &nm_dummy;
    if (!match_cur && cur && cur->n >= 0) { mc_dummy.resize((size_t)cur->n + 1); match_cur = mc_dummy.data(); }
    if (!cur || !pts || !sf || !match_cur || !nmatches || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || cur->n < 0 || pts->n < 0 ||
        cur->n >= 65536 || pts->n > (1 << 20)) {
        orbx_set_error("search_by_projection: invalid argument");
        return ORBX_E_INVALID;
    }
    const bool need_aux = pp_in.need_pos_aux || pp_in.ur_mode || pp_in.chi2;
    if (cur->n && (!cur->x || !cur->y || !cur->octave || !cur->u_right || !cur->desc || (pp_in.check_ori && !cur->angle) ||
                   (pp_in.claims && !cur->occupied))) { orbx_set_error("frame arrays missing"); return ORBX_E_INVALID; }
    if (pts->n && (!pts->u || !pts->v || !pts->level || !pts->desc || !pts->valid || (need_aux && !pts->aux) ||
                   (pp_in.claims == 1 && !pts->has_obs) || (pp_in.check_ori && !pts->angle) ||
                   (pp_in.radius_mode == 1 && !pts->view_cos))) { orbx_set_error("point arrays missing"); return ORBX_E_INVALID; }
    if (pp_in.chi2 && !inv_sigma2) { orbx_set_error("inv_sigma2 missing"); return ORBX_E_INVALID; }
    if (!(cur->max_x > cur->min_x) || !(cur->max_y > cur->min_y)) { orbx_set_error("empty image bounds"); return ORBX_E_INVALID; }
    for (int i = 0; i < pts->n; i++)
        if (pts->valid[i] && (pts->level[i] < 0 || pts->level[i] >= nlevels)) { orbx_set_error("point %d: level %d out of range", i, pts->level[i]); return ORBX_E_INVALID; }
    for (int i = 0; i < cur->n; i++) match_cur[i] = -1;
    *nmatches = 0;
    for (int i = 0; i < pts->n; i++) { if (pt_choice) pt_choice[i] = -1; if (pt_dist) pt_dist[i] = 256; }
    if (cur->n == 0 || pts->n == 0) return ORBX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 16) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    ProjCtx *c = &g_proj[device];
    if (!c->stream) ORBX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t nc = (size_t)cur->n, np = (size_t)pts->n;
    // blob: frame arrays then point arrays
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += pa16(bytes); return r; };
    const size_t fx = take(4 * nc), fy = take(4 * nc), fo = take(4 * nc), fa = take(4 * nc), fu = take(4 * nc), fd = take(32 * nc), fq = take(nc);
    const size_t pu = take(4 * np), pv = take(4 * np), pa = take(4 * np), pl = take(4 * np), pg = take(4 * np), pc = take(4 * np),
                 pd = take(32 * np), pval = take(np), pobs = take(np);
    const size_t blob = o;
    if (blob > c->cap) {
        if (c->h_blob) ORBX_HIP(hipHostFree(c->h_blob));
        if (c->d_blob) ORBX_HIP(hipFree(c->d_blob));
        c->h_blob = nullptr; c->d_blob = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&c->h_blob, blob * 2, hipHostMallocDefault));
        ORBX_HIP(hipMalloc((void **)&c->d_blob, blob * 2));
        c->cap = blob * 2;
    }
    uint8_t *h = c->h_blob;
    memcpy(h + fx, cur->x, 4 * nc); memcpy(h + fy, cur->y, 4 * nc); memcpy(h + fo, cur->octave, 4 * nc);
    if (cur->angle) memcpy(h + fa, cur->angle, 4 * nc); else memset(h + fa, 0, 4 * nc);
    memcpy(h + fu, cur->u_right, 4 * nc); memcpy(h + fd, cur->desc, 32 * nc);
    if (cur->occupied) memcpy(h + fq, cur->occupied, nc); else memset(h + fq, 0, nc);
    memcpy(h + pu, pts->u, 4 * np); memcpy(h + pv, pts->v, 4 * np); memcpy(h + pl, pts->level, 4 * np);
    if (pts->aux) memcpy(h + pa, pts->aux, 4 * np); else memset(h + pa, 0, 4 * np);
    if (pts->angle) memcpy(h + pg, pts->angle, 4 * np); else memset(h + pg, 0, 4 * np);
    if (pts->view_cos) memcpy(h + pc, pts->view_cos, 4 * np); else memset(h + pc, 0, 4 * np);
    memcpy(h + pd, pts->desc, 32 * np); memcpy(h + pval, pts->valid, np);
    if (pts->has_obs) memcpy(h + pobs, pts->has_obs, np); else memset(h + pobs, 1, np);
    ORBX_HIP(hipMemcpyAsync(c->d_blob, h, blob, hipMemcpyHostToDevice, c->stream));
    const uint8_t *d = c->d_blob;
    DevFrame F;
    F.n = cur->n; F.x = (const float *)(d + fx); F.y = (const float *)(d + fy); F.octave = (const int32_t *)(d + fo);
    F.angle = (const float *)(d + fa); F.u_right = (const float *)(d + fu); F.desc = (const uint32_t *)(d + fd); F.occupied = d + fq;
    F.min_x = cur->min_x; F.min_y = cur->min_y; F.max_x = cur->max_x; F.max_y = cur->max_y;
    F.inv_w = (float)PG_COLS / (cur->max_x - cur->min_x);  // src/Frame.cc:164-165
    F.inv_h = (float)PG_ROWS / (cur->max_y - cur->min_y);
    DevPoints P;
    P.n = pts->n; P.u = (const float *)(d + pu); P.v = (const float *)(d + pv); P.aux = (const float *)(d + pa);
    P.level = (const int32_t *)(d + pl); P.angle = (const float *)(d + pg); P.view_cos = (const float *)(d + pc);
    P.desc = (const uint32_t *)(d + pd); P.valid = d + pval; P.has_obs = d + pobs;
    ProjParams pp = pp_in;
    for (int i = 0; i < ORBX_MAX_LEVELS; i++) {
        pp.sf[i] = i < nlevels ? sf[i] : 0.f;
        pp.inv_sigma2[i] = (inv_sigma2 && i < nlevels) ? inv_sigma2[i] : 0.f;
    }
    // work: cell_off[3073] | cell_idx[nc] | beg[np] | cnt[np] | pool_used | choice_a[np] | choice_b[np] | match[nc] | out_n |
    //       pt_choice[np] | pt_dist[np]
    size_t w = 0;
    auto wtake = [&](size_t bytes) { const size_t r = w; w += pa16(bytes); return r; };
    const size_t w_coff = wtake(4 * (PG_CELLS + 1)), w_cidx = wtake(4 * nc), w_beg = wtake(4 * np), w_cnt = wtake(4 * np), w_used = wtake(16),
                 w_ca = wtake(4 * np), w_cb = wtake(4 * np), w_match = wtake(4 * nc), w_n = wtake(16), w_pc = wtake(4 * np),
                 w_pd = wtake(4 * np);
    if (w > c->work_cap) {
        if (c->d_work) ORBX_HIP(hipFree(c->d_work));
        c->d_work = nullptr;
        ORBX_HIP(hipMalloc((void **)&c->d_work, w * 2));
        c->work_cap = w * 2;
    }
    if ((nc + 2 * np + 8) > c->out_cap) {
        if (c->h_out) ORBX_HIP(hipHostFree(c->h_out));
        c->h_out = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&c->h_out, sizeof(int32_t) * (nc + 2 * np + 8) * 2, hipHostMallocDefault));
        c->out_cap = (nc + 2 * np + 8) * 2;
    }
    const size_t resolve_lds = pp.init_search ? (2 * sizeof(uint16_t) + sizeof(float)) * ((nc + 7) & ~(size_t)7) + 16 : sizeof(int) * (nc + 4);   // init: md, m21, the frame's angles
    if (resolve_lds > 150 * 1024 || (pp.init_search && np >= 65535)) { orbx_set_error("too many features for one search"); return ORBX_E_INVALID; }
    if (pp.init_search)
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_resolve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)resolve_lds));
    else
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)resolve_lds));
    uint8_t *wk = c->d_work;
    int *d_coff = (int *)(wk + w_coff), *d_cidx = (int *)(wk + w_cidx), *d_beg = (int *)(wk + w_beg), *d_cnt = (int *)(wk + w_cnt),
        *d_used = (int *)(wk + w_used);
    if (!c->d_entries) { // entry pool: grown on demand, the whole call is simply repeated after an overflow
        const size_t init = 1u << 20;
        ORBX_HIP(hipMalloc((void **)&c->d_entries, sizeof(uint32_t) * init));
        c->ent_cap = init;
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        ORBX_HIP(hipMemsetAsync(d_used, 0, 16, c->stream));
        hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(256), 0, c->stream, F, d_coff, d_cidx);
        hipLaunchKernelGGL(k_proj_lists, dim3((pts->n + 3) / 4), dim3(256), 0, c->stream, F, P, pp, d_coff, d_cidx, d_beg, d_cnt,
                           c->d_entries, (int)c->ent_cap, d_used);
        if (pp.init_search)
            hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(64), resolve_lds, c->stream, F, P, pp, d_beg, d_cnt, c->d_entries,
                               (int *)(wk + w_ca), (int32_t *)(wk + w_pc), (int32_t *)(wk + w_pd), (int *)(wk + w_n));
        else
            hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(1024), resolve_lds, c->stream, F, P, pp, d_beg, d_cnt, c->d_entries,
                               (int *)(wk + w_ca), (int *)(wk + w_cb), (int32_t *)(wk + w_match), (int *)(wk + w_n),
                               (int32_t *)(wk + w_pc), (int32_t *)(wk + w_pd));
        ORBX_HIP(hipGetLastError());
        ORBX_HIP(hipMemcpyAsync(c->h_out, wk + w_match, 4 * nc, hipMemcpyDeviceToHost, c->stream));
        ORBX_HIP(hipMemcpyAsync(c->h_out + nc, wk + w_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        ORBX_HIP(hipMemcpyAsync(c->h_out + nc + 1, d_used, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (pt_choice) ORBX_HIP(hipMemcpyAsync(c->h_out + nc + 8, wk + w_pc, 4 * np, hipMemcpyDeviceToHost, c->stream));
        if (pt_dist) ORBX_HIP(hipMemcpyAsync(c->h_out + nc + 8 + np, wk + w_pd, 4 * np, hipMemcpyDeviceToHost, c->stream));
        ORBX_HIP(hipStreamSynchronize(c->stream));
        const size_t used = (size_t)(unsigned)c->h_out[nc + 1];
        if (used <= c->ent_cap) break;
        if (attempt == 1) { orbx_set_error("candidate pool overflow"); return ORBX_E_CAPACITY; }
        ORBX_HIP(hipFree(c->d_entries));
        c->d_entries = nullptr;
        ORBX_HIP(hipMalloc((void **)&c->d_entries, sizeof(uint32_t) * used * 2));
        c->ent_cap = used * 2;
    }
    memcpy(match_cur, c->h_out, 4 * nc);
    *nmatches = c->h_out[nc];
    if (pt_choice) memcpy(pt_choice, c->h_out + nc + 8, 4 * np);
    if (pt_dist) memcpy(pt_dist, c->h_out + nc + 8 + np, 4 * np);
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection_last_frame(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                                    const float *scale_factors, int nlevels, float th, int direction, float mbf,
                                                    int check_orientation, int32_t *match_cur, int *nmatches)
{
    if (direction < 0 || direction > 2) { orbx_set_error("direction must be 0 (none), 1 (forward) or 2 (backward)"); return ORBX_E_INVALID; }
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.radius_mode = 0; pp.bounds = 1; pp.need_pos_aux = 1; pp.lo_off = -1; pp.hi_off = 1; pp.direction = direction;
    pp.ur_mode = 1; pp.max_dist = 100; pp.check_ori = check_orientation & 1; pp.mark_cleared = (check_orientation >> 1) & 1; pp.claims = 1;
    pp.th = th; pp.mbf = mbf;
    return proj_run(device, cur, pts, scale_factors, nlevels, pp, match_cur, nmatches);
}

extern "C" int orbx_search_by_projection_map_points(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                                    const float *scale_factors, int nlevels, float th, float nnratio,
                                                    int32_t *match_cur, int *nmatches)
{
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.radius_mode = 1; pp.lo_off = -1; pp.hi_off = 0; pp.ur_mode = 2; pp.max_dist = 100; pp.ratio = 1; pp.claims = 1;
    pp.th = th; pp.nnratio = nnratio;
    return proj_run(device, cur, pts, scale_factors, nlevels, pp, match_cur, nmatches);
}

extern "C" int orbx_search_by_projection_keyframe(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                                  const float *scale_factors, int nlevels, float th, int orb_dist,
                                                  int check_orientation, int32_t *match_cur, int *nmatches)
{
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.bounds = 1; pp.lo_off = -1; pp.hi_off = 1; pp.max_dist = orb_dist; pp.check_ori = check_orientation & 1; pp.mark_cleared = (check_orientation >> 1) & 1; pp.claims = 2;
    pp.th = th;
    return proj_run(device, cur, pts, scale_factors, nlevels, pp, match_cur, nmatches);
}

extern "C" int orbx_search_by_projection_sim3(int device, const orbx_frame_feats *kf, const orbx_proj_points *pts,
                                              const float *scale_factors, int nlevels, float th, int32_t *match_kf, int *nmatches)
{
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.bounds = 2; pp.lo_off = -1; pp.hi_off = 0; pp.max_dist = 50; pp.claims = 2;
    pp.th = th;
    return proj_run(device, kf, pts, scale_factors, nlevels, pp, match_kf, nmatches);
}

extern "C" int orbx_window_best(int device, const orbx_frame_feats *kf, const orbx_proj_points *pts, const float *scale_factors,
                                const float *inv_sigma2, int nlevels, float th, int chi2, int max_dist, int32_t *best_idx,
                                int32_t *best_dist, int *nfound)
{
    if (!best_idx || max_dist < 0 || max_dist > 256) { orbx_set_error("orbx_window_best: invalid argument"); return ORBX_E_INVALID; }
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.bounds = 2; pp.lo_off = -1; pp.hi_off = 0; pp.max_dist = max_dist; pp.chi2 = chi2 ? 1 : 0; pp.claims = 0;
    pp.th = th;
    return proj_run(device, kf, pts, scale_factors, nlevels, pp, nullptr, nfound, inv_sigma2, best_idx, best_dist);
}

extern "C" int orbx_search_for_initialization(int device, const orbx_frame_feats *f1, const orbx_frame_feats *f2,
                                              const float *prev_matched_xy, int window_size, float nnratio, int check_orientation,
                                              int32_t *matches12, int *nmatches)
{
    if (!f1 || !f2 || !prev_matched_xy || !matches12 || !nmatches || f1->n < 0 || window_size < 0) {
        orbx_set_error("orbx_search_for_initialization: invalid argument");
        return ORBX_E_INVALID;
    }
    if (f1->n && (!f1->octave || !f1->desc || (check_orientation && !f1->angle))) { orbx_set_error("frame arrays missing"); return ORBX_E_INVALID; }
    const size_t n1 = (size_t)f1->n;
    std::vector<float> u(n1 + 1), v(n1 + 1);
    std::vector<int32_t> lvl(n1 + 1, 0);
    std::vector<uint8_t> valid(n1 + 1);
    for (size_t i = 0; i < n1; i++) {
        u[i] = prev_matched_xy[2 * i]; v[i] = prev_matched_xy[2 * i + 1];
        valid[i] = f1->octave[i] > 0 ? 0 : 1;                          // :451-453
    }
    orbx_proj_points pts;
    memset(&pts, 0, sizeof pts);
    pts.n = f1->n; pts.u = u.data(); pts.v = v.data(); pts.level = lvl.data(); pts.angle = f1->angle; pts.desc = f1->desc;
    pts.valid = valid.data();
    ProjParams pp;
    memset(&pp, 0, sizeof pp);
    pp.radius_mode = 2; pp.lo_off = 0; pp.hi_off = 0; pp.max_dist = 50; pp.check_ori = check_orientation; pp.init_search = 1;
    pp.th = (float)window_size; pp.nnratio = nnratio;
    const float sf1[1] = { 1.0f };
    return proj_run(device, f2, &pts, sf1, 1, pp, nullptr, nmatches, nullptr, matches12, nullptr);
}

extern "C" int orbx_search_by_sim3(int device, const orbx_frame_feats *kf1, const orbx_frame_feats *kf2,
                                   const orbx_proj_points *pts12, const orbx_proj_points *pts21, const float *scale_factors1,
                                   const float *scale_factors2, int nlevels, float th, int32_t *match12, int *nfound)
{
    if (!kf1 || !kf2 || !pts12 || !pts21 || !match12 || !nfound || pts12->n != kf1->n || pts21->n != kf2->n) {
        orbx_set_error("orbx_search_by_sim3: invalid argument (one projected point per keypoint on each side)");
        return ORBX_E_INVALID;
    }
    std::vector<int32_t> m1((size_t)pts12->n + 1), m2((size_t)pts21->n + 1);
    int n1 = 0, n2 = 0;
    int rc = orbx_window_best(device, kf2, pts12, scale_factors2, nullptr, nlevels, th, 0, 100, m1.data(), nullptr, &n1); // :1218-1292
    if (rc) return rc;
    rc = orbx_window_best(device, kf1, pts21, scale_factors1, nullptr, nlevels, th, 0, 100, m2.data(), nullptr, &n2);     // :1295-1372
    if (rc) return rc;
    int found = 0;
    for (int i1 = 0; i1 < pts12->n; i1++) { // the agreement check, :1375-1391
        const int idx2 = m1[i1];
        match12[i1] = -1;
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; found++; }
    }
    *nfound = found;
    return ORBX_OK;
}
