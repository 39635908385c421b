// orbx_vocab.hip — DBoW2 vocabulary tree on device and Frame::ComputeBoW
// (reference src/Frame.cc:459-466 -> TemplatedVocabulary::transform(features, BowVector, FeatureVector, 4),
//  Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194 and :1218-1259; BowVector.cpp:33-77).
// the step between the extractor and the SearchByBoW kernels.
//
// k_vocab_descend: one thread per feature walks the tree (k Hamming distances per level, first
// minimum wins, strict <), yielding word id, word weight and the node id at level L - levelsup.
// k_bow_build: one workgroup sorts (node, feature) and (word, feature) keys with an LDS bitonic
// network and emits the FeatureVector as CSR and the BowVector; weights of one word are added in
// feature order and the L1 norm is accumulated in ascending word order, i.e. in the order
// std::map / addWeight / normalize produce (doubles, so the order is part of the result).
#include "orbx_device.h"
#include <stdio.h>
#include <string.h>
#include <vector>

struct orbx_vocab {
    int device, k, L, nnodes, nwords;
    int *d_child_off, *d_child_ids, *d_word_id;      // breadth-first node order on the device; d_child_ids holds orig_id[] (device node -> DBoW2 node id)
    uint32_t *d_desc;
    double *d_weight;
    hipStream_t stream;
    // per-call scratch (grown on demand)
    uint8_t *d_in; size_t in_cap;
    uint8_t *d_out; size_t out_cap;
    uint8_t *h_out; size_t h_out_cap;
};

// The vocabulary lives on the device in BREADTH-FIRST order (vocab_build): the children of a node are consecutive nodes, child number c of the
// concatenated child lists is node c + 1, and child_off[] (first child list position of a node) is monotonic.  One memory round trip per level then
// carries everything the next step needs: the ten children's descriptors AND their own child ranges (eleven consecutive child_off entries) --
// the walk through separate child-id and offset arrays was three dependent round trips per level, the child-by-child loop before it sixty.
// orig_id[] maps back to DBoW2's node ids (the FeatureVector's node ids, TemplatedVocabulary.h:1228).
__global__ __launch_bounds__(256) void k_vocab_descend(const int *__restrict__ child_off, const int *__restrict__ orig_id,
                                                       const int *__restrict__ word_of, const uint32_t *__restrict__ ndesc,
                                                       const double *__restrict__ nweight, const uint32_t *__restrict__ feat,
                                                       int n, int nid_level, uint32_t *__restrict__ word_id,
                                                       double *__restrict__ word_w, uint32_t *__restrict__ node_id,
                                                       const int *__restrict__ n_dev, int stride, int nnodes)
{
    // batched form: grid.y = frame, n = n_dev[frame], arrays strided by `stride` features per frame
    if (n_dev) {
        const long long o = (long long)blockIdx.y * stride;
        n = min(n_dev[blockIdx.y], stride);
        feat += o * 8; word_id += o; word_w += o; node_id += o;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;    // (64-thread workgroups: a frame's 1000 features on 16 CUs, not 4 -- the descent is bound by
                                                             // the cache-line rate of a CU's vector memory path: 64 scattered lines per load instruction)
    if (i >= n) return;
    uint32_t f[8];
    {
        const uint4 *s = reinterpret_cast<const uint4 *>(feat + (long long)i * 8);
        const uint4 v0 = s[0], v1 = s[1];
        f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
    }
    int nid = 0, node = 0, level = 0; // nid stays 0 (root) when nid_level <= 0 (:1228)
    int c0 = child_off[0], c1 = child_off[1];
    for (;;) {
        ++level;
        // children ten at a time (DBoW2's k; ORBvoc has k = 10)
        int best = -1, best_d = 0x7FFFFFFF, b0 = 0, b1 = 0;
        for (int cb = c0; cb < c1; cb += 10) {
            int co[11];
            uint4 v0[10], v1[10];
#pragma unroll
            for (int j = 0; j < 11; j++) co[j] = child_off[min(cb + 1 + j, nnodes)];
#pragma unroll
            for (int j = 0; j < 10; j++) {
                const uint4 *s = reinterpret_cast<const uint4 *>(ndesc + (long long)(cb + j < c1 ? cb + 1 + j : 0) * 8);
                v0[j] = s[0]; v1[j] = s[1];
            }
#pragma unroll
            for (int j = 0; j < 10; j++) {
                const int d = __popc(f[0] ^ v0[j].x) + __popc(f[1] ^ v0[j].y) + __popc(f[2] ^ v0[j].z) + __popc(f[3] ^ v0[j].w) +
                              __popc(f[4] ^ v1[j].x) + __popc(f[5] ^ v1[j].y) + __popc(f[6] ^ v1[j].z) + __popc(f[7] ^ v1[j].w);
                if (cb + j < c1 && d < best_d) { best_d = d; best = cb + 1 + j; b0 = co[j]; b1 = co[j + 1]; } // first minimum wins (:1238-1249)
            }
        }
        node = best;
        if (level == nid_level) nid = node;
        if (b1 <= b0) break; // isLeaf(): no children
        c0 = b0; c1 = b1;
    }
    word_id[i] = (uint32_t)word_of[node];
    word_w[i] = nweight[node];
    node_id[i] = (uint32_t)orig_id[nid];
}

extern __shared__ __align__(16) unsigned char vocab_smem[];
#define BOW_DUAL_MAX_NPAD 4096

// in-LDS bitonic sort of npad (power of two) u64 keys by 256 threads, ascending.
// Pair t of a stage is (lo, lo + stride) with lo = 2 t - (t & (stride - 1)); a wave takes the pairs t = 64 w + lane (+ 256 k): for strides
// up to 64 those are exactly the elements [128 (w + 4 k), 128 (w + 4 k) + 128) -- the same block in every such stage, touched by no other wave.
// Only the stages with a stride above 64 (6 of the 55 for 1024 keys) need the workgroup barrier; the others are ordered by the wave's own
// in-order LDS accesses (the per-stage barrier was most of the kernel: 84 us per frame for two sorts of 1024 keys).
__device__ void bitonic_sort_u64(unsigned long long *keys, int npad)
{
    bool need_barrier = true;                    // (the caller's stores into keys[] came from other waves)
    for (int size = 2; size <= npad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (need_barrier || stride > 64) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            need_barrier = stride > 64;          // the next stage reads what other waves wrote in this one
            for (int t = threadIdx.x; t < (npad >> 1); t += 512) {       // two pairs per trip: their four LDS reads travel together
                const int t2 = t + 256;
                const bool two = t2 < (npad >> 1);
                const int lo = 2 * t - (t & (stride - 1)); // index with bit `stride` clear
                const int hi = lo + stride;
                const int lo2 = two ? 2 * t2 - (t2 & (stride - 1)) : lo, hi2 = two ? lo2 + stride : hi;
                const bool asc = (lo & size) == 0, asc2 = (lo2 & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi], a2 = keys[lo2], b2 = keys[hi2];
                if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
                if (two && (a2 > b2) == asc2) { keys[lo2] = b2; keys[hi2] = a2; }
            }
        }
    __syncthreads();
}

// The same network with the keys in REGISTERS: thread (wave w, lane l) holds the KPL consecutive elements [KPL (64 w + l), + KPL), npad = 256 KPL.
// Strides below KPL exchange registers of one lane, strides below 64 KPL exchange lanes of one wave (two ds_bpermute per key: no memory, no
// barrier), and only the strides of 64 KPL and more -- 3 of the 55 stages for 1024 keys -- go through LDS.  15.8 -> ~5 us per sort of 1024 keys.
// NS independent key sets (k_bow_build sorts two: (node, feature) and (word, feature)) go through the network TOGETHER: with one wave per SIMD a
// stage is a chain of dependent latencies, and the second set's instructions fill the first set's waits.
template <int KPL, int NS>
__device__ void bitonic_sort_u64_regs(unsigned long long *const (&keys)[NS])
{
    constexpr int NP = 256 * KPL;
    const int lane = threadIdx.x & 63, base = (int)threadIdx.x * KPL;
    unsigned long long k[NS][KPL];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
        for (int r = 0; r < KPL; r++) k[q][r] = keys[q][base + r];
    // (a single wave per SIMD issues a dependent instruction every 5-8 cycles, so a stage has to be FEW instructions: the direction of a
    // stage is formed once per lane -- every element of a lane has the same `size` bit once size > its KPL elements --, the partner's key
    // comes by two raw ds_bpermute, and keeping the smaller / larger key is one 64-bit compare and two selects)
    for (int size = 2; size <= NP; size <<= 1) {
        const bool asc_lane = (base & size) == 0;
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < KPL) {                  // partner = another register of this lane
#pragma unroll
                for (int s_ = 1; s_ < KPL; s_ <<= 1) {
                    if (stride != s_) continue;
#pragma unroll
                    for (int r = 0; r < KPL; r++) {
                        if (r & s_) continue;
                        const bool asc = size < KPL ? (r & size) == 0 : asc_lane;      // (r & size: scalar)
#pragma unroll
                        for (int q = 0; q < NS; q++) {
                            const unsigned long long a_ = k[q][r], b_ = k[q][r | s_];
                            const bool sw = (a_ > b_) == asc;
                            k[q][r] = sw ? b_ : a_; k[q][r | s_] = sw ? a_ : b_;
                        }
                    }
                }
            } else if (stride < 64 * KPL) {      // partner = the same register of lane ^ (stride / KPL)
                const int d = stride / KPL;      // (KPL is a power of two: a shift)
                const int paddr = (lane ^ d) << 2;
                const bool keep_min = ((lane & d) == 0) == asc_lane;
                unsigned long long o[NS][KPL];
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int r = 0; r < KPL; r++) {
                        const unsigned lo_ = (unsigned)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)k[q][r]);
                        const unsigned hi_ = (unsigned)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)(k[q][r] >> 32));
                        o[q][r] = ((unsigned long long)hi_ << 32) | lo_;
                    }
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int r = 0; r < KPL; r++) k[q][r] = ((o[q][r] < k[q][r]) == keep_min) ? o[q][r] : k[q][r];
            } else {                             // partner in another wave: through LDS
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int r = 0; r < KPL; r++) keys[q][base + r] = k[q][r];
                __syncthreads();
                const bool keep_min = ((base & stride) == 0) == asc_lane;
#pragma unroll
                for (int q = 0; q < NS; q++)
#pragma unroll
                    for (int r = 0; r < KPL; r++) {
                        const unsigned long long o = keys[q][(base + r) ^ stride];
                        k[q][r] = ((o < k[q][r]) == keep_min) ? o : k[q][r];
                    }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
        for (int r = 0; r < KPL; r++) keys[q][base + r] = k[q][r];
    __syncthreads();
}

// both key sets of k_bow_build
__device__ __forceinline__ void bow_sort2(unsigned long long *ka, unsigned long long *kb, int npad)
{
    unsigned long long *const both[2] = { ka, kb };
    if (npad == 1024) bitonic_sort_u64_regs<4, 2>(both);         // ORB-SLAM2's 1000 features per frame
    else if (npad == 2048) bitonic_sort_u64_regs<8, 2>(both);    // 2000 (KITTI settings)
    else { bitonic_sort_u64(ka, npad); bitonic_sort_u64(kb, npad); }
}

__global__ __launch_bounds__(256) void k_bow_build(int n, int npad, const uint32_t *__restrict__ word_id,
                                                   const double *__restrict__ word_w, const uint32_t *__restrict__ node_id,
                                                   uint32_t *__restrict__ bow_id, double *__restrict__ bow_val, int *__restrict__ counts,
                                                   uint32_t *__restrict__ fv_node_id, int32_t *__restrict__ fv_node_off,
                                                   uint32_t *__restrict__ fv_feat, const int *__restrict__ n_dev, int stride)
{
    if (n_dev) { // batched form: one workgroup per frame (blockIdx.x)
        const long long o = (long long)blockIdx.x * stride;
        n = min(n_dev[blockIdx.x], stride);
        word_id += o; word_w += o; node_id += o; bow_id += o; bow_val += o; counts += 2 * blockIdx.x;
        fv_node_id += o; fv_node_off += (long long)blockIdx.x * (stride + 4); fv_feat += (long long)blockIdx.x * npad;
    }
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(vocab_smem);     // (node id, feature) keys, then reused for nothing else
    const bool dual = npad <= BOW_DUAL_MAX_NPAD;                                         // both key sets in LDS at once (above: one after the other, 12 bytes per key)
    unsigned long long *keys_w = dual ? keys + npad : keys;                              // (word id, feature) keys
    int *flags = reinterpret_cast<int *>(keys_w + npad);
    __shared__ int s_w[4];
    __shared__ double s_norm;
    const int tid = threadIdx.x;
    // ---- FeatureVector: (node id, feature) ascending; stopped words (weight 0) are in no node (:1157)
    for (int i = tid; i < npad; i += 256) {
        const bool on = i < n && word_w[i] > 0;
        keys[i] = on ? (((unsigned long long)node_id[i] << 32) | (unsigned)i) : ~0ull;
        if (dual) keys_w[i] = on ? (((unsigned long long)word_id[i] << 32) | (unsigned)i) : ~0ull;
    }
    if (dual) bow_sort2(keys, keys_w, npad); else bitonic_sort_u64(keys, npad);
    for (int i = tid; i < npad; i += 256) {
        const bool valid = keys[i] != ~0ull;
        flags[i] = valid && (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32));
        if (valid) fv_feat[i] = (uint32_t)(keys[i] & 0xFFFFFFFFu);
    }
    __syncthreads();
    int m_local = 0;
    for (int i = tid; i < npad; i += 256) m_local += keys[i] != ~0ull;
    const int nn = lds_excl_scan(flags, npad, s_w);
    for (int i = tid; i < npad; i += 256) {
        const bool head = keys[i] != ~0ull && (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32));
        if (head) { fv_node_id[flags[i]] = (uint32_t)(keys[i] >> 32); fv_node_off[flags[i]] = i; }
    }
    int m;
    block_excl_scan256(m_local, &m, s_w);
    if (tid == 0) { fv_node_off[nn] = m; counts[1] = nn; }
    __syncthreads();
    // ---- BowVector: (word id, feature) ascending; per word the weights are added in feature order
    if (dual) keys = keys_w;     // (sorted above, together with the node keys)
    else {
        for (int i = tid; i < npad; i += 256)
            keys[i] = (i < n && word_w[i] > 0) ? (((unsigned long long)word_id[i] << 32) | (unsigned)i) : ~0ull;
        bitonic_sort_u64(keys, npad);
    }
    for (int i = tid; i < npad; i += 256)
        flags[i] = keys[i] != ~0ull && (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32));
    __syncthreads();
    const int nb = lds_excl_scan(flags, npad, s_w);
    for (int i = tid; i < npad; i += 256) {
        const unsigned long long key = keys[i];
        const bool head = key != ~0ull && (i == 0 || (key >> 32) != (keys[i - 1] >> 32));
        if (head) {
            double acc = word_w[key & 0xFFFFFFFFu]; // addWeight: first insert, then += in feature order (BowVector.cpp:33-45)
            for (int j = i + 1; j < npad && (keys[j] >> 32) == (key >> 32) && keys[j] != ~0ull; j++) acc += word_w[keys[j] & 0xFFFFFFFFu];
            bow_id[flags[i]] = (uint32_t)(key >> 32);
            bow_val[flags[i]] = acc;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (tid < 64) { // normalize(L1): ascending word order, strictly sequential doubles (BowVector.cpp:58-77).
        // Wave 0 loads 64 values at a time and adds them in lane order from registers (broadcast reads), so the
        // dependent chain is 64 register adds per memory round trip instead of one.
        // (the broadcast is v_readlane with a constant lane, 64 unrolled steps per block: through __shfl -- an LDS permute per step --
        // this loop was 50 of the kernel's 89 us for a 1000-word vector; lanes past the end hold +0.0, which changes no sum)
        double norm = 0.0;
        for (int base = 0; base < nb; base += 64) {
            const double v = base + tid < nb ? fabs(bow_val[base + tid]) : 0.0;
            const int v_lo = __double2loint(v), v_hi = __double2hiint(v);
#pragma unroll
            for (int k = 0; k < 64; k++) norm += __hiloint2double(__builtin_amdgcn_readlane(v_hi, k), __builtin_amdgcn_readlane(v_lo, k));
        }
        if (tid == 0) { s_norm = norm; counts[0] = nb; }
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int i = tid; i < nb; i += 256) bow_val[i] = bow_val[i] / norm;
}

// ---------------------------------------------------------------- host side

static int vocab_build(int device, int k, int L, int nm1, const int32_t *parent, const uint8_t *is_leaf,
                       const uint8_t *desc, const double *weight, orbx_vocab **out)
{
    if (!out || !parent || !is_leaf || !desc || !weight || nm1 < 1 || k < 0 || k > 20 || L < 1 || L > 10) { // loader limits :1379
        orbx_set_error("orbx_vocab_create: invalid argument");
        return ORBX_E_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    const int n = nm1 + 1;
    std::vector<int> off(n + 1, 0), ids(n > 1 ? n - 1 : 1), word(n, -1);
    for (int i = 1; i < n; i++) {
        const int p = parent[i - 1];
        if (p < 0 || p >= i) { orbx_set_error("vocabulary node %d has parent %d (parents must precede children)", i, p); return ORBX_E_INVALID; }
        off[p + 1]++;
    }
    for (int i = 0; i < n; i++) off[i + 1] += off[i];
    {
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (int i = 1; i < n; i++) ids[cur[parent[i - 1]]++] = i; // children in id order (:1409)
    }
    if (off[1] == 0) { orbx_set_error("vocabulary root has no children"); return ORBX_E_INVALID; }
    int nwords = 0;
    for (int i = 1; i < n; i++) if (is_leaf[i - 1]) word[i] = nwords++; // :1425-1432
    std::vector<double> w(n, 0.0);
    std::vector<uint8_t> d((size_t)n * 32, 0);
    for (int i = 1; i < n; i++) w[i] = weight[i - 1];
    memcpy(d.data() + 32, desc, (size_t)nm1 * 32);
    {   // breadth-first renumbering for the device (see k_vocab_descend): order[p] = DBoW2 id of device node p; a node's children, in id order, are
        // consecutive device nodes, and child number c of the concatenated lists is device node c + 1
        std::vector<int> order; order.reserve(n); order.push_back(0);
        for (size_t q = 0; q < order.size(); q++) { const int x = order[q]; for (int c = off[x]; c < off[x + 1]; c++) order.push_back(ids[c]); }
        if ((int)order.size() != n) { orbx_set_error("vocabulary: %d of %d nodes reachable from the root", (int)order.size(), n); return ORBX_E_INVALID; }
        std::vector<int> noff(n + 1, 0), nword(n);
        std::vector<double> nw(n);
        std::vector<uint8_t> nd((size_t)n * 32);
        for (int p = 0; p < n; p++) {
            const int x = order[p];
            noff[p + 1] = noff[p] + (off[x + 1] - off[x]);
            nword[p] = word[x]; nw[p] = w[x];
            memcpy(nd.data() + 32 * (size_t)p, d.data() + 32 * (size_t)x, 32);
        }
        off.swap(noff); word.swap(nword); w.swap(nw); d.swap(nd);
        ids.assign(order.begin(), order.end());      // uploaded as orig_id[]: device node -> DBoW2 node id
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_vocab *v = new orbx_vocab();
    memset(v, 0, sizeof *v);
    v->device = device; v->k = k; v->L = L; v->nnodes = n; v->nwords = nwords;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_child_off, sizeof(int) * (n + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_child_ids, sizeof(int) * ids.size());
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_word_id, sizeof(int) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_desc, (size_t)n * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_weight, sizeof(double) * n);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_off, off.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_word_id, word.data(), sizeof(int) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, d.data(), (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight, w.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) { orbx_set_error("vocabulary upload failed: %s", hipGetErrorString(e)); orbx_vocab_destroy(v); return ORBX_E_HIP; }
    *out = v;
    return ORBX_OK;
}

extern "C" int orbx_vocab_create(int device, int k, int L, int nnodes_minus_root, const int32_t *parent, const uint8_t *is_leaf,
                                 const uint8_t *desc, const double *weight, orbx_vocab **out)
{
    return vocab_build(device, k, L, nnodes_minus_root, parent, is_leaf, desc, weight, out);
}

// the ORBvoc.txt format of TemplatedVocabulary::loadFromTextFile (:1358-1445)
extern "C" int orbx_vocab_load_text(int device, const char *path, orbx_vocab **out)
{
    if (!path || !out) { orbx_set_error("orbx_vocab_load_text: null argument"); return ORBX_E_INVALID; }
    FILE *f = fopen(path, "r");
    if (!f) { orbx_set_error("cannot open vocabulary file %s", path); return ORBX_E_INVALID; }
    int k, L, n1, n2;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        fclose(f);
        orbx_set_error("Vocabulary loading failure: This is not a correct text file!"); // message of the reference loader (:1381)
        return ORBX_E_INVALID;
    }
    if (n1 != 0 || n2 != 0) { // ORB-SLAM2's vocabulary is L1_NORM (0) + TF_IDF (0); other scorings are not implemented
        fclose(f);
        orbx_set_error("only scoring L1_NORM and weighting TF_IDF are supported (file has %d %d)", n1, n2);
        return ORBX_E_INVALID;
    }
    std::vector<int32_t> parent; std::vector<uint8_t> leaf, desc; std::vector<double> w;
    for (;;) {
        int pid, isl;
        if (fscanf(f, "%d %d", &pid, &isl) != 2) break;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32 && ok; i++) { int b; if (fscanf(f, "%d", &b) != 1) ok = false; else d[i] = (uint8_t)b; }
        double wt;
        if (!ok || fscanf(f, "%lf", &wt) != 1) break;
        parent.push_back(pid); leaf.push_back(isl > 0); desc.insert(desc.end(), d, d + 32); w.push_back(wt);
    }
    fclose(f);
    if (parent.empty()) { orbx_set_error("vocabulary file %s holds no nodes", path); return ORBX_E_INVALID; }
    return vocab_build(device, k, L, (int)parent.size(), parent.data(), leaf.data(), desc.data(), w.data(), out);
}

extern "C" void orbx_vocab_destroy(orbx_vocab *v)
{
    if (!v) return;
    hipSetDevice(v->device);
    if (v->stream) { hipStreamSynchronize(v->stream); hipStreamDestroy(v->stream); }
    void *ptrs[] = { v->d_child_off, v->d_child_ids, v->d_word_id, v->d_desc, v->d_weight, v->d_in, v->d_out };
    for (void *p : ptrs) if (p) hipFree(p);
    if (v->h_out) hipHostFree(v->h_out);
    delete v;
}

extern "C" int orbx_vocab_info(const orbx_vocab *v, int *k, int *L, int *nnodes, int *nwords)
{
    if (!v) { orbx_set_error("null vocabulary"); return ORBX_E_INVALID; }
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return ORBX_OK;
}

static size_t a16v(size_t x) { return (x + 15) & ~(size_t)15; }

extern "C" int orbx_bow_transform(orbx_vocab *v, const uint8_t *desc, int n, int levelsup,
                                  uint32_t *word_id, double *word_weight, uint32_t *node_id,
                                  uint32_t *bow_id, double *bow_val, int *nbow,
                                  uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes)
{
    if (!v || n < 0 || (n && !desc) || !nbow || !fv_nnodes || (n && (!bow_id || !bow_val || !fv_node_id || !fv_node_off || !fv_feat))) {
        orbx_set_error("orbx_bow_transform: invalid argument");
        return ORBX_E_INVALID;
    }
    if (n > 8192) { orbx_set_error("orbx_bow_transform: at most 8192 features per call"); return ORBX_E_INVALID; }
    *nbow = 0; *fv_nnodes = 0;
    if (n == 0) { if (fv_node_off) fv_node_off[0] = 0; return ORBX_OK; }
    ORBX_HIP(hipSetDevice(v->device));
    int npad = 64;
    while (npad < n) npad <<= 1;
    // device scratch: [desc | word u32 | weight f64 | nid u32 | bow_id | bow_val | counts | fv_node_id | fv_node_off | fv_feat]
    const size_t o_word = a16v((size_t)n * 32), o_w = o_word + a16v(4 * (size_t)n), o_nid = o_w + a16v(8 * (size_t)n);
    const size_t o_bid = o_nid + a16v(4 * (size_t)n), o_bval = o_bid + a16v(4 * (size_t)n), o_cnt = o_bval + a16v(8 * (size_t)n);
    const size_t o_fid = o_cnt + 16, o_foff = o_fid + a16v(4 * (size_t)n), o_ffeat = o_foff + a16v(4 * ((size_t)n + 1));
    const size_t total = o_ffeat + a16v(4 * (size_t)npad);
    if (total > v->in_cap) {
        if (v->d_in) ORBX_HIP(hipFree(v->d_in));
        if (v->h_out) ORBX_HIP(hipHostFree(v->h_out));
        v->d_in = nullptr; v->h_out = nullptr;
        ORBX_HIP(hipMalloc((void **)&v->d_in, total * 2));
        ORBX_HIP(hipHostMalloc((void **)&v->h_out, total * 2, hipHostMallocDefault));
        v->in_cap = total * 2;
    }
    uint8_t *d = v->d_in;
    ORBX_HIP(hipMemcpyAsync(d, desc, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    hipLaunchKernelGGL(k_vocab_descend, dim3((n + 63) / 64), dim3(64), 0, v->stream, v->d_child_off, v->d_child_ids, v->d_word_id,
                       v->d_desc, v->d_weight, (const uint32_t *)d, n, v->L - levelsup, (uint32_t *)(d + o_word), (double *)(d + o_w),
                       (uint32_t *)(d + o_nid), nullptr, 0, v->nnodes);
    const size_t lds = (size_t)npad * (npad <= BOW_DUAL_MAX_NPAD ? 20 : 12) + 64;     // key sets (u64: both at once up to BOW_DUAL_MAX_NPAD keys) + flags (int)
    ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bow_build, dim3(1), dim3(256), lds, v->stream, n, npad, (const uint32_t *)(d + o_word), (const double *)(d + o_w),
                       (const uint32_t *)(d + o_nid), (uint32_t *)(d + o_bid), (double *)(d + o_bval), (int *)(d + o_cnt),
                       (uint32_t *)(d + o_fid), (int32_t *)(d + o_foff), (uint32_t *)(d + o_ffeat), nullptr, 0);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(v->h_out + o_word, d + o_word, total - o_word, hipMemcpyDeviceToHost, v->stream));
    ORBX_HIP(hipStreamSynchronize(v->stream));
    const uint8_t *h = v->h_out;
    const int *cnt = (const int *)(h + o_cnt);
    const int nb = cnt[0], nn = cnt[1];
    if (word_id) memcpy(word_id, h + o_word, 4 * (size_t)n);
    if (word_weight) memcpy(word_weight, h + o_w, 8 * (size_t)n);
    if (node_id) memcpy(node_id, h + o_nid, 4 * (size_t)n);
    memcpy(bow_id, h + o_bid, 4 * (size_t)nb);
    memcpy(bow_val, h + o_bval, 8 * (size_t)nb);
    memcpy(fv_node_id, h + o_fid, 4 * (size_t)nn);
    memcpy(fv_node_off, h + o_foff, 4 * ((size_t)nn + 1));
    memcpy(fv_feat, h + o_ffeat, 4 * (size_t)fv_node_off[nn]);
    *nbow = nb; *fv_nnodes = nn;
    return ORBX_OK;
}

// ---------------------------------------------------------------- batched, device-resident form (BASELINE config 3)
// Frame::ComputeBoW for a whole batch of extracted frames without leaving the device: descriptors and counts are the
// extractor's outputs, the FeatureVectors stay in HBM as the DevFeat records orbx_bowdb_search_batch_device consumes.

__global__ __launch_bounds__(256) void k_frames_fill(DevFeat *__restrict__ feats, const orbx_keypoint *__restrict__ kps,
                                                     const uint32_t *__restrict__ desc, const int *__restrict__ n_dev, int cap, int npad,
                                                     const int *__restrict__ counts, const uint32_t *__restrict__ fv_node_id,
                                                     const int32_t *__restrict__ fv_node_off, const uint32_t *__restrict__ fv_feat,
                                                     float *__restrict__ angle, uint32_t *__restrict__ sdesc, uint8_t *__restrict__ sflag)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_dev[b], cap);
    for (int i = tid; i < n; i += 256) angle[(long long)b * cap + i] = kps[(long long)b * cap + i].angle;
    {   // descriptors once more in FeatureVector list order (what the search kernel streams through)
        const int nn = counts[2 * b + 1];
        const int m = fv_node_off[(long long)b * (cap + 4) + nn];
        const uint32_t *list = fv_feat + (long long)b * npad;
        const uint4 *src = reinterpret_cast<const uint4 *>(desc + (long long)b * cap * 8);
        uint4 *dst = reinterpret_cast<uint4 *>(sdesc + (long long)b * npad * 8);
        for (int i = tid; i < 2 * m; i += 256) dst[i] = src[2 * (long long)list[i >> 1] + (i & 1)];
        for (int i = tid; i < m; i += 256) sflag[(long long)b * npad + i] = 1;
    }
    if (tid == 0) {
        DevFeat f;
        f.n = n; f.nnodes = counts[2 * b + 1];
        f.desc = desc + (long long)b * cap * 8;
        f.node_id = fv_node_id + (long long)b * cap;
        f.node_off = fv_node_off + (long long)b * (cap + 4);
        f.feat = fv_feat + (long long)b * npad;
        f.flag = nullptr;                       // the frame side of SearchByBoW(KF, F) has no flag test
        f.angle = angle + (long long)b * cap;
        f.sdesc = sdesc + (long long)b * npad * 8; f.sflag = sflag + (long long)b * npad;
        f.x = f.y = f.u_right = nullptr; f.octave = nullptr;
        feats[b] = f;
    }
}

extern "C" int orbx_bow_frames_create(int device, int max_batch, int cap, orbx_bow_frames **out)
{
    if (!out || max_batch < 1 || cap < 1 || cap > 8192) { orbx_set_error("orbx_bow_frames_create: invalid argument (cap <= 8192)"); return ORBX_E_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        orbx_set_error("no usable HIP device %d (liborbx has no CPU fallback)", device);
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_bow_frames *f = new orbx_bow_frames();
    memset(f, 0, sizeof *f);
    f->device = device; f->batch = max_batch; f->cap = cap;
    f->npad = 64;
    while (f->npad < cap) f->npad <<= 1;
    const size_t B = (size_t)max_batch, c = (size_t)cap;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += a16v(bytes); return r; };
    const size_t o_word = take(4 * B * c), o_w = take(8 * B * c), o_nid = take(4 * B * c), o_bid = take(4 * B * c), o_bval = take(8 * B * c),
                 o_cnt = take(8 * B), o_fid = take(4 * B * c), o_foff = take(4 * B * (c + 4)), o_ffeat = take(4 * B * f->npad),
                 o_ang = take(4 * B * c), o_feats = take(sizeof(DevFeat) * B), o_sd = take(32 * B * f->npad), o_sf = take(B * f->npad);
    if (hipMalloc((void **)&f->d_buf, o) != hipSuccess) { orbx_set_error("orbx_bow_frames_create: hipMalloc of %zu bytes failed", o); delete f; return ORBX_E_HIP; }
    uint8_t *d = f->d_buf;
    f->word_id = (uint32_t *)(d + o_word); f->word_w = (double *)(d + o_w); f->node_id = (uint32_t *)(d + o_nid);
    f->bow_id = (uint32_t *)(d + o_bid); f->bow_val = (double *)(d + o_bval); f->counts = (int *)(d + o_cnt);
    f->fv_node_id = (uint32_t *)(d + o_fid); f->fv_node_off = (int32_t *)(d + o_foff); f->fv_feat = (uint32_t *)(d + o_ffeat);
    f->angle = (float *)(d + o_ang); f->d_feats = (DevFeat *)(d + o_feats);
    f->sdesc = (uint32_t *)(d + o_sd); f->sflag = d + o_sf;
    *out = f;
    return ORBX_OK;
}

extern "C" void orbx_bow_frames_destroy(orbx_bow_frames *f)
{
    if (!f) return;
    hipSetDevice(f->device);
    if (f->d_buf) hipFree(f->d_buf);
    if (f->h_buf) hipHostFree(f->h_buf);
    delete f;
}

extern "C" int orbx_bow_transform_batch_device(orbx_vocab *v, orbx_bow_frames *f, const void *d_kps, const void *d_desc, const void *d_n,
                                               int batch, int levelsup, void *stream)
{
    if (!v || !f || !d_kps || !d_desc || !d_n || batch < 1 || batch > f->batch || v->device != f->device) {
        orbx_set_error("orbx_bow_transform_batch_device: invalid argument");
        return ORBX_E_INVALID;
    }
    ORBX_HIP(hipSetDevice(v->device));
    hipStream_t s = stream ? (hipStream_t)stream : v->stream;
    f->last_stream = s;
    const int cap = f->cap, npad = f->npad;
    hipLaunchKernelGGL(k_vocab_descend, dim3((cap + 63) / 64, batch), dim3(64), 0, s, v->d_child_off, v->d_child_ids, v->d_word_id,
                       v->d_desc, v->d_weight, (const uint32_t *)d_desc, 0, v->L - levelsup, f->word_id, f->word_w, f->node_id,
                       (const int *)d_n, cap, v->nnodes);
    const size_t lds = (size_t)npad * (npad <= BOW_DUAL_MAX_NPAD ? 20 : 12) + 64;     // key sets (u64: both at once up to BOW_DUAL_MAX_NPAD keys) + flags (int)
    ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bow_build, dim3(batch), dim3(256), lds, s, 0, npad, f->word_id, f->word_w, f->node_id, f->bow_id, f->bow_val,
                       f->counts, f->fv_node_id, f->fv_node_off, f->fv_feat, (const int *)d_n, cap);
    hipLaunchKernelGGL(k_frames_fill, dim3(batch), dim3(256), 0, s, f->d_feats, (const orbx_keypoint *)d_kps, (const uint32_t *)d_desc,
                       (const int *)d_n, cap, npad, f->counts, f->fv_node_id, f->fv_node_off, f->fv_feat, f->angle, f->sdesc, f->sflag);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// host copy of one frame's transform (tests, and the adaptor when it needs mBowVec / mFeatVec on the host);
// synchronises `stream`.  Capacities: bow_id / bow_val / fv_node_id / fv_feat hold cap entries, fv_node_off cap + 1.
extern "C" int orbx_bow_frames_read(orbx_bow_frames *f, int index, void *stream, uint32_t *bow_id, double *bow_val, int *nbow,
                                    uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes)
{
    if (!f || index < 0 || index >= f->batch || !nbow || !fv_nnodes) { orbx_set_error("orbx_bow_frames_read: invalid argument"); return ORBX_E_INVALID; }
    ORBX_HIP(hipSetDevice(f->device));
    const size_t c = (size_t)f->cap;
    const size_t o_bid = 16, o_bval = o_bid + a16v(4 * c), o_fid = o_bval + a16v(8 * c), o_foff = o_fid + a16v(4 * c),
                 o_ffeat = o_foff + a16v(4 * (c + 4)), total = o_ffeat + a16v(4 * (size_t)f->npad);
    if (total > f->h_cap) {
        if (f->h_buf) ORBX_HIP(hipHostFree(f->h_buf));
        f->h_buf = nullptr;
        ORBX_HIP(hipHostMalloc((void **)&f->h_buf, total, hipHostMallocDefault));
        f->h_cap = total;
    }
    hipStream_t s = stream ? (hipStream_t)stream : f->last_stream;
    uint8_t *h = f->h_buf;
    const size_t i = (size_t)index;
    ORBX_HIP(hipMemcpyAsync(h, f->counts + 2 * i, 8, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h + o_bid, f->bow_id + i * c, 4 * c, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h + o_bval, f->bow_val + i * c, 8 * c, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h + o_fid, f->fv_node_id + i * c, 4 * c, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h + o_foff, f->fv_node_off + i * (c + 4), 4 * (c + 1), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h + o_ffeat, f->fv_feat + i * f->npad, 4 * (size_t)f->npad, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    const int nb = ((const int *)h)[0], nn = ((const int *)h)[1];
    *nbow = nb; *fv_nnodes = nn;
    if (bow_id) memcpy(bow_id, h + o_bid, 4 * (size_t)nb);
    if (bow_val) memcpy(bow_val, h + o_bval, 8 * (size_t)nb);
    if (fv_node_id) memcpy(fv_node_id, h + o_fid, 4 * (size_t)nn);
    if (fv_node_off) memcpy(fv_node_off, h + o_foff, 4 * ((size_t)nn + 1));
    if (fv_feat && fv_node_off) memcpy(fv_feat, h + o_ffeat, 4 * (size_t)fv_node_off[nn]);
    return ORBX_OK;
}
