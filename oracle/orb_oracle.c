/*
 * orb_oracle.c — CPU restatement (plain C11) of the ORB-SLAM2 ORB front end and matchers.
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h).  PARITY UNPINNED (see orb_oracle.h).
 * Compile with -ffp-contract=off: several expressions below must round after every
 * floating-point operation exactly as written.
 *
 * Every function cites the reference lines it follows (paths relative to the reference
 * root) or the OpenCV primitive it restates.
 */
#include "orb_oracle.h"
#include "orb_pattern.inc"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { PATCH_SIZE = 31, HALF_PATCH = 15, EDGE_TH = 19, MAX_LEVELS = 32 };
enum { TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30 }; /* src/ORBmatcher.cc:37-39 */

/* ------------------------------------------------------------------ primitives */

/* cvRound: round-half-to-even under the default rounding mode. */
int oracle_cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* cv::fastAtan2 (OpenCV 3.x polynomial): degrees in [0,360), fp32, no FMA. */
float oracle_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/*
 * Deterministic sin/cos shared (as an operation sequence) with the HIP kernels.
 * The reference calls libm cosf/sinf on a float (src/ORBextractor.cc:123), whose last
 * bit is libm-dependent; the build fixes it: evaluate in fp64 with a fixed
 * Cody-Waite reduction by pi/2 and fixed minimax polynomials on [-pi/4,pi/4] (the
 * classic fdlibm kernel coefficients), explicit operation order, then round to fp32.
 */
void oracle_sincos(float angle_rad, float *s_out, float *c_out)
{
    static const double INV_PIO2 = 6.36619772367581382433e-01;
    static const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    static const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
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
    switch (q) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1733-1749: SWAR popcount over 8 words. */
int oracle_hamming(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4);
        memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}

/* ------------------------------------------------------------------ resize (E2) */

/* cv::resize INTER_LINEAR for 8UC1, OpenCV<=3.3 generic path. */
static void linear_coeffs(int ssize, int dsize, int *ofs, short *c0, short *c1)
{
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floor((double)f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        c0[d] = sat_short(oracle_cv_round_f((1.f - f) * 2048));
        c1[d] = sat_short(oracle_cv_round_f(f * 2048));
    }
}

void oracle_resize_linear(const uint8_t *src, int sw, int sh, size_t sstride,
                          uint8_t *dst, int dw, int dh, size_t dstride)
{
    /* cv::resize replaces INTER_LINEAR by INTER_AREA when both scale factors are exactly 2 (imgproc/src/imgwarp.cpp [OpenCV 3.2,
     * from memory]: `if( interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2 ) interpolation = INTER_AREA;`);
     * the 8U fast-area kernel for a 2x2 block is (a + b + c + d + 2) >> 2.  scaleFactor 1.2 never gets here; 2.0 does on even sizes. */
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int dy = 0; dy < dh; dy++) {
            const uint8_t *r0 = src + (size_t)(2 * dy) * sstride, *r1 = r0 + sstride;
            for (int dx = 0; dx < dw; dx++)
                dst[(size_t)dy * dstride + dx] = (uint8_t)((r0[2 * dx] + r0[2 * dx + 1] + r1[2 * dx] + r1[2 * dx + 1] + 2) >> 2);
        }
        return;
    }
    int *xofs = malloc(sizeof(int) * dw), *yofs = malloc(sizeof(int) * dh);
    short *a0 = malloc(2 * dw), *a1 = malloc(2 * dw), *b0 = malloc(2 * dh), *b1 = malloc(2 * dh);
    linear_coeffs(sw, dw, xofs, a0, a1);
    linear_coeffs(sh, dh, yofs, b0, b1);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
        const uint8_t *r0 = src + (size_t)sy0 * sstride, *r1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx0 = xofs[dx], sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
            int t0 = r0[sx0] * a0[dx] + r0[sx1] * a1[dx];
            int t1 = r1[sx0] * a0[dx] + r1[sx1] * a1[dx];
            int v = (((b0[dy] * (t0 >> 4)) >> 16) + ((b1[dy] * (t1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dstride + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xofs); free(yofs); free(a0); free(a1); free(b0); free(b1);
}

/* ------------------------------------------------------------------ blur (E6) */

static int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) for 8U (reference src/ORBextractor.cc:1311).  Two OpenCV
 * generations, same arithmetic (exact integer row pass; (sum+2^15)>>16 column pass), different 8-bit fixed-point taps:
 *   profile 0 (TAPS_257; OpenCV 3.2, the version the reference was tested with): float kernel -> cvRound(k*256), not renormalised:
 *             18 34 49 55 49 34 18 (sum 257);
 *   profile 1 (TAPS_256; getGaussianKernelFixedPoint_ED of later 3.4.x / 4.x releases): rounded from the outside in, the rounding error
 *             carried to the next tap, centre = 256 - the rest: 18 34 48 56 48 34 18 (sum 256).
 * Both from memory of OpenCV, and so is the release at which the second replaced the first: parity unpinned. */
void oracle_gaussian_taps(int profile, int taps[7])
{
    double scale2x = -0.5 / (2.0 * 2.0);
    if (profile == 0) {
        float cf[7];
        double sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = i - 3.0;
            double t = exp(scale2x * x * x);
            cf[i] = (float)t;
            sum += cf[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) {
            cf[i] = (float)(cf[i] * sum);
            taps[i] = cv_round_d((double)cf[i] * 256.0);
        }
    } else {
        double k[7], sum = 0, err = 0;
        int rest = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3.0; k[i] = exp(scale2x * x * x); sum += k[i]; }
        for (int i = 0; i < 3; i++) {
            double adj = k[i] / sum * 256.0 + err;
            int v = cv_round_d(adj);
            err = adj - v;
            taps[i] = taps[6 - i] = v;
            rest += 2 * v;
        }
        taps[3] = 256 - rest;
    }
}

void oracle_gaussian_blur7(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride)
{
    oracle_gaussian_blur7_profile(src, w, h, sstride, dst, dstride, 0);
}

void oracle_gaussian_blur7_profile(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride, int profile)
{
    int taps[7];
    oracle_gaussian_taps(profile, taps);
    int *rows = malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -3; k <= 3; k++)
                acc += taps[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            rows[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -3; k <= 3; k++)
                acc += taps[k + 3] * rows[(size_t)reflect101(y + k, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    free(rows);
}

/* ------------------------------------------------------------------ FAST-9/16 (E3) */

/* ring offsets (x,y) of cv::FAST TYPE_9_16 */
static const int RING_X[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int RING_Y[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

/* cv::FAST_t<16> corner test + cornerScore<16> (OpenCV fast.cpp / fast_score.cpp):
 * corner iff >=9 contiguous ring pixels all < v-t or all > v+t; score = max(t,A,B)-1.
 * Returns 0 for a non-corner (a corner's score is >= t >= 1 for the thresholds used). */
int oracle_fast_score(const uint8_t *p, int stride, int threshold)
{
    int v = p[0], d[25], is_corner = 0, cnt;
    for (int k = 0; k < 25; k++)
        d[k] = v - p[RING_Y[k & 15] * stride + RING_X[k & 15]];
    cnt = 0;
    for (int k = 0; k < 25; k++) { /* darker arc: x < v - t */
        if (d[k] > threshold) { if (++cnt > 8) { is_corner = 1; break; } } else cnt = 0;
    }
    cnt = 0;
    for (int k = 0; k < 25 && !is_corner; k++) { /* brighter arc: x > v + t */
        if (-d[k] > threshold) { if (++cnt > 8) { is_corner = 1; break; } } else cnt = 0;
    }
    if (!is_corner) return 0;
    int a0 = threshold;
    for (int k = 0; k < 16; k++) {
        int a = d[k];
        for (int j = 1; j < 9; j++) a = a < d[k + j] ? a : d[k + j];
        if (a > a0) a0 = a;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k++) {
        int b = d[k];
        for (int j = 1; j < 9; j++) b = b > d[k + j] ? b : d[k + j];
        if (b < b0) b0 = b;
    }
    return -b0 - 1;
}

typedef struct { int x, y, resp; } cand_t;
typedef struct { cand_t *v; int n, cap; } cand_vec;

static void cand_push(cand_vec *c, int x, int y, int r)
{
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 1024; c->v = realloc(c->v, sizeof(cand_t) * c->cap); }
    c->v[c->n].x = x; c->v[c->n].y = y; c->v[c->n].resp = r; c->n++;
}

/* cv::FAST(roi, kps, threshold, nonmax=true) on the cell roi [x0,x1)x[y0,y1) of a level:
 * detection on roi rows/cols [3, size-3), uchar score buffer that is 0 outside the detection
 * range, keep iff score strictly greater than all 8 neighbours; row-major output in roi coords. */
static int fast_cell(const uint8_t *img, int stride, int x0, int y0, int x1, int y1, int th,
                     uint8_t *score /* (y1-y0)*(x1-x0) scratch */, cand_vec *out, int addx, int addy)
{
    int cw = x1 - x0, ch = y1 - y0, found = 0;
    memset(score, 0, (size_t)cw * ch);
    for (int y = 3; y < ch - 3; y++)
        for (int x = 3; x < cw - 3; x++)
            score[y * cw + x] = (uint8_t)oracle_fast_score(img + (size_t)(y0 + y) * stride + x0 + x, stride, th);
    for (int y = 3; y < ch - 3; y++)
        for (int x = 3; x < cw - 3; x++) {
            int s = score[y * cw + x];
            if (!s) continue;
            const uint8_t *q = score + y * cw + x;
            if (s > q[-1] && s > q[1] && s > q[-cw - 1] && s > q[-cw] && s > q[-cw + 1] &&
                s > q[cw - 1] && s > q[cw] && s > q[cw + 1]) {
                cand_push(out, x + addx, y + addy, s);
                found++;
            }
        }
    return found;
}

/* ------------------------------------------------------------------ quadtree cull (E4) */

/* ExtractorNode (include/ORBextractor.h:45-56) kept in an index-linked list that mimics
 * std::list push_front/erase; `seq` is the creation order, which DEFINES the tie-break the
 * reference leaves to pointer addresses (src/ORBextractor.cc:832; SURVEY A.4: equal counts ->
 * the node created later is split first). */
typedef struct {
    int ulx, uly, urx, bry; /* UL.x, UL.y, UR.x, BR.y */
    int *keys, nkeys;
    int prev, next, seq, no_more;
} qnode;

typedef struct { qnode *nd; int n, cap, head, tail, size, seq; } qlist;

static int ql_new(qlist *l)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 256; l->nd = realloc(l->nd, sizeof(qnode) * l->cap); }
    qnode *q = &l->nd[l->n];
    memset(q, 0, sizeof *q);
    q->prev = q->next = -1;
    q->seq = l->seq++;
    return l->n++;
}
static void ql_push_back(qlist *l, int i)
{
    l->nd[i].prev = l->tail; l->nd[i].next = -1;
    if (l->tail >= 0) l->nd[l->tail].next = i; else l->head = i;
    l->tail = i; l->size++;
}
static void ql_push_front(qlist *l, int i)
{
    l->nd[i].next = l->head; l->nd[i].prev = -1;
    if (l->head >= 0) l->nd[l->head].prev = i; else l->tail = i;
    l->head = i; l->size++;
}
static int ql_erase(qlist *l, int i) /* returns next */
{
    int p = l->nd[i].prev, n = l->nd[i].next;
    if (p >= 0) l->nd[p].next = n; else l->head = n;
    if (n >= 0) l->nd[n].prev = p; else l->tail = p;
    l->size--;
    free(l->nd[i].keys); l->nd[i].keys = NULL;
    return n;
}

/* ExtractorNode::DivideNode, src/ORBextractor.cc:551-609.  Children are created in the list
 * arena (not yet linked); child[c] = -1 if it received no point. */
static void divide_node(qlist *l, int pi, const int *px, const int *py, int child[4])
{
    qnode P = l->nd[pi];
    int half_x = (int)ceilf((float)(P.urx - P.ulx) / 2);
    int half_y = (int)ceilf((float)(P.bry - P.uly) / 2);
    int cnt[4] = { 0, 0, 0, 0 };
    int *cls = malloc(sizeof(int) * (P.nkeys ? P.nkeys : 1));
    for (int i = 0; i < P.nkeys; i++) {
        float kx = (float)px[P.keys[i]], ky = (float)py[P.keys[i]];
        int c;
        if (kx < (float)(P.ulx + half_x)) c = ky < (float)(P.uly + half_y) ? 0 : 2;
        else c = ky < (float)(P.uly + half_y) ? 1 : 3;
        cls[i] = c; cnt[c]++;
    }
    for (int c = 0; c < 4; c++) {
        child[c] = -1;
        if (!cnt[c]) continue;
        int ci = ql_new(l);
        qnode *q = &l->nd[ci];
        P = l->nd[pi]; /* arena may have moved */
        q->ulx = (c & 1) ? P.ulx + half_x : P.ulx;
        q->urx = (c & 1) ? P.urx : P.ulx + half_x;
        q->uly = (c & 2) ? P.uly + half_y : P.uly;
        q->bry = (c & 2) ? P.bry : P.uly + half_y;
        q->keys = malloc(sizeof(int) * cnt[c]);
        q->nkeys = 0;
        for (int i = 0; i < P.nkeys; i++)
            if (cls[i] == c) q->keys[q->nkeys++] = P.keys[i];
        q->no_more = (q->nkeys == 1);
        child[c] = ci;
    }
    free(cls);
}

typedef struct { int count, seq, idx; } size_ptr;
static int size_ptr_cmp(const void *a, const void *b)
{
    const size_ptr *p = a, *q = b;
    if (p->count != q->count) return p->count < q->count ? -1 : 1;
    return p->seq < q->seq ? -1 : p->seq > q->seq ? 1 : 0;
}

/* ORBextractor::DistributeOctTree, src/ORBextractor.cc:617-915.
 * Points (x,y,resp) are relative to (minX,minY); out_idx receives the index of the kept
 * point of every leaf in final list order.  Returns the number of leaves. */
int oracle_distribute_octtree(const int *px, const int *py, const int *presp, int n,
                              int min_x, int max_x, int min_y, int max_y, int N, int *out_idx, int cap)
{
    qlist L = { 0 };
    L.head = L.tail = -1;
    const int n_ini = (int)roundf((float)(max_x - min_x) / (max_y - min_y));
    const float hx = (float)(max_x - min_x) / n_ini;
    if (n_ini < 1) return -1; /* reference divides by zero */
    int *roots = malloc(sizeof(int) * n_ini);
    for (int i = 0; i < n_ini; i++) {
        int r = ql_new(&L);
        qnode *q = &L.nd[r];
        q->ulx = (int)(hx * (float)i);
        q->urx = (int)(hx * (float)(i + 1));
        q->uly = 0;
        q->bry = max_y - min_y;
        q->keys = malloc(sizeof(int) * (n ? n : 1));
        ql_push_back(&L, r);
        roots[i] = r;
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)px[i] / hx); /* vpIniNodes[kp.pt.x/hX], :681 */
        if (r < 0 || r >= n_ini) { r = r < 0 ? 0 : n_ini - 1; } /* reference: out-of-bounds UB */
        qnode *q = &L.nd[roots[r]];
        q->keys[q->nkeys++] = i;
    }
    for (int it = L.head; it >= 0;) { /* :691-705 */
        if (L.nd[it].nkeys == 1) { L.nd[it].no_more = 1; it = L.nd[it].next; }
        else if (L.nd[it].nkeys == 0) it = ql_erase(&L, it);
        else it = L.nd[it].next;
    }
    free(roots);

    int finish = 0;
    size_ptr *vsz = NULL; int nvsz = 0, capvsz = 0;
#define VSZ_PUSH(ci) do { if (nvsz == capvsz) { capvsz = capvsz ? 2 * capvsz : 256; vsz = realloc(vsz, sizeof(size_ptr) * capvsz); } \
        vsz[nvsz].count = L.nd[ci].nkeys; vsz[nvsz].seq = L.nd[ci].seq; vsz[nvsz].idx = ci; nvsz++; } while (0)
    while (!finish) {
        int prev_size = L.size, n_to_expand = 0;
        nvsz = 0;
        for (int it = L.head; it >= 0;) { /* phase-1 sweep, :719-798 */
            if (L.nd[it].no_more) { it = L.nd[it].next; continue; }
            int ch[4];
            divide_node(&L, it, px, py, ch);
            for (int c = 0; c < 4; c++) {
                if (ch[c] < 0) continue;
                ql_push_front(&L, ch[c]);
                if (L.nd[ch[c]].nkeys > 1) { n_to_expand++; VSZ_PUSH(ch[c]); }
            }
            it = ql_erase(&L, it);
        }
        if (L.size >= N || L.size == prev_size) {
            finish = 1;
        } else if (L.size + n_to_expand * 3 > N) { /* phase 2, :814-886 */
            while (!finish) {
                prev_size = L.size;
                int nprev = nvsz;
                size_ptr *prev = malloc(sizeof(size_ptr) * (nprev ? nprev : 1));
                memcpy(prev, vsz, sizeof(size_ptr) * nprev);
                nvsz = 0;
                qsort(prev, nprev, sizeof(size_ptr), size_ptr_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    int ch[4];
                    divide_node(&L, prev[j].idx, px, py, ch);
                    for (int c = 0; c < 4; c++) {
                        if (ch[c] < 0) continue;
                        ql_push_front(&L, ch[c]);
                        if (L.nd[ch[c]].nkeys > 1) VSZ_PUSH(ch[c]);
                    }
                    ql_erase(&L, prev[j].idx);
                    if (L.size >= N) break;
                }
                free(prev);
                if (L.size >= N || L.size == prev_size) finish = 1;
            }
        }
    }
#undef VSZ_PUSH
    int nout = 0;
    for (int it = L.head; it >= 0; it = L.nd[it].next) { /* :895-912 */
        qnode *q = &L.nd[it];
        int best = q->keys[0];
        float max_resp = (float)presp[best];
        for (int k = 1; k < q->nkeys; k++)
            if ((float)presp[q->keys[k]] > max_resp) { best = q->keys[k]; max_resp = (float)presp[best]; }
        if (nout < cap) out_idx[nout] = best;
        nout++;
    }
    for (int i = 0; i < L.n; i++) free(L.nd[i].keys);
    free(L.nd); free(vsz);
    return nout;
}

/* ------------------------------------------------------------------ extractor object */

typedef struct { int w, h; uint8_t *pix, *blur; cand_vec cand; int nkp; } level_t;

struct orb_oracle {
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor; /* include/ORBextractor.h:117: stored as double, set from a float */
    float sf[MAX_LEVELS], isf[MAX_LEVELS], sig2[MAX_LEVELS], isig2[MAX_LEVELS];
    int quota[MAX_LEVELS];
    int umax[HALF_PATCH + 1];
    int cv_profile; /* which OpenCV generation's GaussianBlur taps: oracle_set_cv_profile */
    level_t lv[MAX_LEVELS];
};

int oracle_set_cv_profile(orb_oracle *o, int profile)
{
    if (!o || (profile != 0 && profile != 1)) return -1;
    o->cv_profile = profile;
    return 0;
}

orb_oracle *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > MAX_LEVELS || nfeatures < 0) return NULL;
    orb_oracle *o = calloc(1, sizeof *o);
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->ini_th = ini_th; o->min_th = min_th;
    o->scale_factor = scale_factor;
    /* :436-461 */
    o->sf[0] = 1.0f; o->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        o->sf[i] = (float)(o->sf[i - 1] * o->scale_factor);
        o->sig2[i] = o->sf[i] * o->sf[i];
    }
    for (int i = 0; i < nlevels; i++) {
        o->isf[i] = 1.0f / o->sf[i];
        o->isig2[i] = 1.0f / o->sig2[i];
    }
    /* :468-493 */
    float factor = (float)(1.0f / o->scale_factor);
    float n_desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        o->quota[l] = oracle_cv_round_f(n_desired);
        sum += o->quota[l];
        n_desired *= factor;
    }
    o->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* :510-533 */
    int v, v0, vmax = (int)floor(HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    return o;
}

static void level_free(level_t *l)
{
    free(l->pix); free(l->blur); free(l->cand.v);
    memset(l, 0, sizeof *l);
}

void oracle_destroy(orb_oracle *o)
{
    if (!o) return;
    for (int l = 0; l < o->nlevels; l++) level_free(&o->lv[l]);
    free(o);
}

int oracle_nlevels(const orb_oracle *o) { return o->nlevels; }
const float *oracle_scale_factors(const orb_oracle *o) { return o->sf; }
const float *oracle_inv_scale_factors(const orb_oracle *o) { return o->isf; }
const float *oracle_level_sigma2(const orb_oracle *o) { return o->sig2; }
const float *oracle_inv_level_sigma2(const orb_oracle *o) { return o->isig2; }
const int *oracle_features_per_level(const orb_oracle *o) { return o->quota; }
const int *oracle_umax(const orb_oracle *o) { return o->umax; }

int oracle_level_dims(const orb_oracle *o, int l, int *w, int *h)
{
    if (l < 0 || l >= o->nlevels || !o->lv[l].pix) return -1;
    *w = o->lv[l].w; *h = o->lv[l].h;
    return 0;
}
const uint8_t *oracle_level_pixels(const orb_oracle *o, int l) { return o->lv[l].pix; }
const uint8_t *oracle_level_blurred(const orb_oracle *o, int l) { return o->lv[l].blur; }
int oracle_level_nkeypoints(const orb_oracle *o, int l) { return o->lv[l].nkp; }
int oracle_level_candidates(const orb_oracle *o, int l, int *x, int *y, int *r, int cap)
{
    const cand_vec *c = &o->lv[l].cand;
    for (int i = 0; i < c->n && i < cap; i++) { x[i] = c->v[i].x; y[i] = c->v[i].y; r[i] = c->v[i].resp; }
    return c->n;
}

/* ORBextractor::ComputePyramid, src/ORBextractor.cc:1345-1394.  The 19-px border the
 * reference adds is never read downstream and is not materialised. */
static void compute_pyramid(orb_oracle *o, const uint8_t *img, int w, int h, size_t stride)
{
    for (int l = 0; l < o->nlevels; l++) {
        level_t *L = &o->lv[l];
        level_free(L);
        float scale = o->isf[l];
        L->w = oracle_cv_round_f((float)w * scale);
        L->h = oracle_cv_round_f((float)h * scale);
        L->pix = malloc((size_t)L->w * L->h);
        if (l == 0)
            for (int y = 0; y < h; y++) memcpy(L->pix + (size_t)y * w, img + (size_t)y * stride, w);
        else
            oracle_resize_linear(o->lv[l - 1].pix, o->lv[l - 1].w, o->lv[l - 1].h, o->lv[l - 1].w,
                                 L->pix, L->w, L->h, L->w);
    }
}

/* IC_Angle, src/ORBextractor.cc:83-111 */
static float ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return oracle_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, src/ORBextractor.cc:116-157 */
static void orb_descriptor(float angle_deg, const uint8_t *img, int stride, int x, int y, uint8_t *desc)
{
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factor_pi, a, b;
    oracle_sincos(angle, &b, &a); /* a = cos, b = sin */
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int p = 8 * i + k;
            float x0 = (float)ORB_PAT_X0[p], y0 = (float)ORB_PAT_Y0[p];
            float x1 = (float)ORB_PAT_X1[p], y1 = (float)ORB_PAT_Y1[p];
            int t0 = center[oracle_cv_round_f(x0 * b + y0 * a) * stride + oracle_cv_round_f(x0 * a - y0 * b)];
            int t1 = center[oracle_cv_round_f(x1 * b + y1 * a) * stride + oracle_cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

int oracle_extract(orb_oracle *o, const uint8_t *img, int w, int h, size_t stride,
                   oracle_kp *kps, uint8_t *desc, int cap)
{
    if (!o || !img || w <= 0 || h <= 0 || stride < (size_t)w) return -1;
    /* every level needs at least one 30-px cell in each direction (else the reference divides by 0) */
    for (int l = 0; l < o->nlevels; l++) {
        int lw = oracle_cv_round_f((float)w * o->isf[l]), lh = oracle_cv_round_f((float)h * o->isf[l]);
        if ((int)((float)(lw - 2 * EDGE_TH + 6) / 30.f) < 1 || (int)((float)(lh - 2 * EDGE_TH + 6) / 30.f) < 1) return -3;
    }
    compute_pyramid(o, img, w, h, stride);

    int total = 0;
    /* ComputeKeyPointsOctTree, src/ORBextractor.cc:925-1052 */
    const float W = 30;
    for (int level = 0; level < o->nlevels; ++level) {
        level_t *L = &o->lv[level];
        const int min_bx = EDGE_TH - 3, min_by = min_bx;
        const int max_bx = L->w - EDGE_TH + 3, max_by = L->h - EDGE_TH + 3;
        const float width = (float)(max_bx - min_bx), height = (float)(max_by - min_by);
        const int n_cols = (int)(width / W), n_rows = (int)(height / W);
        const int w_cell = (int)ceilf(width / n_cols), h_cell = (int)ceilf(height / n_rows);
        uint8_t *score = malloc((size_t)(w_cell + 6) * (h_cell + 6));
        cand_vec *C = &L->cand;
        for (int i = 0; i < n_rows; i++) {
            const float ini_y = (float)(min_by + i * h_cell);
            float max_y = ini_y + h_cell + 6;
            if (ini_y >= max_by - 3) continue;
            if (max_y > max_by) max_y = (float)max_by;
            for (int j = 0; j < n_cols; j++) {
                const float ini_x = (float)(min_bx + j * w_cell);
                float max_x = ini_x + w_cell + 6;
                if (ini_x >= max_bx - 6) continue;
                if (max_x > max_bx) max_x = (float)max_bx;
                int before = C->n;
                int nk = fast_cell(L->pix, L->w, (int)ini_x, (int)ini_y, (int)max_x, (int)max_y,
                                   o->ini_th, score, C, j * w_cell, i * h_cell);
                if (nk == 0) {
                    C->n = before;
                    fast_cell(L->pix, L->w, (int)ini_x, (int)ini_y, (int)max_x, (int)max_y,
                              o->min_th, score, C, j * w_cell, i * h_cell);
                }
            }
        }
        free(score);

        int n = C->n;
        int *px = malloc(sizeof(int) * (n + 1)), *py = malloc(sizeof(int) * (n + 1)), *pr = malloc(sizeof(int) * (n + 1));
        for (int i = 0; i < n; i++) { px[i] = C->v[i].x; py[i] = C->v[i].y; pr[i] = C->v[i].resp; }
        int kcap = n + 1; /* leaves never outnumber points */
        int *keep = malloc(sizeof(int) * kcap);
        int nk = n ? oracle_distribute_octtree(px, py, pr, n, min_bx, max_bx, min_by, max_by, o->quota[level], keep, kcap) : 0;
        if (nk < 0 || nk > kcap) { free(px); free(py); free(pr); free(keep); return -1; }
        L->nkp = nk;
        if (total + nk > cap) { free(px); free(py); free(pr); free(keep); return -2; }
        const int scaled_patch = (int)(PATCH_SIZE * o->sf[level]);
        for (int i = 0; i < nk; i++) {
            oracle_kp *k = &kps[total + i];
            k->x = (float)(px[keep[i]] + min_bx);
            k->y = (float)(py[keep[i]] + min_by);
            k->size = (float)scaled_patch;
            k->response = (float)pr[keep[i]];
            k->octave = level;
            k->class_id = -1;
            k->angle = ic_angle(L->pix, L->w, (int)k->x, (int)k->y, o->umax); /* computeOrientation :538-546 */
        }
        total += nk;
        free(px); free(py); free(pr); free(keep);
    }

    /* operator(): blur + descriptors + rescale, src/ORBextractor.cc:1302-1337 */
    int offset = 0;
    for (int level = 0; level < o->nlevels; ++level) {
        level_t *L = &o->lv[level];
        if (L->nkp == 0) continue;
        L->blur = malloc((size_t)L->w * L->h);
        oracle_gaussian_blur7_profile(L->pix, L->w, L->h, L->w, L->blur, L->w, o->cv_profile);
        for (int i = 0; i < L->nkp; i++) {
            oracle_kp *k = &kps[offset + i];
            orb_descriptor(k->angle, L->blur, L->w, oracle_cv_round_f(k->x), oracle_cv_round_f(k->y), desc + (size_t)(offset + i) * 32);
        }
        if (level != 0) {
            float scale = o->sf[level];
            for (int i = 0; i < L->nkp; i++) { kps[offset + i].x *= scale; kps[offset + i].y *= scale; }
        }
        offset += L->nkp;
    }
    return total;
}

/* ------------------------------------------------------------------ stereo (S1) */

/* pixel of the reference's padded pyramid image: inside = level pixel, outside = the
 * BORDER_REFLECT_101 margin written by copyMakeBorder (src/ORBextractor.cc:1370-1383) */
static int lvl_px(const level_t *L, int x, int y)
{
    return L->pix[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)];
}

typedef struct { int dist, il; } dist_idx;
static int dist_idx_cmp(const void *a, const void *b)
{
    const dist_idx *p = a, *q = b;
    if (p->dist != q->dist) return p->dist < q->dist ? -1 : 1;
    return p->il < q->il ? -1 : p->il > q->il ? 1 : 0;
}

/* Frame::ComputeStereoMatches, src/Frame.cc:577-751.  `mb` (read uninitialised in the
 * reference) is the explicit min_z argument. */
int oracle_stereo_match(const orb_oracle *LE, const orb_oracle *RE,
                        const oracle_kp *kL, const uint8_t *dL, int nL,
                        const oracle_kp *kR, const uint8_t *dR, int nR,
                        float bf, float min_z, float *u_right, float *depth)
{
    for (int i = 0; i < nL; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    const int th_orb = (TH_HIGH + TH_LOW) / 2;
    const int n_rows = LE->lv[0].h;
    /* row table :584-604 */
    int *cnt = calloc(n_rows + 1, sizeof(int));
    int **rows = calloc(n_rows, sizeof(int *));
    int *rcap = calloc(n_rows, sizeof(int));
    for (int ir = 0; ir < nR; ir++) {
        const float ky = kR[ir].y;
        const float r = 2.0f * RE->sf[kR[ir].octave]; /* mvScaleFactors are the left extractor's; identical tables */
        const int maxr = (int)ceilf(ky + r), minr = (int)floorf(ky - r);
        for (int yi = minr; yi <= maxr; yi++) {
            if (yi < 0 || yi >= n_rows) continue; /* reference: out-of-bounds UB */
            if (cnt[yi] == rcap[yi]) { rcap[yi] = rcap[yi] ? 2 * rcap[yi] : 16; rows[yi] = realloc(rows[yi], sizeof(int) * rcap[yi]); }
            rows[yi][cnt[yi]++] = ir;
        }
    }
    const float min_d = 0, max_d = bf / min_z;
    dist_idx *vd = malloc(sizeof(dist_idx) * (nL ? nL : 1));
    int nvd = 0;
    for (int il = 0; il < nL; il++) {
        const int level_l = kL[il].octave;
        const float vl = kL[il].y, ul = kL[il].x;
        const int row = (int)vl;
        if (row < 0 || row >= n_rows || cnt[row] == 0) continue;
        const float min_u = ul - max_d, max_u = ul - min_d;
        if (max_u < 0) continue;
        int best_dist = TH_HIGH, best_r = 0;
        for (int ic = 0; ic < cnt[row]; ic++) {
            const int ir = rows[row][ic];
            if (kR[ir].octave < level_l - 1 || kR[ir].octave > level_l + 1) continue;
            const float ur = kR[ir].x;
            if (ur >= min_u && ur <= max_u) {
                const int dist = oracle_hamming(dL + (size_t)il * 32, dR + (size_t)ir * 32);
                if (dist < best_dist) { best_dist = dist; best_r = ir; }
            }
        }
        if (best_dist < th_orb) {
            const float ur0 = kR[best_r].x;
            const float sfac = LE->isf[level_l];
            const float sul = roundf(kL[il].x * sfac), svl = roundf(kL[il].y * sfac), sur0 = roundf(ur0 * sfac);
            const int w = 5, Lw = 5;
            const level_t *PL = &LE->lv[level_l], *PR = &RE->lv[level_l];
            int il_win[11][11];
            const int cy = (int)svl, cxl = (int)sul;
            for (int dy = -w; dy <= w; dy++)
                for (int dx = -w; dx <= w; dx++)
                    il_win[dy + w][dx + w] = lvl_px(PL, cxl + dx, cy + dy) - lvl_px(PL, cxl, cy);
            int best_sad = INT_MAX, best_inc = 0;
            float vdists[11];
            const float iniu = sur0 + Lw - w, endu = sur0 + Lw + w + 1;
            if (iniu < 0 || endu >= (float)PR->w) continue;
            for (int inc = -Lw; inc <= Lw; inc++) {
                const int cxr = (int)(sur0 + (float)inc);
                float dist = 0;
                const int cr = lvl_px(PR, cxr, cy);
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++)
                        dist += fabsf((float)(il_win[dy + w][dx + w] - (lvl_px(PR, cxr + dx, cy + dy) - cr)));
                if (dist < (float)best_sad) { best_sad = (int)dist; best_inc = inc; }
                vdists[Lw + inc] = dist;
            }
            if (best_inc == -Lw || best_inc == Lw) continue;
            const float d1 = vdists[Lw + best_inc - 1], d2 = vdists[Lw + best_inc], d3 = vdists[Lw + best_inc + 1];
            const float delta = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
            if (delta < -1 || delta > 1) continue;
            float best_ur = LE->sf[level_l] * ((float)sur0 + (float)best_inc + delta);
            float disparity = ul - best_ur;
            if (disparity >= min_d && disparity < max_d) {
                if (disparity <= 0) { disparity = 0.01f; best_ur = (float)(ul - 0.01); }
                depth[il] = bf / disparity;
                u_right[il] = best_ur;
                vd[nvd].dist = best_sad; vd[nvd].il = il; nvd++;
            }
        }
    }
    if (nvd > 0) { /* :737-750; the reference indexes an empty vector when nvd==0 */
        qsort(vd, nvd, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)vd[nvd / 2].dist;
        const float th_dist = 1.5f * 1.4f * median;
        for (int i = nvd - 1; i >= 0; i--) {
            if ((float)vd[i].dist < th_dist) break;
            u_right[vd[i].il] = -1; depth[vd[i].il] = -1;
        }
    }
    for (int i = 0; i < n_rows; i++) free(rows[i]);
    free(rows); free(rcap); free(cnt); free(vd);
    return 0;
}

/* ------------------------------------------------------------------ BoW matchers (M2-M5) */

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1687-1728 */
void oracle_three_maxima(const int *count, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = count[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

typedef struct { int *v[HISTO_LENGTH]; int n[HISTO_LENGTH], cap[HISTO_LENGTH]; } rot_hist;
static void rh_push(rot_hist *h, int bin, int val)
{
    if (h->n[bin] == h->cap[bin]) { h->cap[bin] = h->cap[bin] ? 2 * h->cap[bin] : 64; h->v[bin] = realloc(h->v[bin], sizeof(int) * h->cap[bin]); }
    h->v[bin][h->n[bin]++] = val;
}
static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
/* clear matches outside the three dominant bins; returns how many were removed */
static int rh_filter_mark(rot_hist *h, int32_t *match, int32_t cleared)
{
    int i1, i2, i3, removed = 0;
    oracle_three_maxima(h->n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i != i1 && i != i2 && i != i3)
            for (int j = 0; j < h->n[i]; j++) { match[h->v[i][j]] = cleared; removed++; }
        free(h->v[i]);
    }
    return removed;
}
static int rh_filter(rot_hist *h, int32_t *match) { return rh_filter_mark(h, match, -1); }

/* merge-join cursor advance == std::map::lower_bound on ascending ids */
static int lower_bound_u32(const uint32_t *a, int n, uint32_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

/* src/ORBmatcher.cc:171-303 */
int oracle_search_by_bow_kf_f(const oracle_featset *kf, const oracle_featset *f,
                              float nnratio, int check_ori, int32_t *match_f)
{
    for (int i = 0; i < f->n; i++) match_f[i] = -1;
    int nmatches = 0;
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int a = 0, b = 0;
    while (a < kf->nnodes && b < f->nnodes) {
        if (kf->node_id[a] == f->node_id[b]) {
            for (int ik = kf->node_off[a]; ik < kf->node_off[a + 1]; ik++) {
                const int ridx_kf = (int)kf->feat[ik];
                if (!kf->flag[ridx_kf]) continue;
                int best1 = 256, best_idx = -1, best2 = 256;
                for (int jf = f->node_off[b]; jf < f->node_off[b + 1]; jf++) {
                    const int ridx_f = (int)f->feat[jf];
                    if (match_f[ridx_f] >= 0) continue;
                    const int dist = oracle_hamming(kf->desc + (size_t)ridx_kf * 32, f->desc + (size_t)ridx_f * 32);
                    if (dist < best1) { best2 = best1; best1 = dist; best_idx = ridx_f; }
                    else if (dist < best2) best2 = dist;
                }
                if (best1 <= TH_LOW && (float)best1 < nnratio * (float)best2) {
                    match_f[best_idx] = ridx_kf;
                    if (check_ori) rh_push(&rh, rot_bin(kf->angle[ridx_kf], f->angle[best_idx]), best_idx);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (kf->node_id[a] < f->node_id[b]) a = lower_bound_u32(kf->node_id, kf->nnodes, f->node_id[b]);
        else b = lower_bound_u32(f->node_id, f->nnodes, kf->node_id[a]);
    }
    if (check_ori) nmatches -= rh_filter(&rh, match_f);
    return nmatches;
}

/* src/ORBmatcher.cc:568-702 */
int oracle_search_by_bow_kf_kf(const oracle_featset *k1, const oracle_featset *k2,
                               float nnratio, int check_ori, int32_t *match12)
{
    for (int i = 0; i < k1->n; i++) match12[i] = -1;
    uint8_t *matched2 = calloc(k2->n ? k2->n : 1, 1);
    int nmatches = 0;
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int a = 0, b = 0;
    while (a < k1->nnodes && b < k2->nnodes) {
        if (k1->node_id[a] == k2->node_id[b]) {
            for (int i1 = k1->node_off[a]; i1 < k1->node_off[a + 1]; i1++) {
                const int idx1 = (int)k1->feat[i1];
                if (!k1->flag[idx1]) continue;
                int best1 = 256, best_idx2 = -1, best2 = 256;
                for (int i2 = k2->node_off[b]; i2 < k2->node_off[b + 1]; i2++) {
                    const int idx2 = (int)k2->feat[i2];
                    if (matched2[idx2] || !k2->flag[idx2]) continue;
                    const int dist = oracle_hamming(k1->desc + (size_t)idx1 * 32, k2->desc + (size_t)idx2 * 32);
                    if (dist < best1) { best2 = best1; best1 = dist; best_idx2 = idx2; }
                    else if (dist < best2) best2 = dist;
                }
                if (best1 < TH_LOW && (float)best1 < nnratio * (float)best2) {
                    match12[idx1] = best_idx2;
                    matched2[best_idx2] = 1;
                    if (check_ori) rh_push(&rh, rot_bin(k1->angle[idx1], k2->angle[best_idx2]), idx1);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (k1->node_id[a] < k2->node_id[b]) a = lower_bound_u32(k1->node_id, k1->nnodes, k2->node_id[b]);
        else b = lower_bound_u32(k2->node_id, k2->nnodes, k1->node_id[a]);
    }
    if (check_ori) nmatches -= rh_filter(&rh, match12);
    free(matched2);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:147-164 */
static int check_dist_epipolar(float x1, float y1, float x2, float y2, const float *F12, float sigma2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2;
}

/* src/ORBmatcher.cc:704-871 */
int oracle_search_for_triangulation(const oracle_featset *k1, const oracle_featset *k2,
                                    const float F12[9], float ex, float ey,
                                    const float *sf2, const float *sig2_2,
                                    float nnratio, int check_ori, int only_stereo,
                                    int32_t *pairs, int cap)
{
    (void)nnratio; /* the reference never applies mfNNratio in this search */
    int nmatches = 0;
    int32_t *m12 = malloc(sizeof(int32_t) * (k1->n ? k1->n : 1));
    for (int i = 0; i < k1->n; i++) m12[i] = -1;
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int a = 0, b = 0;
    while (a < k1->nnodes && b < k2->nnodes) {
        if (k1->node_id[a] == k2->node_id[b]) {
            for (int i1 = k1->node_off[a]; i1 < k1->node_off[a + 1]; i1++) {
                const int idx1 = (int)k1->feat[i1];
                if (k1->flag[idx1]) continue;
                const int stereo1 = k1->u_right[idx1] >= 0;
                if (only_stereo && !stereo1) continue;
                int best_dist = TH_LOW, best_idx2 = -1;
                for (int i2 = k2->node_off[b]; i2 < k2->node_off[b + 1]; i2++) {
                    const int idx2 = (int)k2->feat[i2];
                    if (k2->flag[idx2]) continue; /* vbMatched2 is never set in the reference */
                    const int stereo2 = k2->u_right[idx2] >= 0;
                    if (only_stereo && !stereo2) continue;
                    const int dist = oracle_hamming(k1->desc + (size_t)idx1 * 32, k2->desc + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > best_dist) continue;
                    if (!stereo1 && !stereo2) {
                        const float distex = ex - k2->x[idx2], distey = ey - k2->y[idx2];
                        if (distex * distex + distey * distey < 100 * sf2[k2->octave[idx2]]) continue;
                    }
                    if (check_dist_epipolar(k1->x[idx1], k1->y[idx1], k2->x[idx2], k2->y[idx2], F12, sig2_2[k2->octave[idx2]])) {
                        best_idx2 = idx2; best_dist = dist;
                    }
                }
                if (best_idx2 >= 0) {
                    m12[idx1] = best_idx2;
                    nmatches++;
                    if (check_ori) rh_push(&rh, rot_bin(k1->angle[idx1], k2->angle[best_idx2]), idx1);
                }
            }
            a++; b++;
        } else if (k1->node_id[a] < k2->node_id[b]) a = lower_bound_u32(k1->node_id, k1->nnodes, k2->node_id[b]);
        else b = lower_bound_u32(k2->node_id, k2->nnodes, k1->node_id[a]);
    }
    if (check_ori) nmatches -= rh_filter(&rh, m12);
    int np = 0;
    for (int i = 0; i < k1->n; i++) {
        if (m12[i] < 0) continue;
        if (np < cap) { pairs[2 * np] = i; pairs[2 * np + 1] = m12[i]; }
        np++;
    }
    free(m12);
    return np;
}

/* ------------------------------------------------------------------ DBoW2 vocabulary + transform (f2) */

struct oracle_vocab {
    int k, L, nnodes, nwords;  /* nnodes includes the root */
    int *parent, *child_off, *child_ids, *word_id;
    uint8_t *desc;             /* [nnodes][32]; root's is zero */
    double *weight;
};

static oracle_vocab *vocab_finish(oracle_vocab *v, const uint8_t *is_leaf)
{
    /* children lists in id order (loadFromTextFile pushes nid onto m_nodes[pid].children as it reads) */
    int n = v->nnodes;
    v->child_off = calloc(n + 1, sizeof(int));
    v->child_ids = malloc(sizeof(int) * (n > 1 ? n - 1 : 1));
    v->word_id = malloc(sizeof(int) * n);
    for (int i = 1; i < n; i++) {
        if (v->parent[i] < 0 || v->parent[i] >= i) { oracle_vocab_destroy(v); return NULL; } /* parents precede children */
        v->child_off[v->parent[i] + 1]++;
    }
    for (int i = 0; i < n; i++) v->child_off[i + 1] += v->child_off[i];
    int *cur = malloc(sizeof(int) * n);
    memcpy(cur, v->child_off, sizeof(int) * n);
    for (int i = 1; i < n; i++) v->child_ids[cur[v->parent[i]]++] = i;
    free(cur);
    v->nwords = 0;
    v->word_id[0] = -1;
    for (int i = 1; i < n; i++) v->word_id[i] = is_leaf[i - 1] ? v->nwords++ : -1; /* :1425-1432 */
    return v;
}

oracle_vocab *oracle_vocab_create(int k, int L, int nm1, const int32_t *parent, const uint8_t *is_leaf,
                                  const uint8_t *desc, const double *weight)
{
    if (k < 0 || k > 20 || L < 1 || L > 10 || nm1 < 1) return NULL; /* loader limits :1379 */
    oracle_vocab *v = calloc(1, sizeof *v);
    v->k = k; v->L = L; v->nnodes = nm1 + 1;
    v->parent = malloc(sizeof(int) * v->nnodes);
    v->desc = calloc((size_t)v->nnodes, 32);
    v->weight = calloc(v->nnodes, sizeof(double));
    v->parent[0] = -1;
    for (int i = 0; i < nm1; i++) { v->parent[i + 1] = parent[i]; v->weight[i + 1] = weight[i]; }
    memcpy(v->desc + 32, desc, (size_t)nm1 * 32);
    return vocab_finish(v, is_leaf);
}

/* text format of loadFromTextFile: header "k L scoring weighting", then per node "parent isLeaf d0 .. d31 weight" */
oracle_vocab *oracle_vocab_load_text(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    int k, L, n1, n2;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { fclose(f); return NULL; }
    int cap = 1024, n = 0;
    int32_t *parent = malloc(sizeof(int32_t) * cap);
    uint8_t *leaf = malloc(cap), *desc = malloc((size_t)cap * 32);
    double *w = malloc(sizeof(double) * cap);
    for (;;) {
        int pid, isl;
        if (fscanf(f, "%d %d", &pid, &isl) != 2) break;
        if (n == cap) { cap *= 2; parent = realloc(parent, sizeof(int32_t) * cap); leaf = realloc(leaf, cap); desc = realloc(desc, (size_t)cap * 32); w = realloc(w, sizeof(double) * cap); }
        int ok = 1;
        for (int i = 0; i < 32; i++) { int b; if (fscanf(f, "%d", &b) != 1) { ok = 0; break; } desc[(size_t)n * 32 + i] = (uint8_t)b; }
        if (!ok || fscanf(f, "%lf", &w[n]) != 1) break;
        parent[n] = pid; leaf[n] = isl > 0;
        n++;
    }
    fclose(f);
    oracle_vocab *v = n ? oracle_vocab_create(k, L, n, parent, leaf, desc, w) : NULL;
    free(parent); free(leaf); free(desc); free(w);
    return v;
}

void oracle_vocab_destroy(oracle_vocab *v)
{
    if (!v) return;
    free(v->parent); free(v->child_off); free(v->child_ids); free(v->word_id); free(v->desc); free(v->weight);
    free(v);
}
int oracle_vocab_nodes(const oracle_vocab *v) { return v->nnodes; }
int oracle_vocab_words(const oracle_vocab *v) { return v->nwords; }

typedef struct { uint32_t key; int feat; } kf_pair;
static int kf_cmp(const void *a, const void *b)
{
    const kf_pair *p = a, *q = b;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->feat < q->feat ? -1 : p->feat > q->feat ? 1 : 0;
}

/* The accumulation half of TemplatedVocabulary::transform (:1147-1165): for every feature i in order whose word
 * weight is > 0 (:1157), v.addWeight(word_id[i], weight[i]) (BowVector.cpp:33-45: the map value is incremented in
 * FEATURE order, which fixes the fp64 summation order) and fv.addFeature(node_id[i], i) (FeatureVector.cpp:31-45),
 * then v.normalize(L1) (BowVector.cpp:58-77).  Pinned against the reference's own BowVector.cpp / FeatureVector.cpp
 * compiled into oracle/_ref (tests/test_dbow2_ref.py). */
int oracle_bow_accumulate(const uint32_t *word_id, const double *word_weight, const uint32_t *node_id, int n,
                          uint32_t *bow_id, double *bow_val, int *nbow,
                          uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes)
{
    kf_pair *words = malloc(sizeof(kf_pair) * (n ? n : 1)), *nodes = malloc(sizeof(kf_pair) * (n ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (word_weight[i] > 0) { /* not stopped, :1157 */
            words[m].key = word_id[i]; words[m].feat = i;
            nodes[m].key = node_id[i]; nodes[m].feat = i;
            m++;
        }
    /* BowVector: std::map<WordId, WordValue> with addWeight in feature order (BowVector.cpp:33-45) */
    qsort(words, m, sizeof(kf_pair), kf_cmp);
    int nb = 0;
    for (int i = 0; i < m;) {
        int j = i;
        double acc = word_weight[words[i].feat];
        for (j = i + 1; j < m && words[j].key == words[i].key; j++) acc += word_weight[words[j].feat];
        bow_id[nb] = words[i].key; bow_val[nb] = acc; nb++;
        i = j;
    }
    /* L1 normalisation (TF_IDF with L1_NORM: mustNormalize), BowVector.cpp:58-77 */
    double norm = 0.0;
    for (int i = 0; i < nb; i++) norm += fabs(bow_val[i]);
    if (norm > 0.0) for (int i = 0; i < nb; i++) bow_val[i] /= norm;
    *nbow = nb;
    /* FeatureVector: std::map<NodeId, vector<unsigned>> with addFeature in feature order */
    qsort(nodes, m, sizeof(kf_pair), kf_cmp);
    int nn = 0;
    for (int i = 0; i < m; i++) {
        if (i == 0 || nodes[i].key != nodes[i - 1].key) { fv_node_id[nn] = nodes[i].key; fv_node_off[nn] = i; nn++; }
        fv_feat[i] = (uint32_t)nodes[i].feat;
    }
    fv_node_off[nn] = m;
    *fv_nnodes = nn;
    free(words); free(nodes);
    return 0;
}

int oracle_bow_transform(const oracle_vocab *v, const uint8_t *desc, int n, int levelsup,
                         uint32_t *word_id, double *word_weight, uint32_t *node_id,
                         uint32_t *bow_id, double *bow_val, int *nbow,
                         uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes)
{
    uint32_t *wid_all = malloc(sizeof(uint32_t) * (n ? n : 1)), *nid_all = malloc(sizeof(uint32_t) * (n ? n : 1));
    double *wts = malloc(sizeof(double) * (n ? n : 1));
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; i++) {
        /* transform(feature, word_id, weight, nid, levelsup), :1218-1259 */
        const uint8_t *f = desc + (size_t)i * 32;
        int nid = 0, final_id = 0, level = 0;
        do {
            ++level;
            const int c0 = v->child_off[final_id], c1 = v->child_off[final_id + 1];
            final_id = v->child_ids[c0];
            double best_d = oracle_hamming(f, v->desc + (size_t)final_id * 32);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = v->child_ids[c];
                const double d = oracle_hamming(f, v->desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (level == nid_level) nid = final_id;
        } while (v->child_off[final_id + 1] > v->child_off[final_id]);
        wid_all[i] = (uint32_t)v->word_id[final_id];
        wts[i] = v->weight[final_id];
        nid_all[i] = (uint32_t)nid;
        if (word_id) word_id[i] = wid_all[i];
        if (word_weight) word_weight[i] = wts[i];
        if (node_id) node_id[i] = nid_all[i];
    }
    const int rc = oracle_bow_accumulate(wid_all, wts, nid_all, n, bow_id, bow_val, nbow, fv_node_id, fv_node_off, fv_feat, fv_nnodes);
    free(wid_all); free(nid_all); free(wts);
    return rc;
}

/* ------------------------------------------------------------------ MapPoint::ComputeDistinctiveDescriptors (f3) */

static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* src/MapPoint.cc:266-340 */
int oracle_distinctive_descriptor(const uint8_t *desc, int n)
{
    if (n <= 0) return -1;
    float *dist = malloc(sizeof(float) * (size_t)n * n);
    for (int i = 0; i < n; i++) {
        dist[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; j++) {
            const int d = oracle_hamming(desc + (size_t)i * 32, desc + (size_t)j * 32);
            dist[(size_t)i * n + j] = (float)d;
            dist[(size_t)j * n + i] = (float)d;
        }
    }
    int best_median = INT_MAX, best_idx = 0;
    int *row = malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) row[j] = (int)dist[(size_t)i * n + j]; /* vector<int> vDists(float*, float*) */
        qsort(row, n, sizeof(int), int_cmp);
        const int median = row[(int)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best_idx = i; }
    }
    free(row); free(dist);
    return best_idx;
}

/* ------------------------------------------------------------------ projection-guided searches (f1) */

enum { GRID_COLS = 64, GRID_ROWS = 48 }; /* include/Frame.h:37-38 */

typedef struct { int *idx; int n, cap; } gcell;
typedef struct { gcell c[GRID_COLS][GRID_ROWS]; float inv_w, inv_h; } fgrid;

/* Frame::AssignFeaturesToGrid + PosInGrid, src/Frame.cc:261-279, :444-457 */
static fgrid *grid_build(const oracle_frame_feats *f)
{
    fgrid *g = calloc(1, sizeof *g);
    g->inv_w = (float)GRID_COLS / (f->max_x - f->min_x); /* :164-165 */
    g->inv_h = (float)GRID_ROWS / (f->max_y - f->min_y);
    for (int i = 0; i < f->n; i++) {
        const int px = (int)roundf((f->x[i] - f->min_x) * g->inv_w);
        const int py = (int)roundf((f->y[i] - f->min_y) * g->inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        gcell *c = &g->c[px][py];
        if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 4; c->idx = realloc(c->idx, sizeof(int) * c->cap); }
        c->idx[c->n++] = i;
    }
    return g;
}
static void grid_free(fgrid *g)
{
    for (int i = 0; i < GRID_COLS; i++) for (int j = 0; j < GRID_ROWS; j++) free(g->c[i][j].idx);
    free(g);
}
/* Frame::GetFeaturesInArea, src/Frame.cc:386-442; returns the number of indices written to out (capacity f->n) */
static int features_in_area(const oracle_frame_feats *f, const fgrid *g, float x, float y, float r, int min_level, int max_level, int *out)
{
    int n = 0;
    const int min_cx = (int)floorf((x - f->min_x - r) * g->inv_w) > 0 ? (int)floorf((x - f->min_x - r) * g->inv_w) : 0;
    if (min_cx >= GRID_COLS) return 0;
    int max_cx = (int)ceilf((x - f->min_x + r) * g->inv_w);
    if (max_cx > GRID_COLS - 1) max_cx = GRID_COLS - 1;
    if (max_cx < 0) return 0;
    const int min_cy = (int)floorf((y - f->min_y - r) * g->inv_h) > 0 ? (int)floorf((y - f->min_y - r) * g->inv_h) : 0;
    if (min_cy >= GRID_ROWS) return 0;
    int max_cy = (int)ceilf((y - f->min_y + r) * g->inv_h);
    if (max_cy > GRID_ROWS - 1) max_cy = GRID_ROWS - 1;
    if (max_cy < 0) return 0;
    const int check_levels = (min_level > 0) || (max_level >= 0);
    for (int ix = min_cx; ix <= max_cx; ix++)
        for (int iy = min_cy; iy <= max_cy; iy++) {
            const gcell *c = &g->c[ix][iy];
            for (int j = 0; j < c->n; j++) {
                const int k = c->idx[j];
                if (check_levels) {
                    if (f->octave[k] < min_level) continue;
                    if (max_level >= 0 && f->octave[k] > max_level) continue;
                }
                const float distx = f->x[k] - x, disty = f->y[k] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out[n++] = k;
            }
        }
    return n;
}

/* src/ORBmatcher.cc:1396-1553 */
int oracle_search_by_projection_last(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *sf,
                                     float th, int direction, float mbf, int check_ori, int32_t *match_cur)
{
    fgrid *g = grid_build(cur);
    int *cand = malloc(sizeof(int) * (cur->n ? cur->n : 1));
    uint8_t *blocked = malloc(cur->n ? cur->n : 1); /* current holder has Observations() > 0 */
    for (int i = 0; i < cur->n; i++) { match_cur[i] = -1; blocked[i] = cur->occupied[i]; }
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int nmatches = 0;
    for (int i = 0; i < pts->n; i++) {
        if (!pts->valid[i]) continue;
        const float invzc = pts->aux[i];
        if (invzc < 0) continue;
        const float u = pts->u[i], v = pts->v[i];
        if (u < cur->min_x || u > cur->max_x) continue;
        if (v < cur->min_y || v > cur->max_y) continue;
        const int oct = pts->level[i];
        const float radius = th * sf[oct];
        int nc;
        if (direction == 1) nc = features_in_area(cur, g, u, v, radius, oct, -1, cand);
        else if (direction == 2) nc = features_in_area(cur, g, u, v, radius, 0, oct, cand);
        else nc = features_in_area(cur, g, u, v, radius, oct - 1, oct + 1, cand);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (blocked[i2]) continue;
            if (cur->u_right[i2] > 0) {
                const float ur = u - mbf * invzc;
                const float er = fabsf(ur - cur->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = oracle_hamming(pts->desc + (size_t)i * 32, cur->desc + (size_t)i2 * 32);
            if (dist < best_dist) { best_dist = dist; best_idx = i2; }
        }
        if (best_dist <= TH_HIGH) {
            match_cur[best_idx] = i;
            blocked[best_idx] = pts->has_obs[i];
            nmatches++;
            if (check_ori & 1) rh_push(&rh, rot_bin(pts->angle[i], cur->angle[best_idx]), best_idx);
        }
    }
    if (check_ori & 1) nmatches -= rh_filter_mark(&rh, match_cur, (check_ori & 2) ? -2 : -1);
    grid_free(g); free(cand); free(blocked);
    return nmatches;
}

/* src/ORBmatcher.cc:48-129 (+ RadiusByViewingCos :131-137) */
int oracle_search_by_projection_points(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *sf,
                                       float th, float nnratio, int32_t *match_cur)
{
    fgrid *g = grid_build(cur);
    int *cand = malloc(sizeof(int) * (cur->n ? cur->n : 1));
    uint8_t *blocked = malloc(cur->n ? cur->n : 1);
    for (int i = 0; i < cur->n; i++) { match_cur[i] = -1; blocked[i] = cur->occupied[i]; }
    int nmatches = 0;
    const int b_factor = (double)th != 1.0;
    for (int i = 0; i < pts->n; i++) {
        if (!pts->valid[i]) continue;
        const int level = pts->level[i];
        float r = (double)pts->view_cos[i] > 0.998 ? 2.5f : 4.0f;
        if (b_factor) r *= th;
        const int nc = features_in_area(cur, g, pts->u[i], pts->v[i], r * sf[level], level - 1, level, cand);
        if (nc == 0) continue;
        int best_dist = 256, best_level = -1, best_dist2 = 256, best_level2 = -1, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (blocked[idx]) continue;
            if (cur->u_right[idx] > 0) {
                const float er = fabsf(pts->aux[i] - cur->u_right[idx]);
                if (er > r * sf[level]) continue;
            }
            const int dist = oracle_hamming(pts->desc + (size_t)i * 32, cur->desc + (size_t)idx * 32);
            if (dist < best_dist) {
                best_dist2 = best_dist; best_dist = dist;
                best_level2 = best_level; best_level = cur->octave[idx];
                best_idx = idx;
            } else if (dist < best_dist2) {
                best_level2 = cur->octave[idx];
                best_dist2 = dist;
            }
        }
        if (best_dist <= TH_HIGH) {
            if (best_level == best_level2 && (float)best_dist > nnratio * (float)best_dist2) continue;
            match_cur[best_idx] = i;
            blocked[best_idx] = pts->has_obs[i];
            nmatches++;
        }
    }
    grid_free(g); free(cand); free(blocked);
    return nmatches;
}

/* src/ORBmatcher.cc:1555-1685 (Tracking::Relocalization): points = pKF's map points in feature order, the adaptor
 * projects them with the current pose, applies the isBad / sAlreadyFound / distance-range tests (-> valid) and predicts the
 * level.  Any accepted match blocks its feature (mvpMapPoints[i2] != NULL, :1624-1625). */
int oracle_search_by_projection_keyframe(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *sf,
                                         float th, int orb_dist, int check_ori, int32_t *match_cur)
{
    fgrid *g = grid_build(cur);
    int *cand = malloc(sizeof(int) * (cur->n ? cur->n : 1));
    uint8_t *blocked = malloc(cur->n ? cur->n : 1);
    for (int i = 0; i < cur->n; i++) { match_cur[i] = -1; blocked[i] = cur->occupied[i]; }
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int nmatches = 0;
    for (int i = 0; i < pts->n; i++) {
        if (!pts->valid[i]) continue;
        const float u = pts->u[i], v = pts->v[i];
        if (u < cur->min_x || u > cur->max_x) continue;   /* :1591-1594 */
        if (v < cur->min_y || v > cur->max_y) continue;
        const int lvl = pts->level[i];
        const float radius = th * sf[lvl];                /* :1610 */
        const int nc = features_in_area(cur, g, u, v, radius, lvl - 1, lvl + 1, cand);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (blocked[i2]) continue;
            const int dist = oracle_hamming(pts->desc + (size_t)i * 32, cur->desc + (size_t)i2 * 32);
            if (dist < best_dist) { best_dist = dist; best_idx = i2; }
        }
        if (best_dist <= orb_dist) {
            match_cur[best_idx] = i;
            blocked[best_idx] = 1;
            nmatches++;
            if (check_ori & 1) rh_push(&rh, rot_bin(pts->angle[i], cur->angle[best_idx]), best_idx);
        }
    }
    if (check_ori & 1) nmatches -= rh_filter_mark(&rh, match_cur, (check_ori & 2) ? -2 : -1);
    grid_free(g); free(cand); free(blocked);
    return nmatches;
}

/* KeyFrame::IsInImage, src/KeyFrame.cc:649-652 */
static int kf_in_image(const oracle_frame_feats *kf, float x, float y)
{
    return x >= kf->min_x && x < kf->max_x && y >= kf->min_y && y < kf->max_y;
}

/* src/ORBmatcher.cc:305-415 (LoopClosing::ComputeSim3): the adaptor projects vpPoints with Scw and applies the isBad /
 * spAlreadyFound / depth / distance-range / viewing-angle tests (-> valid); occupied = vpMatched[idx] != NULL on entry.
 * match_kf[idx] = index of the point written to vpMatched[idx] by this call, or -1. */
int oracle_search_by_projection_sim3(const oracle_frame_feats *kf, const oracle_proj_points *pts, const float *sf,
                                     float th, int32_t *match_kf)
{
    fgrid *g = grid_build(kf);
    int *cand = malloc(sizeof(int) * (kf->n ? kf->n : 1));
    uint8_t *blocked = malloc(kf->n ? kf->n : 1);
    for (int i = 0; i < kf->n; i++) { match_kf[i] = -1; blocked[i] = kf->occupied[i]; }
    int nmatches = 0;
    for (int i = 0; i < pts->n; i++) {
        if (!pts->valid[i]) continue;
        const float u = pts->u[i], v = pts->v[i];
        if (!kf_in_image(kf, u, v)) continue;             /* :355-356 */
        const int lvl = pts->level[i];
        const float radius = th * sf[lvl];                /* :377 */
        const int nc = features_in_area(kf, g, u, v, radius, -1, -1, cand); /* KeyFrame::GetFeaturesInArea: no level test */
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (blocked[idx]) continue;
            const int kl = kf->octave[idx];
            if (kl < lvl - 1 || kl > lvl) continue;
            const int dist = oracle_hamming(pts->desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
            if (dist < best_dist) { best_dist = dist; best_idx = idx; }
        }
        if (best_dist <= TH_LOW) {
            match_kf[best_idx] = i;
            blocked[best_idx] = 1;
            nmatches++;
        }
    }
    grid_free(g); free(cand); free(blocked);
    return nmatches;
}

/* The search half of both ORBmatcher::Fuse overloads (src/ORBmatcher.cc:873-1038 with chi2 = 1, :1040-1164 with
 * chi2 = 0) and of each direction of SearchBySim3 (:1166-1394, max_dist = TH_HIGH): per projected point the most
 * similar keypoint of the keyframe inside the window, levels [pred-1, pred]; no state is shared between points, the
 * map surgery that follows (Replace / AddObservation / vpReplacePoint) stays with the caller.
 * aux = ur = u - bf*invz (:914) for chi2 = 1.  best_idx[i] = keypoint or -1, best_dist[i] = its distance. */
int oracle_window_best(const oracle_frame_feats *kf, const oracle_proj_points *pts, const float *sf, const float *inv_sigma2,
                       float th, int chi2, int max_dist, int32_t *best_idx_out, int32_t *best_dist_out)
{
    fgrid *g = grid_build(kf);
    int *cand = malloc(sizeof(int) * (kf->n ? kf->n : 1));
    int nfound = 0;
    for (int i = 0; i < pts->n; i++) {
        best_idx_out[i] = -1; best_dist_out[i] = 256;
        if (!pts->valid[i]) continue;
        const float u = pts->u[i], v = pts->v[i];
        if (!kf_in_image(kf, u, v)) continue;
        const int lvl = pts->level[i];
        const float radius = th * sf[lvl];
        const int nc = features_in_area(kf, g, u, v, radius, -1, -1, cand);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int kl = kf->octave[idx];
            if (kl < lvl - 1 || kl > lvl) continue;
            if (chi2) {
                const float ex = u - kf->x[idx], ey = v - kf->y[idx];
                if (kf->u_right[idx] >= 0) {              /* :967-980 */
                    const float er = pts->aux[i] - kf->u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kl] > 7.8) continue;
                } else {                                   /* :981-992 */
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kl] > 5.99) continue;
                }
            }
            const int dist = oracle_hamming(pts->desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
            if (dist < best_dist) { best_dist = dist; best_idx = idx; }
        }
        if (best_dist <= max_dist) { best_idx_out[i] = best_idx; best_dist_out[i] = best_dist; nfound++; }
    }
    grid_free(g); free(cand);
    return nfound;
}

/* ORBmatcher::SearchBySim3, src/ORBmatcher.cc:1166-1394: pts12 = KF1's map points (one per KF1 keypoint, valid = has a
 * point, not already matched, not bad, depth/range tests) projected into KF2; pts21 the reverse; match12[i1] = idx2
 * where both directions agree (:1375-1391), else -1. */
int oracle_search_by_sim3(const oracle_frame_feats *kf1, const oracle_frame_feats *kf2, const oracle_proj_points *pts12,
                          const oracle_proj_points *pts21, const float *sf1, const float *sf2, float th, int32_t *match12)
{
    int32_t *m1 = malloc(sizeof(int32_t) * (pts12->n ? pts12->n : 1)), *m2 = malloc(sizeof(int32_t) * (pts21->n ? pts21->n : 1));
    int32_t *d1 = malloc(sizeof(int32_t) * (pts12->n ? pts12->n : 1)), *d2 = malloc(sizeof(int32_t) * (pts21->n ? pts21->n : 1));
    oracle_window_best(kf2, pts12, sf2, NULL, th, 0, TH_HIGH, m1, d1);
    oracle_window_best(kf1, pts21, sf1, NULL, th, 0, TH_HIGH, m2, d2);
    int nfound = 0;
    for (int i1 = 0; i1 < pts12->n; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && idx2 < pts21->n && m2[idx2] == i1) { match12[i1] = idx2; nfound++; }
    }
    free(m1); free(m2); free(d1); free(d2);
    return nfound;
}

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:430-556 (without the vbPrevMatched update, :544-546) */
int oracle_search_for_initialization(const oracle_frame_feats *f1, const oracle_frame_feats *f2, const float *prev_xy,
                                     int window_size, float nnratio, int check_ori, int32_t *matches12)
{
    fgrid *g = grid_build(f2);
    int *cand = malloc(sizeof(int) * (f2->n ? f2->n : 1));
    int *matched_dist = malloc(sizeof(int) * (f2->n ? f2->n : 1));
    int *matches21 = malloc(sizeof(int) * (f2->n ? f2->n : 1));
    for (int i = 0; i < f2->n; i++) { matched_dist[i] = INT_MAX; matches21[i] = -1; }
    for (int i = 0; i < f1->n; i++) matches12[i] = -1;
    rot_hist rh; memset(&rh, 0, sizeof rh);
    int nmatches = 0;
    for (int i1 = 0; i1 < f1->n; i1++) {
        const int level1 = f1->octave[i1];
        if (level1 > 0) continue;
        const int nc = features_in_area(f2, g, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window_size, level1, level1, cand);
        if (nc == 0) continue;
        int best_dist = INT_MAX, best_dist2 = INT_MAX, best_idx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = oracle_hamming(f1->desc + (size_t)i1 * 32, f2->desc + (size_t)i2 * 32);
            if (matched_dist[i2] <= dist) continue;
            if (dist < best_dist) { best_dist2 = best_dist; best_dist = dist; best_idx2 = i2; }
            else if (dist < best_dist2) best_dist2 = dist;
        }
        if (best_dist <= TH_LOW) {
            if ((float)best_dist < (float)best_dist2 * nnratio) {
                if (matches21[best_idx2] >= 0) { matches12[matches21[best_idx2]] = -1; nmatches--; }
                matches12[i1] = best_idx2;
                matches21[best_idx2] = i1;
                matched_dist[best_idx2] = best_dist;
                nmatches++;
                if (check_ori) rh_push(&rh, rot_bin(f1->angle[i1], f2->angle[best_idx2]), i1);
            }
        }
    }
    if (check_ori) { /* :514-541: only entries still standing are removed */
        int cnt[HISTO_LENGTH], i1, i2, i3;
        for (int b = 0; b < HISTO_LENGTH; b++) cnt[b] = rh.n[b];
        oracle_three_maxima(cnt, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int j = 0; j < rh.n[b]; j++) {
                const int idx1 = rh.v[b][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int b = 0; b < HISTO_LENGTH; b++) free(rh.v[b]);
    grid_free(g); free(cand); free(matched_dist); free(matches21);
    return nmatches;
}

/* ------------------------------------------------------------------ grayscale ingest (f4, first half) */

/* cv::cvtColor RGB(A)/BGR(A) -> GRAY, 8U (src/Tracking.cc:177-202 call sites) */
void oracle_cvt_gray(const uint8_t *src, int w, int h, size_t sstride, int channels, int rgb_order, uint8_t *dst, size_t dstride)
{
    const int cr = 4899, cg = 9617, cb = 1868; /* R2Y, G2Y, B2Y at yuv_shift = 14 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = src + (size_t)y * sstride + (size_t)x * channels;
            const int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
            dst[(size_t)y * dstride + x] = (uint8_t)((r * cr + g * cg + b * cb + (1 << 13)) >> 14);
        }
}

/* ------------------------------------------------------------------ stereo rectification (f4, second half) */

/* cv::remap(src, dst, map1 CV_32FC1, map2 CV_32FC1, INTER_LINEAR, BORDER_CONSTANT, Scalar()) for 8UC1, the call of
 * Examples/Stereo/stereo_euroc.cc:136-137 [OpenCV generic path restated from memory: imgwarp.cpp initInterTab2D,
 * remap()'s CV_32FC1 branch, remapBilinear<FixedPtCast<int,uchar,15>, short>].  The bilinear table is built the way
 * OpenCV builds it, including saturate_cast<short> and the sum fix-up (which, with ksize 2, looks at entries k1,k2 in
 * {1,2}, i.e. at the fourth tap and into the next, still zero, table entry). */
static short g_bilinear_tab[32 * 32 + 1][4];
static int g_bilinear_ready = 0;
static void init_bilinear_tab(void)
{
    if (g_bilinear_ready) return;
    memset(g_bilinear_tab, 0, sizeof g_bilinear_tab);
    float tab1[32][2];
    for (int i = 0; i < 32; i++) { const float x = (float)i * (1.f / 32); tab1[i][0] = 1.f - x; tab1[i][1] = x; } /* interpolateLinear */
    short *flat = &g_bilinear_tab[0][0];
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            short *itab = flat + (size_t)(i * 32 + j) * 4;
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++)
                for (int k2 = 0; k2 < 2; k2++) {
                    const float v = tab1[i][k1] * tab1[j][k2];
                    const long r = lrintf(v * 32768.f);
                    itab[k1 * 2 + k2] = (short)(r > 32767 ? 32767 : r < -32768 ? -32768 : r);
                    isum += itab[k1 * 2 + k2];
                }
            if (isum != 32768) {
                const int diff = isum - 32768;
                int Mk1 = 1, Mk2 = 1, mk1 = 1, mk2 = 1;
                for (int k1 = 1; k1 < 3; k1++)
                    for (int k2 = 1; k2 < 3; k2++) {
                        if (itab[k1 * 2 + k2] < itab[mk1 * 2 + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (itab[k1 * 2 + k2] > itab[Mk1 * 2 + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) itab[Mk1 * 2 + Mk2] = (short)(itab[Mk1 * 2 + Mk2] - diff);
                else itab[mk1 * 2 + mk2] = (short)(itab[mk1 * 2 + mk2] - diff);
            }
        }
    g_bilinear_ready = 1;
}

void oracle_remap_bilinear(const uint8_t *src, int sw, int sh, size_t sstride, const float *map_x, const float *map_y,
                           uint8_t *dst, int dw, int dh, size_t dstride)
{
    init_bilinear_tab();
    const int width1 = sw - 1 > 0 ? sw - 1 : 0, height1 = sh - 1 > 0 ? sh - 1 : 0;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const int fsx = (int)lrintf(map_x[(size_t)y * dw + x] * 32), fsy = (int)lrintf(map_y[(size_t)y * dw + x] * 32);
            int sx = fsx >> 5, sy = fsy >> 5;
            sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;
            sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
            const short *w = g_bilinear_tab[(fsy & 31) * 32 + (fsx & 31)];
            int val;
            if ((unsigned)sx < (unsigned)width1 && (unsigned)sy < (unsigned)height1) {
                const uint8_t *S = src + (size_t)sy * sstride + sx;
                val = S[0] * w[0] + S[1] * w[1] + S[sstride] * w[2] + S[sstride + 1] * w[3];
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                dst[(size_t)y * dstride + x] = 0;
                continue;
            } else {
                const int x0 = (unsigned)sx < (unsigned)sw ? sx : -1, x1 = (unsigned)(sx + 1) < (unsigned)sw ? sx + 1 : -1; /* borderInterpolate, BORDER_CONSTANT */
                const int y0 = (unsigned)sy < (unsigned)sh ? sy : -1, y1 = (unsigned)(sy + 1) < (unsigned)sh ? sy + 1 : -1;
                const int v0 = x0 >= 0 && y0 >= 0 ? src[(size_t)y0 * sstride + x0] : 0, v1 = x1 >= 0 && y0 >= 0 ? src[(size_t)y0 * sstride + x1] : 0;
                const int v2 = x0 >= 0 && y1 >= 0 ? src[(size_t)y1 * sstride + x0] : 0, v3 = x1 >= 0 && y1 >= 0 ? src[(size_t)y1 * sstride + x1] : 0;
                val = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
            }
            val = (val + (1 << 14)) >> 15;
            dst[(size_t)y * dstride + x] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
        }
}

/* ------------------------------------------------------------------ Frame::UndistortKeyPoints (f2) */

/* cv::undistortPoints(src, dst, K, D, noArray(), P = K) on CV_32FC2 points, src/Frame.cc:470-515 [cvUndistortPoints of
 * OpenCV 3.2 restated from memory: double arithmetic, iters = 5; rational / thin-prism / tilt terms are zero for
 * ORB-SLAM2's 4- or 5-coefficient models and drop out exactly]. */
void oracle_undistort_points(const float *xy, int n, float fx_, float fy_, float cx_, float cy_, const float *dist, int ndist,
                             float *out)
{
    double k[14] = { 0 };
    for (int i = 0; i < ndist && i < 14; i++) k[i] = dist[i];
    const double fx = fx_, fy = fy_, ifx = 1. / fx, ify = 1. / fy, cx = cx_, cy = cy_;
    for (int i = 0; i < n; i++) {
        double x = xy[2 * i], y = xy[2 * i + 1], x0, y0;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        x0 = x; y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}
