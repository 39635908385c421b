/* Host-fed camera streams through the pipelined C ABI: S threads, one extractor handle (= one camera stream) each, every
 * thread keeps orbx_pipeline_depth() stereo frames in flight with orbx_extract_stereo_submit / orbx_extract_stereo_wait.
 * Frames live in host memory (page-locked with orbx_pinned_alloc unless --pageable); every frame is uploaded, processed
 * (extract L + extract R + ComputeStereoMatches) and its keypoints / descriptors / uRight / depth are downloaded.
 * This is the frame loop of the reference's Examples/Stereo/stereo_kitti.cc:68-117 with the per-frame work behind the ABI.
 *   gcc -O2 -std=c99 -Iinclude examples/stereo_stream.c -Lorb-slam2_amd -lorbx -lpthread -Wl,-rpath,$PWD/orb-slam2_amd -o stereo_stream
 *   ./stereo_stream [--streams S] [--frames N] [--nfeat F] [--pageable] [--sync] [--in file.bin]
 * --in: raw file "int32 w, int32 h, then K pairs of (left, right) w*h bytes"; default: a synthetic blocky texture.
 * Prints one JSON line. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "orbx.h"

static int g_w = 1241, g_h = 376, g_nfeat = 1000, g_frames = 1000, g_pairs = 4, g_pinned = 1, g_sync = 0, g_depth = 0;
static uint8_t **g_left, **g_right;
static const float BF = 386.1448f, MINZ = 386.1448f / 718.856f;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { int id, rc; long matched, kps; double lat_sum, lat_max, t0, t1; float *lat; int nlat; } worker_t;
static int cmp_float(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }
static pthread_barrier_t g_bar;

static void *worker(void *arg)
{
    worker_t *wk = (worker_t *)arg;
    orbx_extractor *ex = NULL;
    wk->rc = orbx_extractor_create(&ex, g_nfeat, 1.2f, 8, 20, 7, 0, g_w, g_h, 2);
    if (wk->rc) { fprintf(stderr, "create: %s\n", orbx_last_error()); pthread_barrier_wait(&g_bar); return NULL; }
    const int cap = orbx_max_keypoints(ex, g_w, g_h), depth = g_depth;
    orbx_keypoint *kps = (orbx_keypoint *)malloc(sizeof(orbx_keypoint) * 2 * (size_t)cap);
    uint8_t *desc = (uint8_t *)malloc((size_t)64 * cap);
    float *ur = (float *)malloc(4 * (size_t)cap), *z = (float *)malloc(4 * (size_t)cap);
    int n[2], tickets[16];
    double t_sub[16];
    /* warm-up OUTSIDE the timed region: geometry tables and workspaces (first frame), then 2 x depth frames through the pipelined form with
     * `depth` of them in flight, so that every pipeline slot, every kernel lane (the handle and its shadow handles, each with its own stream
     * and workspaces) and the pinned result blocks exist and have been touched before the clock starts -- a lane made inside the timed loop
     * was a 3-19 ms outlier of the round-3 figures */
    wk->rc = orbx_pipeline_warm(ex, g_w, g_h);     /* (the first submit would do it by itself) */
    if (!wk->rc) {
        int t;
        wk->rc = orbx_extract_stereo_submit(ex, g_left[0], g_right[0], g_w, g_h, (size_t)g_w, BF, MINZ, &t);
        if (!wk->rc) wk->rc = orbx_extract_stereo_wait(ex, t, kps, desc, cap, n, ur, z);
    }
    for (int i = 0; i < 3 * depth && !wk->rc && !g_sync; i++) {
        if (i >= depth) wk->rc = orbx_extract_stereo_wait(ex, tickets[(i - depth) % depth], kps, desc, cap, n, ur, z);
        if (i < 2 * depth && !wk->rc)
            wk->rc = orbx_extract_stereo_submit(ex, g_left[i % g_pairs], g_right[i % g_pairs], g_w, g_h, (size_t)g_w, BF, MINZ, &tickets[i % depth]);
    }
    if (wk->rc) fprintf(stderr, "warm-up: %s\n", orbx_last_error());
    wk->lat = (float *)malloc(sizeof(float) * (size_t)(g_frames > 0 ? g_frames : 1));
    wk->nlat = 0;
    pthread_barrier_wait(&g_bar);   /* every stream is warm: the timed region starts */
    wk->t0 = now();
    for (int i = 0; i < g_frames + depth && !wk->rc; i++) {
        if (g_sync) {
            if (i >= g_frames) break;
            const double t0 = now();
            wk->rc = orbx_extract_stereo(ex, g_left[i % g_pairs], g_right[i % g_pairs], g_w, g_h, (size_t)g_w, BF, MINZ, kps, desc, cap, n, ur, z);
            const double dt = now() - t0;
            wk->lat_sum += dt; if (dt > wk->lat_max) wk->lat_max = dt;
            wk->lat[wk->nlat++] = (float)(1e6 * dt);
        } else {
            if (i >= depth) {
                const int s = (i - depth) % depth;
                wk->rc = orbx_extract_stereo_wait(ex, tickets[s], kps, desc, cap, n, ur, z);
                const double dt = now() - t_sub[s];
                wk->lat_sum += dt; if (dt > wk->lat_max) wk->lat_max = dt;
                wk->lat[wk->nlat++] = (float)(1e6 * dt);
            }
            if (i < g_frames && !wk->rc) {
                const int s = i % depth;
                t_sub[s] = now();
                wk->rc = orbx_extract_stereo_submit(ex, g_left[i % g_pairs], g_right[i % g_pairs], g_w, g_h, (size_t)g_w, BF, MINZ, &tickets[s]);
            }
            if (i < depth) continue;
        }
        if (!wk->rc) { wk->kps += n[0] + n[1]; for (int k = 0; k < n[0]; k++) wk->matched += ur[k] >= 0; }
    }
    wk->t1 = now();
    if (wk->rc) fprintf(stderr, "stream %d: %s\n", wk->id, orbx_last_error());
    orbx_extractor_destroy(ex);
    free(kps); free(desc); free(ur); free(z);
    return NULL;
}

int main(int argc, char **argv)
{
    int streams = 2;
    const char *in = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--streams") && i + 1 < argc) streams = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) g_frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nfeat") && i + 1 < argc) g_nfeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) g_depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--in") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "--pageable")) g_pinned = 0;
        else if (!strcmp(argv[i], "--sync")) g_sync = 1;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
    }
    if (g_depth < 1 || g_depth > orbx_pipeline_depth()) g_depth = orbx_pipeline_depth();
    if (streams < 1 || streams > 64 || orbx_device_count() < 1) { fprintf(stderr, "no device / bad --streams\n"); return 2; }
    FILE *f = NULL;
    if (in) {
        int hdr[2];
        f = fopen(in, "rb");
        if (!f || fread(hdr, 4, 2, f) != 2) { fprintf(stderr, "cannot read %s\n", in); return 2; }
        g_w = hdr[0]; g_h = hdr[1];
    }
    const size_t npx = (size_t)g_w * g_h;
    g_left = (uint8_t **)malloc(sizeof(void *) * g_pairs); g_right = (uint8_t **)malloc(sizeof(void *) * g_pairs);
    for (int p = 0; p < g_pairs; p++) {
        g_left[p] = g_pinned ? (uint8_t *)orbx_pinned_alloc(npx) : (uint8_t *)malloc(npx);
        g_right[p] = g_pinned ? (uint8_t *)orbx_pinned_alloc(npx) : (uint8_t *)malloc(npx);
        if (!g_left[p] || !g_right[p]) { fprintf(stderr, "allocation failed\n"); return 2; }
        if (f) {
            if (fread(g_left[p], 1, npx, f) != npx || fread(g_right[p], 1, npx, f) != npx) { g_pairs = p; break; }
        } else {
            unsigned s = 12345u + 77u * p;
            for (int y = 0; y < g_h; y++)
                for (int x = 0; x < g_w; x++) {
                    s = s * 1664525u + 1013904223u;
                    g_left[p][(size_t)y * g_w + x] = (uint8_t)(((x / 24) * 37 + (y / 24) * 91 + 13 * p) % 200 + (s >> 28));
                }
            for (int y = 0; y < g_h; y++)
                for (int x = 0; x < g_w; x++) g_right[p][(size_t)y * g_w + x] = g_left[p][(size_t)y * g_w + (x + 7 < g_w ? x + 7 : g_w - 1)];
        }
    }
    if (f) fclose(f);
    if (g_pairs < 1) { fprintf(stderr, "no frames\n"); return 2; }
    pthread_t th[64];
    worker_t wk[64];
    memset(wk, 0, sizeof wk);
    pthread_barrier_init(&g_bar, NULL, (unsigned)streams);
    for (int i = 0; i < streams; i++) { wk[i].id = i; pthread_create(&th[i], NULL, worker, &wk[i]); }
    for (int i = 0; i < streams; i++) pthread_join(th[i], NULL);
    double t0 = wk[0].t0, t1 = wk[0].t1;
    for (int i = 1; i < streams; i++) { if (wk[i].t0 < t0) t0 = wk[i].t0; if (wk[i].t1 > t1) t1 = wk[i].t1; }
    const double el = t1 - t0;
    long matched = 0, kps = 0; double lat = 0, lmax = 0; int rc = 0;
    for (int i = 0; i < streams; i++) { matched += wk[i].matched; kps += wk[i].kps; lat += wk[i].lat_sum; if (wk[i].lat_max > lmax) lmax = wk[i].lat_max; rc |= wk[i].rc; }
    const long total = (long)streams * g_frames;
    /* latency distribution over every frame of every stream (submit -> results in the caller's buffers), and the first 16 frames of stream 0 */
    long nl = 0;
    for (int i = 0; i < streams; i++) nl += wk[i].nlat;
    float *all = (float *)malloc(sizeof(float) * (size_t)(nl > 0 ? nl : 1));
    nl = 0;
    for (int i = 0; i < streams; i++) { memcpy(all + nl, wk[i].lat, sizeof(float) * (size_t)wk[i].nlat); nl += wk[i].nlat; }
    char first[512] = "";
    for (int i = 0, o = 0; i < 16 && i < wk[0].nlat && o < 480; i++) o += snprintf(first + o, sizeof first - (size_t)o, "%s%.0f", i ? ", " : "", wk[0].lat[i]);
    qsort(all, (size_t)nl, sizeof(float), cmp_float);
    const double p50 = nl ? all[nl / 2] : 0, p99 = nl ? all[(long)(0.99 * (nl - 1))] : 0, p999 = nl ? all[(long)(0.999 * (nl - 1))] : 0;
    printf("{\"latency_us_p50\": %.1f, \"latency_us_p99\": %.1f, \"latency_us_p999\": %.1f, \"first_frames_us\": [%s], ", p50, p99, p999, first);
    printf("\"frames_per_s\": %.1f, \"streams\": %d, \"frames_in_flight_per_stream\": %d, \"pinned_frame_buffers\": %s, \"frames\": %ld, "
           "\"timed_s\": %.3f, \"latency_us_mean\": %.1f, \"latency_us_max\": %.1f, \"keypoints_per_image\": %.1f, "
           "\"stereo_matches_per_frame\": %.1f, \"h2d_bytes_per_frame\": %zu, \"form\": \"%s\", \"w\": %d, \"h\": %d, \"nfeatures\": %d}\n",
           total / el, streams, g_sync ? 1 : g_depth, g_pinned ? "true" : "false", total, el, 1e6 * lat / total, 1e6 * lmax,
           kps / (2.0 * total), (double)matched / total, 2 * npx, g_sync ? "orbx_extract_stereo" : "orbx_extract_stereo_submit/_wait", g_w, g_h, g_nfeat);
    return rc ? 3 : 0;
}
