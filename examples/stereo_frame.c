/* Minimal C client of the orbx C ABI: one stereo frame through host pointers (what an ORB-SLAM2 adaptor does per frame).
 *   gcc -std=c99 -Iinclude examples/stereo_frame.c -Lorb-slam2_amd -lorbx -Wl,-rpath,$PWD/orb-slam2_amd -o stereo_frame
 * Synthetic input: a textured image and the same image shifted by 7 pixels as the right eye. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orbx.h"

int main(void)
{
    const int w = 752, h = 480, disparity = 7;
    uint8_t *left = (uint8_t *)malloc((size_t)w * h), *right = (uint8_t *)malloc((size_t)w * h);
    unsigned s = 12345u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            s = s * 1664525u + 1013904223u;
            const int blk = ((x / 24) * 37 + (y / 24) * 91) % 200;          /* blocky texture + a little noise */
            left[y * w + x] = (uint8_t)(blk + (s >> 28));
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) right[y * w + x] = left[y * w + (x + disparity < w ? x + disparity : w - 1)];

    orbx_extractor *ex = NULL;
    int rc = orbx_extractor_create(&ex, 1000, 1.2f, 8, 20, 7, /*device*/ 0, w, h, /*max_batch*/ 2);
    if (rc != ORBX_OK) { fprintf(stderr, "orbx_extractor_create: %s\n", orbx_last_error()); return 2; }
    const int cap = orbx_max_keypoints(ex, w, h);
    orbx_keypoint *kps = (orbx_keypoint *)malloc(sizeof(orbx_keypoint) * 2 * (size_t)cap);
    uint8_t *desc = (uint8_t *)malloc((size_t)64 * cap);
    float *u_right = (float *)malloc(sizeof(float) * (size_t)cap), *depth = (float *)malloc(sizeof(float) * (size_t)cap);
    int n[2] = { 0, 0 };
    rc = orbx_extract_stereo(ex, left, right, w, h, (size_t)w, /*bf*/ 386.1448f, /*mb*/ 0.5372f, kps, desc, cap, n, u_right, depth);
    if (rc != ORBX_OK) { fprintf(stderr, "orbx_extract_stereo: %s\n", orbx_last_error()); return 3; }
    int matched = 0, close_to_truth = 0;
    for (int i = 0; i < n[0]; i++)
        if (u_right[i] >= 0) {
            matched++;
            const float d = kps[i].x - u_right[i];
            if (d > disparity - 1.5f && d < disparity + 1.5f) close_to_truth++;
        }
    printf("left %d keypoints, right %d, stereo matches %d (%d within 1.5 px of the true disparity %d)\n", n[0], n[1], matched,
           close_to_truth, disparity);
    orbx_extractor_destroy(ex);
    free(left); free(right); free(kps); free(desc); free(u_right); free(depth);
    return (n[0] > 100 && matched > 20 && close_to_truth * 2 > matched) ? 0 : 1;
}
