/*
 * orbx.h — C ABI of the MI355X-native ORB front end + Hamming matchers (liborbx.so).
 *
 * Drop-in boundary for the ORB-SLAM2 hot path.  The reference has no
 * FFI; the seam is two C++ classes and one Frame method, so each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the C++ adaptor a maintainer adds on the reference side (ORBextractor/ORBmatcher shims).
 *
 * Conventions: plain C, no exceptions; every function returns ORBX_OK (0) or a negative
 * ORBX_E_* code and records a message readable through orbx_last_error() (thread-local).
 * All buffers are caller-allocated with explicit capacities.  "host" entry points take host
 * pointers and synchronise; "*_device" entry points take device (HBM) pointers, enqueue on a
 * HIP stream and return without synchronising (the throughput path).
 * There is no CPU fallback: without a usable gfx950 device every compute call fails with
 * ORBX_E_NO_DEVICE.
 *
 * Threading (as the reference): one extractor handle is used by one thread
 * at a time; distinct handles may be used concurrently.
 *
 * Streams: a handle remembers the stream of its most recent "*_device" launches.  Whenever a call
 * has to rebuild the per-image-size tables (first call / a new image size) or to grow a workspace,
 * it first waits for that stream and for the handle's own stream, so tables and buffers are never
 * rewritten or freed under running kernels.  A call that launches on a DIFFERENT stream than the
 * previous one on the same handle is ordered behind the earlier work by an event (the handle's
 * pyramid and candidate workspaces are shared by all its launches), without a host synchronisation.
 * A caller stream must therefore stay valid until the handle has been used with another stream or
 * destroyed.  Steady state (same size, same batch, same stream) neither synchronises nor records events.  A kernel-side error (ORBX_E_CAPACITY from orbx_sync) belongs to the work
 * that orbx_sync has just waited for; the flag is cleared when it is reported.
 */
#ifndef ORBX_H
#define ORBX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_E_INVALID (-1)    /* bad argument */
#define ORBX_E_CAPACITY (-2)   /* an output capacity is too small */
#define ORBX_E_TOO_SMALL (-3)  /* image smaller than one 30-px FAST cell at some level (reference: div by 0) */
#define ORBX_E_NO_DEVICE (-4)  /* no gfx950 device / device index out of range */
#define ORBX_E_HIP (-5)        /* a HIP runtime call failed */

/* == cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id.
 * Output element type of ORBextractor::operator() (include/ORBextractor.h:74-76). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbx_keypoint;

typedef struct orbx_extractor orbx_extractor; /* opaque; owns device pyramid + workspace */

const char *orbx_last_error(void);
int orbx_device_count(void);
/* "<pci bus id> <uuid hex> <gcn arch>" of one visible device: an N-process launch (one process per GPU) reports it
 * per rank so that the line shows every rank held its own card.  No reference counterpart (the reference is single-process CPU). */
int orbx_device_identity(int device, char *buf, int cap);

/* ---- ORBextractor (include/ORBextractor.h:58-139, src/ORBextractor.cc:429-534) ---------- */

/* replaces `new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)`
 * (src/Tracking.cc:124-130).  device = HIP device index; max_w/max_h/max_batch size the
 * workspace (images of any size <= max and batches <= max_batch are accepted later). */
int orbx_extractor_create(orbx_extractor **out, int nfeatures, float scale_factor, int nlevels,
                          int ini_th_fast, int min_th_fast, int device, int max_w, int max_h, int max_batch);
void orbx_extractor_destroy(orbx_extractor *e);

/* Which 7-tap table cv::GaussianBlur(7x7, sigma 2) (src/ORBextractor.cc:1311) is computed with.  OpenCV's 8-bit Gaussian has had two
 * fixed-point kernels over the releases the reference builds against ("OpenCV 2.4.3 or later, tested with 2.4.11 and 3.2", README.md:68):
 *   ORBX_CV_TAPS_257: every tap rounded by itself, cvRound(k * 256): 18 34 49 55 49 34 18 (sum 257)  -- default;
 *   ORBX_CV_TAPS_256: rounding error diffused so that the taps sum to exactly 256: 18 34 48 56 48 34 18.
 * Different taps give different blurred pixels and so different rBRIEF bits; the arithmetic around the taps (exact integer row pass,
 * (sum + 2^15) >> 16 column pass) is the same for both.  WHICH OpenCV RELEASE USES WHICH TABLE IS PARITY UNPINNED: the reference holds no
 * fixture and OpenCV is not installed here.  From memory: 3.2 (the tested version) has the 257 table; the fixed-point path that came with
 * 3.4.2 may still round tap by tap, the error-diffused table arrived with a later 3.4.x / 4.x release.  A deployment decides by comparing one
 * blurred image against its own OpenCV.  The *_PROFILE_* names are the older spelling of the same two values.
 * Call right after orbx_extractor_create; it applies to all later extractions. */
#define ORBX_CV_TAPS_257 0
#define ORBX_CV_TAPS_256 1
#define ORBX_CV_PROFILE_3_2 ORBX_CV_TAPS_257
#define ORBX_CV_PROFILE_3_4_2 ORBX_CV_TAPS_256
int orbx_extractor_set_cv_profile(orbx_extractor *e, int profile);
int orbx_gaussian_taps(int profile, int taps[7]);

/* Tuning, no effect on results: extractions of at most `max_images` images per launch build the pyramid (ComputePyramid,
 * src/ORBextractor.cc:1347-1370) with two launches that each produce several levels (a single frame's time is its chain of dependent
 * launches); launches of up to 24 images still take the small levels (3 and up) in one launch; larger batches keep one launch per
 * level, which moves fewer bytes.  Default 8; 0 = always one launch per level. */
int orbx_extractor_set_pyramid_group_limit(orbx_extractor *e, int max_images);

/* getters: GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors /
 * GetScaleSigmaSquares / GetInverseScaleSigmaSquares (include/ORBextractor.h:78-98).
 * Each output array has nlevels entries; NULL pointers are skipped. */
int orbx_get_levels(const orbx_extractor *e);
float orbx_get_scale_factor(const orbx_extractor *e);
int orbx_get_scale_tables(const orbx_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2);
/* per-level feature quotas mnFeaturesPerLevel (src/ORBextractor.cc:468-493) */
int orbx_get_features_per_level(const orbx_extractor *e, int *quota);
/* smallest per-image keypoint capacity that can never overflow (quadtree may return up to
 * max(quota+2, 4*nIni) per level) */
int orbx_max_keypoints(const orbx_extractor *e, int w, int h);

/* replaces ORBextractor::operator()(image, mask, keypoints, descriptors)
 * (src/ORBextractor.cc:1261-1339; caller src/Frame.cc:285-292).  img: 8-bit grey, row stride in
 * bytes.  kps[cap], desc[cap*32].  An empty image (w==0||h==0) returns ORBX_OK with *n_out = 0
 * (reference :1264-1265 returns silently). */
int orbx_extract(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride,
                 orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out);

/* Same for a colour frame: the cvtColor(CV_RGB2GRAY | CV_BGR2GRAY | CV_RGBA2GRAY | CV_BGRA2GRAY) that
 * Tracking::GrabImageStereo/RGBD/Monocular apply first (src/Tracking.cc:177-202,:217-231,:254-268) runs on device into the
 * level-0 staging (first half).  channels 3 or 4; rgb_order 1 = R first (mbRGB), 0 = B first.
 * gray_out (optional, may be NULL; stride gray_stride) receives the grey image, i.e. what mImGray holds afterwards. */
int orbx_extract_color(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride, int channels, int rgb_order,
                       orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, uint8_t *gray_out, size_t gray_stride);

/* ---- stereo rectification (second half) ----------------------------------
 * cv::remap(imLeft, imLeftRect, M1l, M2l, cv::INTER_LINEAR) of Examples/Stereo/stereo_euroc.cc:136-137 (8-bit grey,
 * BORDER_CONSTANT 0).  map_x / map_y = the CV_32FC1 pair cv::initUndistortRectifyMap returns (:103-104), dst_h x dst_w
 * floats each, dense; converted once to OpenCV's fixed-point form on the device. */
typedef struct orbx_rectifier orbx_rectifier;
int orbx_rectifier_create(int device, int src_w, int src_h, int dst_w, int dst_h, const float *map_x, const float *map_y,
                          orbx_rectifier **out);
void orbx_rectifier_destroy(orbx_rectifier *r);
int orbx_rectifier_size(const orbx_rectifier *r, int *dst_w, int *dst_h);
/* batch of device images -> rectified device images (d_dst, dst_pitch and dst_stride multiples of 4); asynchronous on `stream` */
int orbx_remap_batch_device(const orbx_rectifier *r, const void *d_src, size_t src_stride, size_t src_pitch, int batch,
                            void *d_dst, size_t dst_stride, size_t dst_pitch, void *stream);
/* raw grey frame (host) -> rectify on device -> extract; rect_out (optional) receives the rectified image */
int orbx_extract_rectified(orbx_extractor *e, const orbx_rectifier *r, const uint8_t *img, int w, int h, size_t stride,
                           orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, uint8_t *rect_out, size_t rect_stride);

/* One stereo frame in one call (host pointers): both ORBextractor::operator() calls of Frame::Frame(imLeft, imRight, ...)
 * (src/Frame.cc:82-85, two threads in the reference) and Frame::ComputeStereoMatches (:97, :577-751).  e must have been
 * created with max_batch >= 2.  kps[2*cap] / desc[2*cap*32]: left eye at index 0, right eye at index cap; n_out[2];
 * u_right[cap], depth[cap] = mvuRight, mvDepth of the left keypoints.  bf = mbf, min_z = mb (see orbx_stereo_match). */
int orbx_extract_stereo(orbx_extractor *e, const uint8_t *img_left, const uint8_t *img_right, int w, int h, size_t stride,
                        float bf, float min_z, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                        float *u_right, float *depth);

/* Pipelined form of orbx_extract_stereo for a camera stream fed from host memory: _submit enqueues one stereo frame
 * (upload on a copy stream, both extractions + ComputeStereoMatches on the compute stream, download on a second copy
 * stream) and returns a ticket without waiting; _wait blocks until that frame's results are on the host and copies them
 * out (same output contract as orbx_extract_stereo).  Up to orbx_pipeline_depth() frames may be in flight per handle:
 * frame i+1 uploads and frame i-1 downloads while frame i computes.  Tickets must be waited for in submission order;
 * a submit with all slots in flight fails with ORBX_E_INVALID.  If img_left / img_right lie in memory obtained from
 * orbx_pinned_alloc and stride == w, the upload reads them in place (they must stay untouched until _wait returns);
 * otherwise they are copied to the handle's pinned staging inside _submit and may be reused at once.
 * Replaces the per-frame sequence of Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:82-97) in the frame loop of
 * Examples/Stereo/stereo_kitti.cc:68-117. */
int orbx_pipeline_depth(void);
/* Makes and touches every pipeline slot and kernel lane of the pipelined forms for images of w x h by running one scratch frame through each
 * (tens of milliseconds, once).  orbx_extract_stereo_submit does this by itself on the first frame of a size; a camera loop that wants that
 * first frame on time as well (Examples/Stereo/stereo_kitti.cc:83-117 times every frame) calls it before the loop. */
int orbx_pipeline_warm(orbx_extractor *e, int w, int h);
int orbx_extract_stereo_submit(orbx_extractor *e, const uint8_t *img_left, const uint8_t *img_right, int w, int h, size_t stride,
                               float bf, float min_z, int *ticket);
int orbx_extract_stereo_wait(orbx_extractor *e, int ticket, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                             float *u_right, float *depth);
/* the same pipeline for a monocular stream (Frame::Frame(imGray, ...), src/Frame.cc:176-215; frame loop of
 * Examples/Monocular/mono_tum.cc): one extraction per ticket, same output contract as orbx_extract.  Mono and stereo
 * tickets share the handle's slots and ordering. */
int orbx_extract_submit(orbx_extractor *e, const uint8_t *img, int w, int h, size_t stride, int *ticket);
int orbx_extract_wait(orbx_extractor *e, int ticket, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out);
/* page-locked host memory for frame buffers (uploads from it need no staging copy) */
void *orbx_pinned_alloc(size_t bytes);
void orbx_pinned_free(void *p);

/* B images of one size in one pass (host pointers). kps[B*cap], desc[B*cap*32], n_out[B]. */
int orbx_extract_batch(orbx_extractor *e, const uint8_t *const *imgs, int batch, int w, int h, size_t stride,
                       orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out);

/* Throughput path: images already resident in HBM.  Image b starts at d_imgs + b*img_stride and
 * has row pitch `pitch` bytes; outputs are device buffers d_kps[B*cap], d_desc[B*cap*32],
 * d_n_out[B].  Enqueued on `stream` (a hipStream_t; NULL = the handle's own stream); returns
 * before completion.  The input must stay valid until orbx_stereo_match*_device /
 * orbx_pyramid_level calls that read level 0 have completed (level 0 is read in place). */
int orbx_extract_batch_device(orbx_extractor *e, const void *d_imgs, size_t img_stride, size_t pitch,
                              int batch, int w, int h, void *d_kps, void *d_desc, int cap, void *d_n_out,
                              void *stream);
/* wait for everything enqueued on the handle's stream (or `stream`) */
int orbx_sync(orbx_extractor *e, void *stream);

/* backs the public member mvImagePyramid (include/ORBextractor.h:100): copy level `level` of
 * image `image_index` of the most recent extract call to host memory (no 19-px border: the
 * border is never read by the hot path). */
int orbx_pyramid_level(orbx_extractor *e, int image_index, int level, uint8_t *dst, size_t dst_stride, int *w, int *h);

/* ---- Frame::ComputeStereoMatches (src/Frame.cc:577-751) ---------------------------------- */

/* L and R have each run extract on the left/right image (their pyramids are read).
 * min_z replaces the `mb` member the reference reads uninitialised; maxD = bf/min_z.
 * u_right[nL], depth[nL]: -1 where unmatched (mvuRight / mvDepth). */
int orbx_stereo_match(orbx_extractor *L, orbx_extractor *R,
                      const orbx_keypoint *kL, const uint8_t *dL, int nL,
                      const orbx_keypoint *kR, const uint8_t *dR, int nR,
                      float bf, float min_z, float *u_right, float *depth);

/* Batched, device-resident: pair p uses image (imgL0+p) of handle L and (imgR0+p) of handle R
 * from their most recent extract_batch calls (L and R may be the same handle).  Keypoints /
 * descriptors / counts are the device outputs of orbx_extract_batch_device with capacity `cap`.
 * d_u_right, d_depth: [batch*cap] floats.
 * `row_table` states where the row table of src/Frame.cc:584-604 comes from -- never guessed from addresses:
 *   ORBX_ROWTAB_FROM_KEYPOINTS  built from the right keypoints handed in (d_kR, d_nR), whatever they are: always correct;
 *   ORBX_ROWTAB_OF_EXTRACTION   the by-product table R's most recent orbx_extract_batch_device left behind (one launch less).
 *                               The caller thereby asserts that d_kR still holds exactly what that extraction wrote; the call is
 *                               refused with ORBX_E_INVALID when d_kR / cap / the image range are not the ones of that extraction,
 *                               or when that extraction built no table (images taller than 600 rows). */
#define ORBX_ROWTAB_FROM_KEYPOINTS 0
#define ORBX_ROWTAB_OF_EXTRACTION 1
int orbx_stereo_match_batch_device(orbx_extractor *L, int imgL0, orbx_extractor *R, int imgR0, int batch,
                                   const void *d_kL, const void *d_dL, const void *d_nL,
                                   const void *d_kR, const void *d_dR, const void *d_nR, int cap,
                                   float bf, float min_z, void *d_u_right, void *d_depth, int row_table, void *stream);
/* 1 when R's most recent extraction left a row table behind that ORBX_ROWTAB_OF_EXTRACTION could use for (d_kR, imgR0, batch, cap) */
int orbx_stereo_row_table_available(const orbx_extractor *R, const void *d_kR, int imgR0, int batch, int cap);

/* ---- ORBmatcher (include/ORBmatcher.h:41-83) ---------------------------------------------- */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1733-1749), host inline popcount */
int orbx_hamming(const uint8_t *a, const uint8_t *b);

/* One side of a BoW search: descriptors + DBoW2::FeatureVector flattened to CSR
 * (Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22: node ids ascending, feature indices ascending
 * within a node, every feature in at most one node) + per-feature attributes.  Host pointers. */
typedef struct {
    int n;                   /* number of features */
    const uint8_t *desc;     /* [n][32] */
    int nnodes;
    const uint32_t *node_id; /* [nnodes] */
    const int32_t *node_off; /* [nnodes+1] */
    const uint32_t *feat;    /* [node_off[nnodes]] */
    const uint8_t *flag;     /* [n] per-search meaning, see below */
    const float *angle;      /* [n] keypoint angle, degrees */
    const float *x, *y;      /* [n] undistorted position (triangulation only, else may be NULL) */
    const int32_t *octave;   /* [n] (triangulation only) */
    const float *u_right;    /* [n] (triangulation only): <0 = monocular feature */
} orbx_featset;

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:171-303).
 * kf->flag[i] != 0 <=> KF feature i holds a non-bad MapPoint; f->flag unused.
 * match_f[f->n] = matched KF feature index or -1; *nmatches = return value of the reference. */
int orbx_search_by_bow_kf_f(int device, const orbx_featset *kf, const orbx_featset *f,
                            float nnratio, int check_orientation, int32_t *match_f, int *nmatches);
/* many keyframes against one frame in one launch (Tracking::Relocalization loop,
 * src/Tracking.cc:1661-1682): match_f[nkf][f->n], nmatches[nkf] */
int orbx_search_by_bow_kf_f_batch(int device, const orbx_featset *kfs, int nkf, const orbx_featset *f,
                                  float nnratio, int check_orientation, int32_t *match_f, int *nmatches);

/* Device-resident keyframe set for the relocalisation / place-recognition loops
 * (src/Tracking.cc:1661-1682 calls SearchByBoW once per candidate keyframe): the keyframes'
 * descriptors, CSR feature vectors, flags and angles are uploaded once; each search uploads only
 * the frame side.  match_f[nkf][f->n], nmatches[nkf] as in orbx_search_by_bow_kf_f_batch. */
typedef struct orbx_bowdb orbx_bowdb;
int orbx_bowdb_create(int device, const orbx_featset *kfs, int nkf, orbx_bowdb **out);
int orbx_bowdb_search(orbx_bowdb *db, const orbx_featset *f, float nnratio, int check_orientation,
                      int32_t *match_f, int *nmatches);
int orbx_bowdb_size(const orbx_bowdb *db);
void orbx_bowdb_destroy(orbx_bowdb *db);

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (src/ORBmatcher.cc:568-702).
 * flag != 0 <=> non-bad MapPoint (both sides).  match12[k1->n] = KF2 feature index or -1. */
int orbx_search_by_bow_kf_kf(int device, const orbx_featset *k1, const orbx_featset *k2,
                             float nnratio, int check_orientation, int32_t *match12, int *nmatches);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:704-871) incl. CheckDistEpipolarLine
 * (:147-164).  flag != 0 <=> the feature already has a MapPoint (skipped).  F12 row-major 3x3;
 * (ex,ey) = epipole of camera 1 in image 2 (:712-718, computed by the adaptor);
 * scale_factors2 / level_sigma2_2: KF2's per-level tables [nlevels2].
 * pairs[2*cap] receives (idx1, idx2) sorted by idx1; *npairs = number found (may exceed cap:
 * then ORBX_E_CAPACITY). */
int orbx_search_for_triangulation(int device, const orbx_featset *k1, const orbx_featset *k2,
                                  const float F12[9], float ex, float ey,
                                  const float *scale_factors2, const float *level_sigma2_2, int nlevels2,
                                  int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs);

/* The loops the reference runs these searches in, as ONE call (one launch, one PCIe round trip):
 * LoopClosing::ComputeSim3 calls SearchByBoW(mpCurrentKF, pKF, ...) per loop candidate (src/LoopClosing.cc:293-323):
 * k1 against k2s[0..n2): match12[n2][k1->n], nmatches[n2]. */
int orbx_search_by_bow_kf_kf_batch(int device, const orbx_featset *k1, const orbx_featset *k2s, int n2,
                                   float nnratio, int check_orientation, int32_t *match12, int *nmatches);
/* LocalMapping::CreateNewMapPoints calls SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, ...) per neighbour keyframe
 * (src/LocalMapping.cc:241-309): k1 against k2s[0..n2), each with its own F12s[i][9] and epipole epipoles[i][2]; one set of
 * per-level tables (all keyframes of a map share one extractor).  pairs[n2][2*cap], npairs[n2]; ORBX_E_CAPACITY if any
 * list is longer than cap (the first cap entries of each are written). */
int orbx_search_for_triangulation_batch(int device, const orbx_featset *k1, const orbx_featset *k2s, int n2,
                                        const float *F12s, const float *epipoles,
                                        const float *scale_factors2, const float *level_sigma2_2, int nlevels2,
                                        int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs);

/* ---- resident keyframes --------------------------------------------------------------------------
 * A keyframe's descriptors (KeyFrame::mDescriptors), FeatureVector (mFeatVec) and keypoint attributes (mvKeysUn, mvuRight)
 * never change once the keyframe exists (src/KeyFrame.cc:29-60); what changes is which features hold map points.  orbx_kf
 * keeps the immutable part in HBM in FeatureVector order (made once per keyframe, e.g. from the KeyFrame constructor; a
 * Frame can be uploaded the same way for the per-frame searches); the searches below then take the map-point flags per
 * call and move only those, the node intersection and the results over PCIe.  fs->flag is ignored by orbx_kf_create;
 * x / y / octave / u_right may be NULL for a set that is only used in SearchByBoW.
 * flag arrays are per FEATURE ([n] of that keyframe), with the meaning of the host-pointer call they mirror. */
typedef struct orbx_kf orbx_kf;
int orbx_kf_create(int device, const orbx_featset *fs, orbx_kf **out);
void orbx_kf_destroy(orbx_kf *k);
int orbx_kf_size(const orbx_kf *k);
/* The per-call matchers keep a small per-THREAD context per device (a stream, mapped pinned blobs, device scratch).  It is given back
 * when the thread ends; a long-lived thread that is done matching (or a pool worker between jobs) may give it back now. */
void orbx_thread_release(void);
/* SearchByBoW(pKF, F): kf_flag as orbx_search_by_bow_kf_f's kf->flag; match_f[f's n] */
int orbx_kf_search_by_bow_kf_f(const orbx_kf *kf, const uint8_t *kf_flag, const orbx_kf *f,
                               float nnratio, int check_orientation, int32_t *match_f, int *nmatches);
/* Tracking::Relocalization's loop (src/Tracking.cc:1661-1682): SearchByBoW(pKF, mCurrentFrame) for nkf candidate RESIDENT keyframes against one
 * frame as ONE call; the frame lives one frame time and is given as host pointers (f->flag is not read): match_f[nkf][f->n], nmatches[nkf] */
int orbx_kf_search_by_bow_kfs_f(const orbx_kf *const *kfs, const uint8_t *const *kf_flags, int nkf, const orbx_featset *f,
                                float nnratio, int check_orientation, int32_t *match_f, int *nmatches);
/* SearchByBoW(pKF1, pKF2) for n2 candidates: match12[n2][k1's n], nmatches[n2] */
int orbx_kf_search_by_bow_kf_kf(const orbx_kf *k1, const uint8_t *flag1, const orbx_kf *const *k2s, const uint8_t *const *flags2, int n2,
                                float nnratio, int check_orientation, int32_t *match12, int *nmatches);
/* SearchForTriangulation for n2 neighbours; flag1 / flags2 (or flags2[i]) may be NULL = no feature has a map point */
int orbx_kf_search_for_triangulation(const orbx_kf *k1, const uint8_t *flag1, const orbx_kf *const *k2s, const uint8_t *const *flags2, int n2,
                                     const float *F12s, const float *epipoles,
                                     const float *scale_factors2, const float *level_sigma2_2, int nlevels2,
                                     int only_stereo, int check_orientation, int32_t *pairs, int cap, int *npairs);

/* ---- Frame::ComputeBoW (src/Frame.cc:459-466) --------------------------- */

/* DBoW2 vocabulary tree resident in HBM.  Arrays describe nodes 1..N in id order exactly as
 * TemplatedVocabulary::loadFromTextFile builds them (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
 * 1358-1445): parent id (0 = root), leaf flag, 32-byte descriptor, weight; children are attached
 * to their parent in id order and word ids are given to leaves in id order.  k <= 20, L <= 10 as
 * in the reference loader.  Only TF_IDF weighting with L1_NORM scoring (ORBvoc.txt) is supported. */
typedef struct orbx_vocab orbx_vocab;
int orbx_vocab_create(int device, int k, int L, int nnodes_minus_root, const int32_t *parent, const uint8_t *is_leaf,
                      const uint8_t *desc, const double *weight, orbx_vocab **out);
/* reads the ORBvoc.txt text format: header "k L scoring weighting", then one line per node
 * "parent isLeaf d0 .. d31 weight" */
int orbx_vocab_load_text(int device, const char *path, orbx_vocab **out);
int orbx_vocab_info(const orbx_vocab *v, int *k, int *L, int *nnodes, int *nwords);
void orbx_vocab_destroy(orbx_vocab *v);

/* TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) (:1127-1194, :1218-1259).
 * desc[n][32] (n <= 8192).  Per-feature outputs (may be NULL): word id, word weight (0 = stopped word:
 * the feature is in neither vector), node id at tree level L - levelsup.  BowVector: bow_id/bow_val[*nbow],
 * ascending word ids, values L1-normalised, bit-exact doubles (weights added in feature order, norm in
 * word order as std::map iteration gives).  FeatureVector as CSR ready for orbx_featset:
 * fv_node_id[*fv_nnodes], fv_node_off[*fv_nnodes + 1], fv_feat[].  Capacities: n (n + 1 for fv_node_off). */
int orbx_bow_transform(orbx_vocab *v, const uint8_t *desc, int n, int levelsup,
                       uint32_t *word_id, double *word_weight, uint32_t *node_id,
                       uint32_t *bow_id, double *bow_val, int *nbow,
                       uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes);

/* ---- projection-guided tracking matchers -------------------------------- */

/* the current Frame: undistorted keypoints, right coordinates, descriptors, image bounds (the 64x48 feature
 * grid of Frame::AssignFeaturesToGrid / GetFeaturesInArea, src/Frame.cc:261-279,:386-457, is rebuilt on device) */
/* ---- batched, device-resident ComputeBoW + relocalisation search (BASELINE config 3 at throughput) ----
 * orbx_bow_frames holds, in HBM, what Frame::ComputeBoW produces (mBowVec, mFeatVec) for a batch of frames whose
 * keypoints / descriptors / counts are the device outputs of orbx_extract_batch_device (same cap).  Nothing crosses
 * PCIe between extraction, transform and search; orbx_bow_frames_read copies one frame's vectors to the host.
 * stream = NULL: the transform runs on the vocabulary's own stream, read / search on the stream of the last transform. */
typedef struct orbx_bow_frames orbx_bow_frames;
int orbx_bow_frames_create(int device, int max_batch, int cap, orbx_bow_frames **out);
void orbx_bow_frames_destroy(orbx_bow_frames *f);
int orbx_bow_transform_batch_device(orbx_vocab *v, orbx_bow_frames *f, const void *d_kps, const void *d_desc, const void *d_n,
                                    int batch, int levelsup, void *stream);
int orbx_bow_frames_read(orbx_bow_frames *f, int index, void *stream, uint32_t *bow_id, double *bow_val, int *nbow,
                         uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes);
/* every keyframe of db against frames 0..batch-1 of f (SearchByBoW(KF, F), src/ORBmatcher.cc:171-303, in the loop of
 * Tracking::Relocalization :1661-1682): d_match[batch][nkf][cap] (int32, KF feature per frame feature or -1) and
 * d_nmatches[batch][nkf], both device memory; asynchronous on `stream`. */
int orbx_bowdb_search_batch_device(orbx_bowdb *db, const orbx_bow_frames *f, int batch, float nnratio, int check_orientation,
                                   void *d_match, void *d_nmatches, void *stream);
/* The same search with compact results: d_pairs[batch][nkf][cap_pairs][2] int32 receives, per (frame, keyframe), the first
 * min(count, cap_pairs) matches as (frame feature, keyframe feature) in frame-feature order -- what Tracking::Relocalization hands to its PnP
 * solver next (src/Tracking.cc:1682-1693) -- and d_nmatches[batch][nkf] the counts (a count above cap_pairs says the list was cut). */
int orbx_bowdb_search_batch_device_compact(orbx_bowdb *db, const orbx_bow_frames *f, int batch, float nnratio, int check_orientation,
                                           void *d_pairs, int cap_pairs, void *d_nmatches, void *stream);

typedef struct {
    int n;
    const float *x, *y;        /* mvKeysUn[i].pt */
    const int32_t *octave;     /* mvKeysUn[i].octave */
    const float *angle;        /* mvKeysUn[i].angle */
    const float *u_right;      /* mvuRight */
    const uint8_t *desc;       /* mDescriptors [n][32] */
    const uint8_t *occupied;   /* 1: mvpMapPoints[i] is set and has Observations() > 0 before the call */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
} orbx_frame_feats;
/* the projected map points in the reference's iteration order; the adaptor (which has the poses) projects */
typedef struct {
    int n;
    const float *u, *v;        /* last-frame search: u, v of src/ORBmatcher.cc:1428-1429 ; map-point search: mTrackProjX/Y */
    const float *aux;          /* last-frame search: invzc (:1425)                   ; map-point search: mTrackProjXR */
    const int32_t *level;      /* last-frame search: LastFrame.mvKeys[i].octave      ; map-point search: mnTrackScaleLevel */
    const float *angle;        /* last-frame search only: LastFrame.mvKeysUn[i].angle */
    const float *view_cos;     /* map-point search only: mTrackViewCos */
    const uint8_t *desc;       /* pMP->GetDescriptor() [n][32] */
    const uint8_t *valid;      /* last-frame: pMP && !mvbOutlier[i] ; map-point: mbTrackInView && !isBad() */
    const uint8_t *has_obs;    /* pMP->Observations() > 0: a match by this point blocks the feature for later points */
} orbx_proj_points;

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1396-1553;
 * Tracking::TrackWithMotionModel).  direction: 0 none, 1 bForward, 2 bBackward (:1412-1413).  match_cur[cur->n] =
 * index of the point each current feature finally holds (-1 none); *nmatches = the reference's return value
 * (it counts every accepted point, also one whose feature a later point overwrote).
 * check_orientation: 0 off, 1 on, 3 on + a feature whose match the rotation filter cleared reads -2 instead of -1: the
 * reference sets CurrentFrame.mvpMapPoints[...] = NULL there (:1526-1545), which differs from "never matched" when the
 * feature held a point before the call; the same two bits in orbx_search_by_projection_keyframe (:1660-1680). */
int orbx_search_by_projection_last_frame(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                         const float *scale_factors, int nlevels, float th, int direction, float mbf,
                                         int check_orientation, int32_t *match_cur, int *nmatches);
/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:48-129; Tracking::SearchLocalPoints) */
int orbx_search_by_projection_map_points(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                         const float *scale_factors, int nlevels, float th, float nnratio,
                                         int32_t *match_cur, int *nmatches);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1555-1685; Tracking::Relocalization).  pts = pKF->GetMapPointMatches() in keypoint order, projected
 * by the adaptor with CurrentFrame.mTcw: u, v (:1588-1589), level = PredictScale (:1607), angle = pKF->mvKeysUn[i].angle,
 * valid = pMP && !isBad() && !sAlreadyFound.count(pMP) && distance range (:1577-1605); aux / view_cos / has_obs unused.
 * cur->occupied = CurrentFrame.mvpMapPoints[i] != NULL; every accepted match blocks its feature (:1624-1625). */
int orbx_search_by_projection_keyframe(int device, const orbx_frame_feats *cur, const orbx_proj_points *pts,
                                       const float *scale_factors, int nlevels, float th, int orb_dist,
                                       int check_orientation, int32_t *match_cur, int *nmatches);
/* ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, th)
 * (src/ORBmatcher.cc:305-415; LoopClosing::ComputeSim3).  pts = vpPoints projected with Scw (:336-352), level = PredictScale
 * (:374), valid = !isBad() && !spAlreadyFound.count(pMP) && depth > 0 && distance range && viewing angle (:331-372); the
 * IsInImage test (:355) is done here from kf's bounds.  kf->occupied = vpMatched[idx] != NULL.  match_kf[kf->n] = the
 * point this call writes to vpMatched[idx], or -1. */
int orbx_search_by_projection_sim3(int device, const orbx_frame_feats *kf, const orbx_proj_points *pts,
                                   const float *scale_factors, int nlevels, float th, int32_t *match_kf, int *nmatches);
/* The search half of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:873-1038; chi2 = 1,
 * max_dist = TH_LOW 50, aux = ur = u - bf*invz, inv_sigma2 = pKF->mvInvLevelSigma2), of Fuse(KeyFrame*, Scw, vpPoints, th,
 * vpReplacePoint) (:1040-1164; chi2 = 0, max_dist = 50) and of each direction of SearchBySim3 (max_dist = TH_HIGH 100):
 * best_idx[pts->n] = most similar keypoint of kf inside the window with octave in [level-1, level] whose distance is
 * <= max_dist, else -1; best_dist (may be NULL) its distance; *nfound = points with a result.  Points do not interact;
 * Replace / AddObservation / vpReplacePoint stay with the adaptor, which also re-checks isBad()/IsInKeyFrame() as
 * it walks the results in order (those flags can change through its own Replace calls). */
int orbx_window_best(int device, const orbx_frame_feats *kf, const orbx_proj_points *pts, const float *scale_factors,
                     const float *inv_sigma2, int nlevels, float th, int chi2, int max_dist, int32_t *best_idx,
                     int32_t *best_dist, int *nfound);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1166-1394; LoopClosing::ComputeSim3).
 * pts12 = KF1's map points (one entry per KF1 keypoint; valid = has a point, !vbAlreadyMatched1, !isBad(), depth and range
 * tests :1221-1259) projected into KF2 with level = PredictScale(dist3D, pKF2); pts21 the reverse direction.
 * match12[kf1->n] = idx2 where the two directions agree (:1375-1391), else -1; *nfound = the return value. */
int orbx_search_by_sim3(int device, const orbx_frame_feats *kf1, const orbx_frame_feats *kf2, const orbx_proj_points *pts12,
                        const orbx_proj_points *pts21, const float *scale_factors1, const float *scale_factors2, int nlevels,
                        float th, int32_t *match12, int *nfound);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:430-556;
 * Tracking::MonocularInitialization).  prev_matched_xy[f1->n][2] = vbPrevMatched; matches12[f1->n] = vnMatches12;
 * *nmatches = the return value.  The update of vbPrevMatched (:544-546: the matched F2 keypoint's position) is a
 * copy the adaptor does from matches12.  Only level-0 keypoints of either frame take part (:451-455). */
int orbx_search_for_initialization(int device, const orbx_frame_feats *f1, const orbx_frame_feats *f2,
                                   const float *prev_matched_xy, int window_size, float nnratio, int check_orientation,
                                   int32_t *matches12, int *nmatches);

/* ---- Frame::UndistortKeyPoints (src/Frame.cc:470-515) -------------------
 * cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on n keypoint positions xy[n][2] -> xy_out[n][2] (may alias).
 * fx, fy, cx, cy = mK's float entries; dist_coef = mDistCoef (k1 k2 p1 p2 [k3]).  dist_coef[0] == 0 copies (:472-476).
 * Frame::ComputeImageBounds (:517-552) is the same call on the four image corners followed by min/max on the host. */
int orbx_undistort_keypoints(int device, const float *xy, int n, float fx, float fy, float cx, float cy,
                             const float *dist_coef, int ndist, float *xy_out);

/* ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:266-340) -- */

/* Batched over map points: point p owns descriptors desc[off[p] .. off[p+1]) (its non-bad observations in the
 * reference's std::map iteration order, at most 256 each).  best_idx[p] = index (relative to off[p]) of the
 * descriptor with the smallest median Hamming distance to the others (first wins ties), or -1 for an empty
 * point.  Host pointers. */
int orbx_distinctive_descriptors(int device, const uint8_t *desc, const int32_t *off, int npoints, int32_t *best_idx);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------- */

enum { ORBX_STAGE_RESIZE = 0, ORBX_STAGE_FAST = 1, ORBX_STAGE_TREE = 2, ORBX_STAGE_DESC = 3,
       ORBX_STAGE_STEREO = 4, ORBX_STAGE_STEREO_CUT = 5, ORBX_STAGE_COUNT = 6 };
/* (ORBX_STAGE_STEREO_CUT is kept for the numbering only: the median cut of src/Frame.cc:737-750 now runs inside the stereo launch,
 * in the last workgroup of each pair, and the row table of :584-604 inside the descriptor launch -- their times are part of
 * ORBX_STAGE_STEREO / ORBX_STAGE_DESC and this stage reads 0.) */
/* enable: record HIP events around every kernel launch of this handle (on the launch stream) */
int orbx_profile_enable(orbx_extractor *e, int enable);
/* Which stages record events while profiling is enabled: bit ORBX_STAGE_* (default: all).  An event between two kernels costs a
 * few microseconds of idle GPU (barrier packet), so a throughput measurement selects only the kernel it needs the duration of. */
int orbx_profile_stages(orbx_extractor *e, unsigned stage_mask);
/* synchronises, then returns accumulated kernel time (ms) and launch count per stage since the
 * last reset; arrays of ORBX_STAGE_COUNT entries */
int orbx_profile_read(orbx_extractor *e, float *ms, int *launches, int reset);

/* debug/inspection (used by the parity tests to localise a mismatch): FAST candidates and
 * per-level keypoint counts of image `image_index` of the most recent extract call.
 * x,y are relative to (16,16) like the reference's vToDistributeKeys. */
int orbx_debug_candidates(orbx_extractor *e, int image_index, int level, int32_t *x, int32_t *y, int32_t *resp, int cap, int *n);
int orbx_debug_level_counts(orbx_extractor *e, int image_index, int32_t *counts /*[nlevels]*/);
/* which FAST kernel the most recent extraction launched: 1 = a cell per wave (or several waves per cell), 2 = a pair of cells per wave */
int orbx_debug_fast_form(const orbx_extractor *e);
/* test hook: the SearchByBoW kernels exist in a latency form (one 16-wave workgroup per pair) and a throughput form
 * (LDS distance table + row fixpoint); a call picks by problem size.  form = 1 / 2 forces the wave / table form for
 * every later call of the process, 0 restores the automatic choice.  Both forms return identical matches. */
int orbx_debug_set_bow_form(int form);
/* test hook: a per-call matcher search of ONE pair with at most 144 work items hands them to its kernel by value, in the kernel-argument
 * segment (one dependent PCIe read less per call); in_memory = 1 makes such calls read their items from mapped host memory as calls with
 * several pairs do, 0 restores the default.  Both forms return identical matches. */
int orbx_debug_set_match_items(int in_memory);
/* host phases of the calling thread's most recent per-call matcher search (orbx_match.hip), microseconds:
 * [0] prepare (node intersection, participation bytes, packing), [1] launch, [2] wait for the kernel's ticket, [3] copy-out */
int orbx_debug_match_timing(double *out4);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
