/*
 * orb_oracle.h — CPU restatement of the ORB-SLAM2 feature front end + Hamming matchers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (orb-slam2_amd/, include/)
 * includes, links or calls this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker / timed baseline.
 *
 * PARITY UNPINNED: the reference (zhuwsh/ORB-SLAM2) has no tests or golden vectors, and
 * its arithmetic partly lives in OpenCV (>=2.4.3, "tested with 2.4.11 and 3.2",
 * reference README.md:68), which is neither vendored nor installed here, so the reference
 * cannot be compiled (src/ORBextractor.cc:57 fails at #include <opencv2/core/core.hpp>).
 * This file restates the reference's own code (citations per function) plus the published
 * OpenCV-3.2 generic-C++ (non-IPP) algorithms for resize/FAST/GaussianBlur/fastAtan2/cvRound
 * (SURVEY.md Appendix B).  It is pinned only by the handful of known answers the reference
 * text holds (umax table, feature quotas, thresholds: see tests/test_oracle_known_answers.py).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == cv::KeyPoint, 28 bytes (reference include/ORBextractor.h:74-76 output type) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} oracle_kp;

typedef struct orb_oracle orb_oracle;

/* ORBextractor::ORBextractor, src/ORBextractor.cc:429-534 */
orb_oracle *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void oracle_destroy(orb_oracle *o);

/* tables (E0/E9): each array has nlevels entries */
int oracle_nlevels(const orb_oracle *o);
const float *oracle_scale_factors(const orb_oracle *o);
const float *oracle_inv_scale_factors(const orb_oracle *o);
const float *oracle_level_sigma2(const orb_oracle *o);
const float *oracle_inv_level_sigma2(const orb_oracle *o);
const int *oracle_features_per_level(const orb_oracle *o);
const int *oracle_umax(const orb_oracle *o); /* 16 entries */

/* ORBextractor::operator(), src/ORBextractor.cc:1261-1339.
 * Returns number of keypoints (<= cap) or <0 on error (-1 bad args, -2 cap too small,
 * -3 image too small for the cell grid: the reference divides by zero there). */
int oracle_extract(orb_oracle *o, const uint8_t *img, int w, int h, size_t stride,
                   oracle_kp *kps, uint8_t *desc, int cap);

/* stage dumps, valid after oracle_extract */
int oracle_level_dims(const orb_oracle *o, int level, int *w, int *h);
const uint8_t *oracle_level_pixels(const orb_oracle *o, int level); /* tightly packed w*h */
const uint8_t *oracle_level_blurred(const orb_oracle *o, int level); /* tightly packed w*h */
/* FAST candidates of a level in reference order, coordinates relative to (16,16) */
int oracle_level_candidates(const orb_oracle *o, int level, int *x, int *y, int *resp, int cap);
int oracle_level_nkeypoints(const orb_oracle *o, int level);

/* single primitives, exposed for unit tests */
int oracle_cv_round_f(float v);
float oracle_fast_atan2(float y, float x);
void oracle_sincos(float angle_rad, float *s, float *c);
int oracle_fast_score(const uint8_t *center, int stride, int threshold); /* 0 if not a corner */
void oracle_resize_linear(const uint8_t *src, int sw, int sh, size_t sstride,
                          uint8_t *dst, int dw, int dh, size_t dstride);
void oracle_gaussian_blur7(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride);
/* profile 0 = OpenCV <= 3.4.1 taps (the default everywhere), 1 = OpenCV >= 3.4.2 fixed-point taps */
void oracle_gaussian_taps(int profile, int taps[7]);
void oracle_gaussian_blur7_profile(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride, int profile);
int oracle_set_cv_profile(orb_oracle *o, int profile);
int oracle_distribute_octtree(const int *x, const int *y, const int *resp, int n,
                              int min_x, int max_x, int min_y, int max_y, int nfeat, int *out_idx, int cap);

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1733-1749 */
int oracle_hamming(const uint8_t *a, const uint8_t *b);

/* Frame::ComputeStereoMatches, src/Frame.cc:577-751.  L and R must each have run
 * oracle_extract on the left / right image (their unblurred pyramids are read). */
int oracle_stereo_match(const orb_oracle *L, const orb_oracle *R,
                        const oracle_kp *kL, const uint8_t *dL, int nL,
                        const oracle_kp *kR, const uint8_t *dR, int nR,
                        float bf, float min_z, float *u_right, float *depth);

/* DBoW2::FeatureVector flattened to CSR */
typedef struct {
    int n;                  /* features */
    const uint8_t *desc;    /* [n][32] */
    int nnodes;
    const uint32_t *node_id; /* [nnodes] ascending */
    const int32_t *node_off; /* [nnodes+1] */
    const uint32_t *feat;    /* [node_off[nnodes]] ascending inside each node */
    const uint8_t *flag;     /* [n] meaning depends on the search (see each function) */
    const float *angle;      /* [n] keypoint angle (degrees) */
    const float *x, *y;      /* [n] undistorted keypoint position (triangulation only) */
    const int32_t *octave;   /* [n] (triangulation only) */
    const float *u_right;    /* [n] (triangulation only), <0 = mono */
} oracle_featset;

/* ORBmatcher::SearchByBoW(KeyFrame*,Frame&,...), src/ORBmatcher.cc:171-303.
 * kf.flag[i]!=0 <=> KF feature i has a non-bad MapPoint.  match_f[nF] = KF index or -1. */
int oracle_search_by_bow_kf_f(const oracle_featset *kf, const oracle_featset *f,
                              float nnratio, int check_ori, int32_t *match_f);
/* ORBmatcher::SearchByBoW(KeyFrame*,KeyFrame*,...), src/ORBmatcher.cc:568-702.
 * flag!=0 <=> non-bad MapPoint on that side.  match12[n1] = KF2 index or -1. */
int oracle_search_by_bow_kf_kf(const oracle_featset *k1, const oracle_featset *k2,
                               float nnratio, int check_ori, int32_t *match12);
/* ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:704-871 (+CheckDistEpipolarLine :147-164).
 * flag!=0 <=> the feature already HAS a MapPoint (such features are skipped).
 * pairs[2*cap] receives (idx1, idx2) sorted by idx1; returns number of pairs. */
int oracle_search_for_triangulation(const oracle_featset *k1, const oracle_featset *k2,
                                    const float F12[9], float ex, float ey,
                                    const float *scale_factors2, const float *level_sigma2_2,
                                    float nnratio, int check_ori, int only_stereo,
                                    int32_t *pairs, int cap);

/* ---- DBoW2 vocabulary tree + transform (Frame::ComputeBoW, src/Frame.cc:459-466) */
typedef struct oracle_vocab oracle_vocab;
/* nodes in id order as TemplatedVocabulary::loadFromTextFile builds them
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1358-1445): node 0 is the root and is NOT in the arrays;
 * entry j describes node j+1: parent id, leaf flag, 32-byte descriptor, weight.  Children are appended to
 * their parent in id order; word ids are assigned to leaves in id order. */
oracle_vocab *oracle_vocab_create(int k, int L, int nnodes_minus_root, const int32_t *parent, const uint8_t *is_leaf,
                                  const uint8_t *desc, const double *weight);
oracle_vocab *oracle_vocab_load_text(const char *path);
void oracle_vocab_destroy(oracle_vocab *v);
int oracle_vocab_nodes(const oracle_vocab *v);
int oracle_vocab_words(const oracle_vocab *v);
/* TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup), TF_IDF weighting + L1 norm
 * (:1127-1194, :1218-1259; BowVector.cpp:33-77).  Per-feature outputs may be NULL.  BowVector: ascending word
 * ids with L1-normalised values; FeatureVector: CSR.  Capacities: n entries each (fv_node_off: n+1).
 * Returns 0. */
int oracle_bow_transform(const oracle_vocab *v, const uint8_t *desc, int n, int levelsup,
                         uint32_t *word_id, double *word_weight, uint32_t *node_id,
                         uint32_t *bow_id, double *bow_val, int *nbow,
                         uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes);

/* accumulation half of transform (BowVector::addWeight in feature order + normalize(L1), FeatureVector::addFeature);
 * features with word_weight[i] <= 0 are skipped (TemplatedVocabulary.h:1157).  Same output contract as above. */
int oracle_bow_accumulate(const uint32_t *word_id, const double *word_weight, const uint32_t *node_id, int n,
                          uint32_t *bow_id, double *bow_val, int *nbow,
                          uint32_t *fv_node_id, int32_t *fv_node_off, uint32_t *fv_feat, int *fv_nnodes);

/* MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:266-340, for one map point: desc[n][32] are the
 * descriptors of its (non-bad) observations in std::map iteration order; returns BestIdx (-1 if n == 0). */
int oracle_distinctive_descriptor(const uint8_t *desc, int n);

/* ---- projection-guided searches: the two per-frame tracking matchers */
typedef struct {            /* the current Frame */
    int n;
    const float *x, *y;     /* mvKeysUn[i].pt */
    const int32_t *octave;  /* mvKeysUn[i].octave */
    const float *angle;     /* mvKeysUn[i].angle */
    const float *u_right;   /* mvuRight */
    const uint8_t *desc;    /* mDescriptors [n][32] */
    const uint8_t *occupied;/* 1: mvpMapPoints[i] != NULL with Observations() > 0 before the call */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
} oracle_frame_feats;
typedef struct {            /* the projected map points, in the reference's iteration order */
    int n;
    const float *u, *v;     /* projection into the current frame (A: u,v of :1428-1429; B: mTrackProjX/Y) */
    const float *aux;       /* A: invzc (:1425) ; B: mTrackProjXR */
    const int32_t *level;   /* A: LastFrame.mvKeys[i].octave ; B: mnTrackScaleLevel */
    const float *angle;     /* A: LastFrame.mvKeysUn[i].angle */
    const float *view_cos;  /* B: mTrackViewCos */
    const uint8_t *desc;    /* pMP->GetDescriptor() [n][32] */
    const uint8_t *valid;   /* A: pMP && !mvbOutlier ; B: mbTrackInView && !isBad() */
    const uint8_t *has_obs; /* pMP->Observations() > 0 */
} oracle_proj_points;
/* Frame::AssignFeaturesToGrid + PosInGrid + GetFeaturesInArea restated inside both searches
 * (src/Frame.cc:261-279, :386-457).
 * ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1396-1553.
 * direction: 0 none, 1 bForward, 2 bBackward (:1412-1413, computed by the caller).
 * match_cur[cur->n] = index of the point finally held by each current feature, or -1; returns nmatches. */
int oracle_search_by_projection_last(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *scale_factors,
                                     float th, int direction, float mbf, int check_ori, int32_t *match_cur);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:48-129 */
int oracle_search_by_projection_points(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *scale_factors,
                                       float th, float nnratio, int32_t *match_cur);

/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1555-1685 */
int oracle_search_by_projection_keyframe(const oracle_frame_feats *cur, const oracle_proj_points *pts, const float *scale_factors,
                                         float th, int orb_dist, int check_ori, int32_t *match_cur);
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:305-415 */
int oracle_search_by_projection_sim3(const oracle_frame_feats *kf, const oracle_proj_points *pts, const float *scale_factors,
                                     float th, int32_t *match_kf);
/* search half of ORBmatcher::Fuse (both overloads, :873-1164) and of each direction of SearchBySim3 */
int oracle_window_best(const oracle_frame_feats *kf, const oracle_proj_points *pts, const float *scale_factors,
                       const float *inv_sigma2, float th, int chi2, int max_dist, int32_t *best_idx, int32_t *best_dist);
/* ORBmatcher::SearchBySim3, src/ORBmatcher.cc:1166-1394 */
int oracle_search_by_sim3(const oracle_frame_feats *kf1, const oracle_frame_feats *kf2, const oracle_proj_points *pts12,
                          const oracle_proj_points *pts21, const float *sf1, const float *sf2, float th, int32_t *match12);

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:430-556; prev_xy[f1->n][2] = vbPrevMatched (not updated) */
int oracle_search_for_initialization(const oracle_frame_feats *f1, const oracle_frame_feats *f2, const float *prev_xy,
                                     int window_size, float nnratio, int check_ori, int32_t *matches12);

/* cv::cvtColor(CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) for 8U, the conversion Tracking::GrabImage*
 * applies to colour input (src/Tracking.cc:177-202, :217-231, :254-268) [OpenCV generic path, from memory:
 * fixed point, yuv_shift 14, R2Y 4899, G2Y 9617, B2Y 1868, rounding 1<<13].  channels 3 or 4; rgb_order 1 = R first. */
void oracle_cvt_gray(const uint8_t *src, int w, int h, size_t sstride, int channels, int rgb_order, uint8_t *dst, size_t dstride);

/* cv::remap(..., INTER_LINEAR) with CV_32FC1 maps on 8UC1, BORDER_CONSTANT 0 (Examples/Stereo/stereo_euroc.cc:136-137) */
void oracle_remap_bilinear(const uint8_t *src, int sw, int sh, size_t sstride, const float *map_x, const float *map_y,
                           uint8_t *dst, int dw, int dh, size_t dstride);

/* cv::undistortPoints(pts, pts, K, D, noArray(), K), src/Frame.cc:470-515 */
void oracle_undistort_points(const float *xy, int n, float fx, float fy, float cx, float cy, const float *dist, int ndist, float *out);

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1687-1728 */
void oracle_three_maxima(const int *count, int L, int *ind1, int *ind2, int *ind3);

#ifdef __cplusplus
}
#endif
#endif
