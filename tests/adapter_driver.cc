// tests/adapter_driver.cc -- drives the compiled C++ adaptor (adapter/*.cc, the code a maintainer drops into ORB-SLAM2)
// the way the reference's Tracking / LocalMapping / LoopClosing threads drive ORBextractor, Frame::ComputeStereoMatches
// and ORBmatcher.  Built by tests/test_adapter.py with g++ against tests/cvstub (OpenCV is not installed) and linked
// with liborbx.so.
//
//   adapter_driver csr                 FeatureVector -> flatten() -> orbx_featset round trip (no GPU needed)
//   adapter_driver run in.bin out.txt  stereo pair through the adaptor classes; every result is written as
//                                      "name n v0 v1 ..." lines (floats as their uint32 bit patterns)
//   adapter_driver track in.bin voc.txt out.txt   the per-frame searches of Tracking (SearchByProjection x 2, SearchForInitialization),
//                                      Frame::ComputeBoW / UndistortKeyPoints and MapPoint::ComputeDistinctiveDescriptors through the
//                                      compiled adaptors; built with -DORBX_ADAPTER_CAPTURE so that the inputs each search handed to the
//                                      ABI are written next to its results (the test gives the CPU oracle the same inputs)
//   adapter_driver map in.bin out.txt  the relocalisation / local-mapping / loop-closing searches (adapter/ORBmatcher_fuse.cc) the same way
//   adapter_driver batch in.bin out.txt [reps]   the loops of LocalMapping::CreateNewMapPoints / LoopClosing::ComputeSim3 / Tracking::Relocalization over
//                                      20 derived keyframes through adapter/ORBmatcher_batch.cc (resident keyframes, one call per loop), the same
//                                      pairs through the single-pair adaptors (cache-aware), every keyframe's data written out for the oracle, and
//                                      adaptor-inclusive timings (flags + flatten + call + map-back, std::chrono) as "time_*" lines in nanoseconds
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <utility>
#include <vector>

#include <map>

#include "Frame.h"
#include "KeyFrame.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "orbx_adapter.h"
#include "orbx_batch.h"

#include <algorithm>
#include <chrono>

using namespace ORB_SLAM2;

float Frame::fx = 718.856f, Frame::fy = 718.856f, Frame::cx = 607.1928f, Frame::cy = 185.2157f;
float Frame::mnMinX = 0.f, Frame::mnMaxX = 1241.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 376.f;

namespace orbx_adapter { void DistinctiveDescriptors(const std::vector<ORB_SLAM2::MapPoint *> &points); }

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

template <class T> static void dump_i(FILE *f, const char *name, const std::vector<T> &v)
{
    fprintf(f, "%s %zu", name, v.size());
    for (size_t i = 0; i < v.size(); i++) fprintf(f, " %lld", (long long)v[i]);
    fprintf(f, "\n");
}
static void dump_f(FILE *f, const char *name, const std::vector<float> &v)
{
    fprintf(f, "%s %zu", name, v.size());
    for (size_t i = 0; i < v.size(); i++) fprintf(f, " %u", bits(v[i]));
    fprintf(f, "\n");
}
static void dump_keys(FILE *f, const char *name, const std::vector<cv::KeyPoint> &k)
{
    fprintf(f, "%s %zu", name, k.size() * 7);
    for (size_t i = 0; i < k.size(); i++)
        fprintf(f, " %u %u %u %u %u %d %d", bits(k[i].pt.x), bits(k[i].pt.y), bits(k[i].size), bits(k[i].angle), bits(k[i].response), k[i].octave, k[i].class_id);
    fprintf(f, "\n");
}
static void dump_mat(FILE *f, const char *name, const cv::Mat &m)
{
    fprintf(f, "%s %zu", name, m.total());
    for (int r = 0; r < m.rows; r++)
        for (int c = 0; c < m.cols; c++) fprintf(f, " %d", (int)m.at<uchar>(r, c));
    fprintf(f, "\n");
}

// the synthetic vocabulary node of a descriptor (the tests use the same rule): bytes 0 of the descriptor picks one of 16
// nodes, features whose byte 1 is a multiple of 8 are "stopped" (in no node)
static int g_fv_nodes = 16;     // batch mode: 100 (a FeatureVector at levelsup = 4 of the k = 10, L = 6 vocabulary has up to 100 nodes, src/Frame.cc:464)
static void feature_vector(const cv::Mat &desc, DBoW2::FeatureVector &fv)
{
    for (int i = 0; i < desc.rows; i++)
        if (desc.at<uchar>(i, 1) % 8 != 0) fv.addFeature(100 + (g_fv_nodes == 16 ? (desc.at<uchar>(i, 0) & 15) : desc.at<uchar>(i, 0) % g_fv_nodes), (unsigned)i);
}

static int mode_csr()
{
    DBoW2::FeatureVector fv;
    const unsigned node[] = { 7, 3, 7, 900000, 3, 3, 0, 7 };
    for (unsigned i = 0; i < 8; i++) fv.addFeature(node[i], 10 + i);
    const orbx_adapter::Csr c = orbx_adapter::flatten(fv);
    orbx_adapter::Side s;
    s.csr = c;
    static uint8_t desc[18 * 32];
    s.flag.assign(18, 1);
    s.angle.assign(18, 0.f);
    s.bind(desc, 18);
    printf("id");
    for (int i = 0; i < s.fs.nnodes; i++) printf(" %u", s.fs.node_id[i]);
    printf("\noff");
    for (int i = 0; i <= s.fs.nnodes; i++) printf(" %d", s.fs.node_off[i]);
    printf("\nfeat");
    for (int i = 0; i < s.fs.node_off[s.fs.nnodes]; i++) printf(" %u", s.fs.feat[i]);
    // the featset goes through the ABI: ORBX_OK with a GPU, ORBX_E_NO_DEVICE (-4) without one -- never a host fallback
    std::vector<int32_t> match(18);
    int n = -1;
    const int rc = orbx_search_by_bow_kf_f(0, &s.fs, &s.fs, 0.75f, 1, &match[0], &n);
    printf("\nrc %d\n", rc);
    return 0;
}

static void fill_keyframe(KeyFrame &kf, const std::vector<cv::KeyPoint> &keys, const cv::Mat &desc, const std::vector<float> &uRight,
                          std::vector<MapPoint> &store, int bad_every, int none_every, const std::vector<float> &sf, const std::vector<float> &s2)
{
    kf.N = (int)keys.size();
    kf.mvKeysUn = keys;
    kf.mDescriptors = desc;
    kf.mvuRight = uRight;
    kf.mvuRight.resize(kf.N, -1.0f);
    feature_vector(desc, kf.mFeatVec);
    store.clear();
    store.reserve(kf.N);
    kf.mvpMapPoints.assign(kf.N, static_cast<MapPoint *>(NULL));
    for (int i = 0; i < kf.N; i++) {
        store.push_back(MapPoint(i % bad_every == 1));
        if (i % none_every != 2) kf.mvpMapPoints[i] = &store[i];
    }
    kf.mvScaleFactors = sf;
    kf.mvLevelSigma2 = s2;
    kf.fx = 718.856f; kf.fy = 718.856f; kf.cx = 607.1928f; kf.cy = 185.2157f;
}

static int mode_run(const char *in_path, const char *out_path)
{
    FILE *fi = fopen(in_path, "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in_path); return 2; }
    int hdr[2];
    if (fread(hdr, 4, 2, fi) != 2) return 2;
    const int w = hdr[0], h = hdr[1];
    cv::Mat imL(h, w, CV_8UC1), imR(h, w, CV_8UC1);
    if (fread(imL.data, 1, (size_t)w * h, fi) != (size_t)w * h || fread(imR.data, 1, (size_t)w * h, fi) != (size_t)w * h) return 2;
    fclose(fi);
    FILE *fo = fopen(out_path, "w");
    if (!fo) return 2;

    // Tracking::Tracking (src/Tracking.cc:124-130) + Frame::Frame stereo (src/Frame.cc:60-120)
    ORBextractor exL(1000, 1.2f, 8, 20, 7), exR(1000, 1.2f, 8, 20, 7);
    Frame F;
    F.mpORBextractorLeft = &exL; F.mpORBextractorRight = &exR;
    F.mbf = 386.1448f;
    exL(imL, cv::Mat(), F.mvKeys, F.mDescriptors);         // Frame::ExtractORB(0, imLeft), src/Frame.cc:285-292
    exR(imR, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight);
    F.N = (int)F.mvKeys.size();
    F.mvKeysUn = F.mvKeys;
    F.ComputeStereoMatches();
    dump_keys(fo, "keysL", F.mvKeys); dump_mat(fo, "descL", F.mDescriptors);
    dump_keys(fo, "keysR", F.mvKeysRight); dump_mat(fo, "descR", F.mDescriptorsRight);
    dump_f(fo, "uRight", F.mvuRight); dump_f(fo, "depth", F.mvDepth);
    dump_f(fo, "scaleFactors", exL.GetScaleFactors()); dump_f(fo, "levelSigma2", exL.GetScaleSigmaSquares());
    dump_f(fo, "invScaleFactors", exL.GetInverseScaleFactors()); dump_f(fo, "invLevelSigma2", exL.GetInverseScaleSigmaSquares());
    exL.FetchPyramid();
    std::vector<int> dims;
    for (int l = 0; l < exL.GetLevels(); l++) { dims.push_back(exL.mvImagePyramid[l].cols); dims.push_back(exL.mvImagePyramid[l].rows); }
    dump_i(fo, "pyramidDims", dims);
    dump_mat(fo, "pyramid3", exL.mvImagePyramid[3]);

    // the same frame through the one-call stereo front end on ONE extractor
    {
        std::vector<cv::KeyPoint> kl, kr; cv::Mat dl, dr; std::vector<float> ur, dp;
        exL.ExtractStereo(imL, imR, F.mbf, F.mbf / Frame::fx, kl, dl, kr, dr, ur, dp);
        dump_keys(fo, "oneCallKeysL", kl); dump_mat(fo, "oneCallDescR", dr); dump_f(fo, "oneCallURight", ur);
    }

    // ORBmatcher: the left eye is the Frame, the right eye plays a KeyFrame, the left eye again a second KeyFrame
    feature_vector(F.mDescriptors, F.mFeatVec);
    KeyFrame kfR, kfL;
    std::vector<MapPoint> storeR, storeL;
    const std::vector<float> sf = exL.GetScaleFactors(), s2 = exL.GetScaleSigmaSquares();
    fill_keyframe(kfR, F.mvKeysRight, F.mDescriptorsRight, std::vector<float>(), storeR, 5, 3, sf, s2);
    fill_keyframe(kfL, F.mvKeys, F.mDescriptors, F.mvuRight, storeL, 6, 3, sf, s2);
    ORBmatcher matcher(0.75f, true);
    std::vector<MapPoint *> vpF;
    const int n1 = matcher.SearchByBoW(&kfR, F, vpF);
    std::vector<int> m1(vpF.size(), -1);
    for (size_t i = 0; i < vpF.size(); i++) if (vpF[i]) m1[i] = (int)(vpF[i] - &storeR[0]);
    dump_i(fo, "bowKfF", m1); dump_i(fo, "bowKfF_n", std::vector<int>(1, n1));
    std::vector<MapPoint *> vp12;
    const int n2 = matcher.SearchByBoW(&kfR, &kfL, vp12);
    std::vector<int> m2(vp12.size(), -1);
    for (size_t i = 0; i < vp12.size(); i++) if (vp12[i]) m2[i] = (int)(vp12[i] - &storeL[0]);
    dump_i(fo, "bowKfKf", m2); dump_i(fo, "bowKfKf_n", std::vector<int>(1, n2));
    // SearchForTriangulation: rectified stereo geometry, camera 2 (left) sits 0.54 m to the left of camera 1 (right)
    kfR.Ow = cv::Mat(3, 1, CV_32F); kfR.Ow.at<float>(0) = 0.5372f; kfR.Ow.at<float>(1) = 0.f; kfR.Ow.at<float>(2) = 0.02f;
    kfL.Rcw = cv::Mat(3, 3, CV_32F);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) kfL.Rcw.at<float>(r, c) = r == c ? 1.f : 0.f;
    kfL.tcw = cv::Mat(3, 1, CV_32F); kfL.tcw.at<float>(0) = 0.f; kfL.tcw.at<float>(1) = 0.f; kfL.tcw.at<float>(2) = 0.f;
    cv::Mat F12(3, 3, CV_32F);
    const float f12[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, -1.f / 718.856f, 0.f, 1.f / 718.856f, 0.f };   // y1 == y2 epipolar lines
    for (int i = 0; i < 9; i++) F12.at<float>(i / 3, i % 3) = f12[i];
    std::vector<std::pair<size_t, size_t> > pairs;
    ORBmatcher tri(0.6f, false);
    const int n3 = tri.SearchForTriangulation(&kfR, &kfL, F12, pairs, false);
    std::vector<long long> flat;
    for (size_t i = 0; i < pairs.size(); i++) { flat.push_back((long long)pairs[i].first); flat.push_back((long long)pairs[i].second); }
    dump_i(fo, "triPairs", flat); dump_i(fo, "tri_n", std::vector<int>(1, n3));
    fclose(fo);
    printf("adaptor run ok: %d left / %d right keypoints, bow %d / %d, triangulation pairs %d\n", F.N, (int)F.mvKeysRight.size(), n1, n2, n3);
    return 0;
}


#ifdef ORBX_ADAPTER_CAPTURE
static void dump_u8(FILE *f, const char *name, const std::vector<uint8_t> &v)
{
    fprintf(f, "%s %zu", name, v.size());
    for (size_t i = 0; i < v.size(); i++) fprintf(f, " %d", (int)v[i]);
    fprintf(f, "\n");
}
static void dump_capture(FILE *fo, const std::string &pre, bool points, int slot = 0)
{
    const orbx_adapter::Capture &c = orbx_adapter::capture(slot);
    dump_f(fo, (pre + "cur_x").c_str(), c.cx); dump_f(fo, (pre + "cur_y").c_str(), c.cy); dump_f(fo, (pre + "cur_angle").c_str(), c.cangle);
    dump_f(fo, (pre + "cur_uright").c_str(), c.curight); dump_i(fo, (pre + "cur_octave").c_str(), c.coctave);
    dump_u8(fo, (pre + "cur_occupied").c_str(), c.coccupied); dump_u8(fo, (pre + "cur_desc").c_str(), c.cdesc);
    dump_f(fo, (pre + "cur_bounds").c_str(), std::vector<float>(c.bounds, c.bounds + 4));
    if (!points) return;
    dump_f(fo, (pre + "pts_u").c_str(), c.pu); dump_f(fo, (pre + "pts_v").c_str(), c.pv); dump_f(fo, (pre + "pts_aux").c_str(), c.paux);
    dump_f(fo, (pre + "pts_angle").c_str(), c.pangle); dump_f(fo, (pre + "pts_view").c_str(), c.pview); dump_i(fo, (pre + "pts_level").c_str(), c.plevel);
    dump_u8(fo, (pre + "pts_desc").c_str(), c.pdesc); dump_u8(fo, (pre + "pts_valid").c_str(), c.pvalid); dump_u8(fo, (pre + "pts_has_obs").c_str(), c.phas_obs);
}

// per feature: the index (inside `store`) of the MapPoint it holds, -1 none, -2 a point from elsewhere (it held it before the search)
static std::vector<int> held(const std::vector<MapPoint *> &vp, const std::vector<MapPoint> &store)
{
    std::vector<int> out(vp.size(), -1);
    for (size_t i = 0; i < vp.size(); i++)
        if (vp[i]) out[i] = (vp[i] >= &store[0] && vp[i] < &store[0] + store.size()) ? (int)(vp[i] - &store[0]) : -2;
    return out;
}

static cv::Mat pose(float tx, float ty, float tz)
{
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.at<float>(r, c) = r == c ? 1.f : 0.f;
    // a small rotation about y so that Rcw is not the identity
    const float a = 0.01f, ca = 0.99995f, sa = 0.0099998f; (void)a;
    T.at<float>(0, 0) = ca; T.at<float>(0, 2) = sa; T.at<float>(2, 0) = -sa; T.at<float>(2, 2) = ca;
    T.at<float>(0, 3) = tx; T.at<float>(1, 3) = ty; T.at<float>(2, 3) = tz;
    return T;
}

static int mode_track(const char *in_path, const char *voc_path, const char *out_path)
{
    FILE *fi = fopen(in_path, "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in_path); return 2; }
    int hdr[2];
    if (fread(hdr, 4, 2, fi) != 2) return 2;
    const int w = hdr[0], h = hdr[1];
    cv::Mat imL(h, w, CV_8UC1), imR(h, w, CV_8UC1);
    if (fread(imL.data, 1, (size_t)w * h, fi) != (size_t)w * h || fread(imR.data, 1, (size_t)w * h, fi) != (size_t)w * h) return 2;
    fclose(fi);
    FILE *fo = fopen(out_path, "w");
    if (!fo) return 2;
    ORBextractor exL(1000, 1.2f, 8, 20, 7), exR(1000, 1.2f, 8, 20, 7);
    Frame cur, last;
    exL(imL, cv::Mat(), cur.mvKeys, cur.mDescriptors);
    exR(imR, cv::Mat(), last.mvKeys, last.mDescriptors);
    const std::vector<float> sf = exL.GetScaleFactors();
    Frame *fr[2] = { &cur, &last };
    for (int k = 0; k < 2; k++) {
        Frame &F = *fr[k];
        F.N = (int)F.mvKeys.size();
        F.mvKeysUn = F.mvKeys;
        F.mvScaleFactors = sf; F.mnScaleLevels = 8; F.mbf = 386.1448f; F.mb = F.mbf / Frame::fx;
        F.mvuRight.assign(F.N, -1.0f);
        for (int i = 0; i < F.N; i += 4) F.mvuRight[i] = F.mvKeys[i].pt.x - 7.5f;
        F.mvpMapPoints.assign(F.N, static_cast<MapPoint *>(NULL));
        F.mvbOutlier.assign(F.N, false);
    }
    // ---- the last frame's map points: each right-eye keypoint backprojected into the CURRENT camera 12 px to the right of where
    // the right eye saw it, at 8 .. 14 m, and expressed in world coordinates through the current pose
    std::vector<MapPoint> store(last.N);
    std::vector<MapPoint> old(cur.N);           // points the current frame already holds
    for (int i = 0; i < last.N; i++) {
        MapPoint &P = store[i];
        P.mDescriptor = last.mDescriptors.row(i).clone();
        P.nObs = i % 3;
        if (i % 5 != 3) last.mvpMapPoints[i] = &P;
        last.mvbOutlier[i] = i % 9 == 4;
    }
    for (int i = 0; i < cur.N; i += 11) { old[i].nObs = i % 22 == 0 ? 2 : 0; cur.mvpMapPoints[i] = &old[i]; }
    dump_keys(fo, "keysCur", cur.mvKeys); dump_keys(fo, "keysLast", last.mvKeys);
    const std::vector<MapPoint *> cur_before = cur.mvpMapPoints;
    const float tz[2] = { -0.2f, -0.9f };       // |twc.z| below / above mb = 0.537: the plain window, then the forward (bForward) one
    for (int variant = 0; variant < 2; variant++) {
        cur.mTcw = pose(0.05f, -0.02f, tz[variant]);
        last.mTcw = pose(0.f, 0.f, 0.f);
        for (int i = 0; i < last.N; i++) {
            const float z = 8.f + (float)(i % 7), u = last.mvKeys[i].pt.x + 12.f, v = last.mvKeys[i].pt.y;
            const float xc = (u - Frame::cx) / Frame::fx * z, yc = (v - Frame::cy) / Frame::fy * z;
            // x3Dw = Rcw^T (x3Dc - tcw)
            cv::Mat d(3, 1, CV_32F);
            d.at<float>(0) = xc - cur.mTcw.at<float>(0, 3); d.at<float>(1) = yc - cur.mTcw.at<float>(1, 3); d.at<float>(2) = z - cur.mTcw.at<float>(2, 3);
            store[i].mWorldPos = cur.mTcw.rowRange(0, 3).colRange(0, 3).t() * d;
        }
        cur.mvpMapPoints = cur_before;
        ORBmatcher m(0.9f, true);
        const int n = m.SearchByProjection(cur, last, 15.f, variant == 0);
        const std::string pre = variant == 0 ? "lastA_" : "lastB_";
        dump_capture(fo, pre, true);
        dump_i(fo, (pre + "held").c_str(), held(cur.mvpMapPoints, store)); dump_i(fo, (pre + "n").c_str(), std::vector<int>(1, n));
        dump_i(fo, (pre + "mono").c_str(), std::vector<int>(1, variant == 0));
    }
    // ---- Tracking::SearchLocalPoints: map points with their tracking variables set by Frame::isInFrustum
    {
        std::vector<MapPoint> local(last.N);
        std::vector<MapPoint *> vp(last.N);
        for (int i = 0; i < last.N; i++) {
            MapPoint &P = local[i];
            P.mDescriptor = last.mDescriptors.row(i).clone();
            P.nObs = i % 4;
            P.mbTrackInView = i % 7 != 3;
            P.mbBad = i % 13 == 5;
            P.mTrackProjX = last.mvKeys[i].pt.x + 10.f; P.mTrackProjY = last.mvKeys[i].pt.y + 0.5f; P.mTrackProjXR = P.mTrackProjX - 7.f;
            P.mnTrackScaleLevel = last.mvKeys[i].octave;
            P.mTrackViewCos = i % 2 ? 0.999f : 0.9f;
            vp[i] = &P;
        }
        cur.mvpMapPoints = cur_before;
        ORBmatcher m(0.8f, true);
        const int n = m.SearchByProjection(cur, vp, 3.f);
        dump_capture(fo, "local_", true);
        dump_i(fo, "local_held", held(cur.mvpMapPoints, local)); dump_i(fo, "local_n", std::vector<int>(1, n));
    }
    // ---- Tracking::MonocularInitialization
    {
        std::vector<cv::Point2f> prev(cur.N);
        for (int i = 0; i < cur.N; i++) prev[i] = cur.mvKeysUn[i].pt;
        std::vector<int> m12;
        ORBmatcher m(0.9f, true);
        const int n = m.SearchForInitialization(cur, last, prev, m12, 100);
        dump_capture(fo, "init_", false);
        dump_i(fo, "init_m12", m12); dump_i(fo, "init_n", std::vector<int>(1, n));
        std::vector<float> pv;
        for (int i = 0; i < cur.N; i++) { pv.push_back(prev[i].x); pv.push_back(prev[i].y); }
        dump_f(fo, "init_prev_after", pv);
    }
    // ---- Frame::ComputeBoW with the vocabulary file the test wrote
    if (orbx_adapter::LoadVocabulary(voc_path) != ORBX_OK) { fprintf(stderr, "vocabulary: %s\n", orbx_last_error()); return 4; }
    cur.ComputeBoW();
    {
        std::vector<long long> ids, fid, foff(1, 0), ffeat; std::vector<double> vals;
        for (DBoW2::BowVector::const_iterator it = cur.mBowVec.begin(); it != cur.mBowVec.end(); ++it) { ids.push_back(it->first); vals.push_back(it->second); }
        for (DBoW2::FeatureVector::const_iterator it = cur.mFeatVec.begin(); it != cur.mFeatVec.end(); ++it) {
            fid.push_back(it->first);
            for (size_t k = 0; k < it->second.size(); k++) ffeat.push_back(it->second[k]);
            foff.push_back((long long)ffeat.size());
        }
        dump_i(fo, "bow_id", ids); dump_i(fo, "fv_id", fid); dump_i(fo, "fv_off", foff); dump_i(fo, "fv_feat", ffeat);
        fprintf(fo, "bow_val %zu", vals.size());
        for (size_t i = 0; i < vals.size(); i++) { unsigned long long u; memcpy(&u, &vals[i], 8); fprintf(fo, " %llu", u); }
        fprintf(fo, "\n");
    }
    // ---- Frame::UndistortKeyPoints (a TUM-like lens)
    {
        cur.mK = cv::Mat(3, 3, CV_32F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cur.mK.at<float>(r, c) = 0.f;
        cur.mK.at<float>(0, 0) = 517.3f; cur.mK.at<float>(1, 1) = 516.5f; cur.mK.at<float>(0, 2) = 318.6f; cur.mK.at<float>(1, 2) = 255.3f; cur.mK.at<float>(2, 2) = 1.f;
        cur.mDistCoef = cv::Mat(5, 1, CV_32F);
        const float dc[5] = { 0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f };
        for (int i = 0; i < 5; i++) cur.mDistCoef.at<float>(i) = dc[i];
        cur.UndistortKeyPoints();
        std::vector<float> xy;
        for (int i = 0; i < cur.N; i++) { xy.push_back(cur.mvKeysUn[i].pt.x); xy.push_back(cur.mvKeysUn[i].pt.y); }
        dump_f(fo, "undist_xy", xy);
    }
    // ---- MapPoint::ComputeDistinctiveDescriptors: points seen by "keyframes" = the two eyes and a third with mixed rows
    {
        KeyFrame kfs[3];
        kfs[0].mDescriptors = cur.mDescriptors; kfs[1].mDescriptors = last.mDescriptors;
        kfs[2].mDescriptors = cur.mDescriptors.clone();
        for (int i = 0; i + 1 < kfs[2].mDescriptors.rows; i += 2) memcpy(kfs[2].mDescriptors.ptr<uchar>(i), last.mDescriptors.ptr<uchar>(i % last.N), 32);
        kfs[1].mbBad = false;
        const int npts = 40;
        std::vector<MapPoint> pts(npts);
        std::vector<MapPoint *> pp(npts);
        std::vector<long long> obs;
        for (int p = 0; p < npts; p++) {
            const int nobs = 1 + p % 9;
            for (int o = 0; o < nobs && o < 3; o++) pts[p].mObservations[&kfs[o]] = (size_t)((p * 7 + o * 13) % std::min(cur.N, last.N));
            pts[p].mbBad = p == 17;
            pp[p] = &pts[p];
        }
        kfs[1].mbBad = true;        // its observations are left out (:287)
        for (int p = 0; p < npts; p += 2) pts[p].ComputeDistinctiveDescriptors();           // one at a time ...
        std::vector<MapPoint *> rest;
        for (int p = 1; p < npts; p += 2) rest.push_back(pp[p]);
        orbx_adapter::DistinctiveDescriptors(rest);                                          // ... and the batched form
        fprintf(fo, "distinct_obs %d", npts * 7);
        for (int p = 0; p < npts; p++) {
            // per point: bad flag, then (keyframe index, row) of up to three observations in the order of the std::map walk
            fprintf(fo, " %d", (int)pts[p].mbBad);
            int k = 0;
            for (std::map<KeyFrame *, size_t>::iterator it = pts[p].mObservations.begin(); it != pts[p].mObservations.end(); ++it, ++k)
                fprintf(fo, " %d %d", (int)(it->first - &kfs[0]), (int)it->second);
            for (; k < 3; k++) fprintf(fo, " -1 -1");
        }
        fprintf(fo, "\n");
        fprintf(fo, "distinct_desc %d", npts * 33);
        for (int p = 0; p < npts; p++) {
            fprintf(fo, " %d", pts[p].mDescriptor.empty() ? 0 : 1);
            for (int b = 0; b < 32; b++) fprintf(fo, " %d", pts[p].mDescriptor.empty() ? 0 : (int)pts[p].mDescriptor.data[b]);
        }
        fprintf(fo, "\n");
        dump_mat(fo, "kf2_desc", kfs[2].mDescriptors);
    }
    dump_mat(fo, "descCur", cur.mDescriptors); dump_mat(fo, "descLast", last.mDescriptors);
    fclose(fo);
    printf("adaptor track ok: %d / %d keypoints\n", cur.N, last.N);
    return 0;
}
// MapPoint::PredictScale (reference src/MapPoint.cc:393-415): level = ceil(log(maxDistance / dist) / log(scaleFactor)), clamped
namespace ORB_SLAM2 {
static int predict(float maxd, float dist, float logsf, int nlevels)
{
    const float ratio = maxd / dist;
    int n = (int)ceil(log(ratio) / logsf);
    if (n < 0) n = 0; else if (n >= nlevels) n = nlevels - 1;
    return n;
}
int MapPoint::PredictScale(const float &currentDist, KeyFrame *pKF) { return predict(mfMaxDistance, currentDist, pKF->mfLogScaleFactor, pKF->mnScaleLevels); }
int MapPoint::PredictScale(const float &currentDist, Frame *pF) { return predict(mfMaxDistance, currentDist, pF->mfLogScaleFactor, pF->mnScaleLevels); }
}

static void dump_points(FILE *fo, const char *name, const std::vector<MapPoint> &st, KeyFrame *pKF)
{
    // per point: bad, index of the point it was replaced by (-1 none; -2 a point of another store), observation index in pKF (-1 none), Observations()
    fprintf(fo, "%s %zu", name, st.size() * 4);
    for (size_t i = 0; i < st.size(); i++) {
        MapPoint &P = const_cast<MapPoint &>(st[i]);
        long long rep = -1;
        if (P.mpReplaced) rep = (P.mpReplaced >= &st[0] && P.mpReplaced < &st[0] + st.size()) ? (long long)(P.mpReplaced - &st[0]) : -2;
        fprintf(fo, " %d %lld %d %d", (int)P.mbBad, rep, P.GetIndexInKeyFrame(pKF), P.nObs);
    }
    fprintf(fo, "\n");
}

// local mapping / loop closing / relocalisation searches through adapter/ORBmatcher_fuse.cc
static int mode_map(const char *in_path, const char *out_path)
{
    FILE *fi = fopen(in_path, "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in_path); return 2; }
    int hdr[2];
    if (fread(hdr, 4, 2, fi) != 2) return 2;
    const int w = hdr[0], h = hdr[1];
    cv::Mat imL(h, w, CV_8UC1), imR(h, w, CV_8UC1);
    if (fread(imL.data, 1, (size_t)w * h, fi) != (size_t)w * h || fread(imR.data, 1, (size_t)w * h, fi) != (size_t)w * h) return 2;
    fclose(fi);
    FILE *fo = fopen(out_path, "w");
    if (!fo) return 2;
    ORBextractor exL(1000, 1.2f, 8, 20, 7), exR(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> keysL, keysR;
    cv::Mat descL, descR;
    exL(imL, cv::Mat(), keysL, descL);
    exR(imR, cv::Mat(), keysR, descR);
    const std::vector<float> sf = exL.GetScaleFactors(), s2 = exL.GetScaleSigmaSquares(), is2 = exL.GetInverseScaleSigmaSquares();
    dump_keys(fo, "keysL", keysL); dump_keys(fo, "keysR", keysR);
    const int nL = (int)keysL.size(), nR = (int)keysR.size();
    const float logsf = logf(1.2f);
    // the keyframe the points are projected into: the left eye, with a pose
    KeyFrame kfL, kfR;
    std::vector<MapPoint> storeL, storeR;
    std::vector<float> urL(nL, -1.f);
    for (int i = 0; i < nL; i += 3) urL[i] = keysL[i].pt.x - 6.f;
    fill_keyframe(kfL, keysL, descL, urL, storeL, 6, 3, sf, s2);
    fill_keyframe(kfR, keysR, descR, std::vector<float>(), storeR, 5, 3, sf, s2);
    KeyFrame *kk[2] = { &kfL, &kfR };
    for (int k = 0; k < 2; k++) {
        KeyFrame &K = *kk[k];
        K.mbf = 386.1448f; K.mnScaleLevels = 8; K.mfLogScaleFactor = logsf; K.mvInvLevelSigma2 = is2;
        K.mnMinX = 0; K.mnMinY = 0; K.mnMaxX = w; K.mnMaxY = h;
        const cv::Mat T = pose(k == 0 ? 0.03f : -0.02f, 0.01f, k == 0 ? -0.1f : 0.05f);
        K.Rcw = T.rowRange(0, 3).colRange(0, 3).clone(); K.tcw = T.rowRange(0, 3).col(3).clone();
        K.Ow = -K.Rcw.t() * K.tcw;
    }
    // map points of the RIGHT keyframe's features, placed in the world so that they project into the LEFT keyframe 9 px right of their
    // right-eye position at 6 .. 11 m; maxDistance chosen so that PredictScale returns about the feature's octave
    for (int i = 0; i < nR; i++) {
        MapPoint &P = storeR[i];
        const float z = 6.f + (float)(i % 6), u = keysR[i].pt.x + 9.f, v = keysR[i].pt.y;
        cv::Mat pc(3, 1, CV_32F);
        pc.at<float>(0) = (u - kfL.cx) / kfL.fx * z; pc.at<float>(1) = (v - kfL.cy) / kfL.fy * z; pc.at<float>(2) = z;
        P.mWorldPos = kfL.Rcw.t() * (pc - kfL.tcw);
        const cv::Mat PO = P.mWorldPos - kfL.Ow;
        const float dist = (float)cv::norm(PO);
        P.mNormalVector = (i % 10 == 7 ? -1.0 : 1.0) / dist * PO;      // every tenth point is seen from behind: rejected by the viewing-angle test
        P.mfMaxDistance = dist * powf(1.2f, (float)keysR[i].octave) * 0.95f;
        P.mfMinDistance = i % 17 == 3 ? dist * 2.f : 0.05f;            // some fail the distance range
        P.mDescriptor = descR.row(i).clone();
        P.nObs = 1 + i % 4;
        P.mObservations[&kfR] = (size_t)i;
    }
    for (int i = 0; i < nL; i++) {                                      // the left keyframe's own points (what Fuse may replace / be replaced by)
        storeL[i].mDescriptor = descL.row(i).clone(); storeL[i].nObs = 1 + i % 5; storeL[i].mObservations[&kfL] = (size_t)i;
        storeL[i].mWorldPos = cv::Mat(3, 1, CV_32F);
        const float z = 7.f + (float)(i % 5);
        cv::Mat pc(3, 1, CV_32F);
        pc.at<float>(0) = (keysL[i].pt.x - 9.f - kfL.cx) / kfL.fx * z; pc.at<float>(1) = (keysL[i].pt.y - kfL.cy) / kfL.fy * z; pc.at<float>(2) = z;
        storeL[i].mWorldPos = kfL.Rcw.t() * (pc - kfL.tcw);
        storeL[i].mfMaxDistance = (float)cv::norm(storeL[i].mWorldPos - kfL.Ow) * powf(1.2f, (float)keysL[i].octave) * 0.95f;
        storeL[i].mfMinDistance = 0.05f;
        storeL[i].mNormalVector = (1.0 / cv::norm(storeL[i].mWorldPos - kfL.Ow)) * (storeL[i].mWorldPos - kfL.Ow);
    }
    const std::vector<MapPoint> storeR0 = storeR, storeL0 = storeL;
    const std::vector<MapPoint *> mpL0 = kfL.mvpMapPoints, mpR0 = kfR.mvpMapPoints;
    std::vector<MapPoint *> vpR(nR);
    for (int i = 0; i < nR; i++) vpR[i] = &storeR[i];

    // ---- 1. relocalisation: SearchByProjection(CurrentFrame, pKF = the right keyframe, sAlreadyFound, th, ORBdist)
    {
        Frame cur;
        cur.N = nL; cur.mvKeys = keysL; cur.mvKeysUn = keysL; cur.mDescriptors = descL; cur.mvuRight = urL;
        cur.mvScaleFactors = sf; cur.mnScaleLevels = 8; cur.mfLogScaleFactor = logsf; cur.mbf = 386.1448f; cur.mb = 0.537f;
        cur.mTcw = pose(0.03f, 0.01f, -0.1f);                          // the left keyframe's pose
        cur.mvpMapPoints.assign(nL, static_cast<MapPoint *>(NULL));
        for (int i = 0; i < nL; i += 13) cur.mvpMapPoints[i] = &storeL[i];
        std::set<MapPoint *> found;
        for (int i = 0; i < nR; i += 19) found.insert(&storeR[i]);
        ORBmatcher m(0.9f, true);
        const int n = m.SearchByProjection(cur, &kfR, found, 10.f, 100);
        dump_capture(fo, "reloc_", true);
        dump_i(fo, "reloc_held", held(cur.mvpMapPoints, storeR)); dump_i(fo, "reloc_n", std::vector<int>(1, n));
    }
    // ---- 2. loop closing: SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)
    cv::Mat Scw = pose(0.03f, 0.01f, -0.1f);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Scw.at<float>(r, c) *= 1.1f;   // s [R | t]
    {
        std::vector<MapPoint *> matched(nL, static_cast<MapPoint *>(NULL));
        for (int i = 0; i < nL; i += 9) matched[i] = &storeL[i];
        matched[1] = &storeR[40];                                       // a point of vpPoints that is already matched: skipped (:331)
        ORBmatcher m(0.75f, true);
        const int n = m.SearchByProjection(&kfL, Scw, vpR, matched, 10);
        dump_capture(fo, "sim3p_", true);
        dump_i(fo, "sim3p_held", held(matched, storeR)); dump_i(fo, "sim3p_n", std::vector<int>(1, n));
    }
    // ---- 3. local mapping: Fuse(pKF, vpMapPoints, th)
    {
        std::vector<MapPoint *> vp = vpR;
        for (int i = 0; i < nR; i += 23) vp[i] = NULL;
        for (int i = 5; i < nR; i += 29) storeR[i].mObservations[&kfL] = 0;   // already in the keyframe
        ORBmatcher m(0.6f, true);
        const int n = m.Fuse(&kfL, vp, 3.f);
        dump_capture(fo, "fuse_", true);
        dump_i(fo, "fuse_n", std::vector<int>(1, n));
        dump_points(fo, "fuse_ptsR", storeR, &kfL); dump_points(fo, "fuse_ptsL", storeL, &kfL);
        dump_i(fo, "fuse_kf_points", held(kfL.mvpMapPoints, storeR));
        storeR = storeR0; storeL = storeL0; kfL.mvpMapPoints = mpL0;
    }
    // ---- 4. loop closing: Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    {
        std::vector<MapPoint *> repl(nR, static_cast<MapPoint *>(NULL));
        ORBmatcher m(0.8f, true);
        const int n = m.Fuse(&kfL, Scw, vpR, 4.f, repl);
        dump_capture(fo, "fuse2_", true);
        dump_i(fo, "fuse2_n", std::vector<int>(1, n));
        dump_i(fo, "fuse2_replace", held(repl, storeL));
        dump_points(fo, "fuse2_ptsR", storeR, &kfL);
        dump_i(fo, "fuse2_kf_points", held(kfL.mvpMapPoints, storeR));
        storeR = storeR0; storeL = storeL0; kfL.mvpMapPoints = mpL0;
    }
    // ---- 5. loop closing: SearchBySim3(pKF1 = left, pKF2 = right, vpMatches12, s12, R12, t12, th)
    {
        // the right keyframe's OWN features carry points that project into the left one (above); give the left keyframe's points the
        // mirror property: storeL[i] projects into the right keyframe 9 px left of its left-eye position
        kfR.Rcw = kfL.Rcw.clone(); kfR.tcw = kfL.tcw.clone(); kfR.Ow = kfL.Ow.clone();     // same pose: T12 = identity, s12 = 1
        cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R12.at<float>(r, c) = r == c ? 1.f : 0.f; t12.at<float>(r) = 0.f; }
        std::vector<MapPoint *> m12(nL, static_cast<MapPoint *>(NULL));
        for (int i = 0; i < nL; i += 31) if (kfL.mvpMapPoints[i]) m12[i] = &storeR[(i * 3) % nR];   // already matched pairs
        ORBmatcher m(0.75f, true);
        const int n = m.SearchBySim3(&kfL, &kfR, m12, 1.0f, R12, t12, 7.5f);
        dump_capture(fo, "sim3_12_", true, 0); dump_capture(fo, "sim3_21_", true, 1);
        dump_i(fo, "sim3_held", held(m12, storeR)); dump_i(fo, "sim3_n", std::vector<int>(1, n));
    }
    dump_mat(fo, "descL", descL); dump_mat(fo, "descR", descR);
    fclose(fo);
    printf("adaptor map ok: %d / %d keypoints\n", nL, nR);
    return 0;
}
#endif


// ---------------------------------------------------------------- batch mode: resident keyframes + the three loops as one call each

// keyframe k of the batch scene: the features of eye (k odd: right, even: left) rotated by 37 k positions, ~4 % of the descriptor bits flipped by
// a per-keyframe LCG (so that matches exist but differ), map-point pattern (bad_every, none_every) depending on k
static void derived_keyframe(KeyFrame &kf, std::vector<MapPoint> &store, int k, const std::vector<cv::KeyPoint> &keys, const cv::Mat &desc,
                             const std::vector<float> &uRight, const std::vector<float> &sf, const std::vector<float> &s2)
{
    const int n = (int)keys.size(), rot = n ? (37 * k) % n : 0;
    std::vector<cv::KeyPoint> kk(n);
    cv::Mat dd(n, 32, CV_8UC1);
    std::vector<float> ur(n, -1.0f);
    uint32_t lcg = 12345u + 977u * (uint32_t)k;
    for (int i = 0; i < n; i++) {
        const int j = (i + rot) % n;
        kk[i] = keys[j];
        if (j < (int)uRight.size()) ur[i] = uRight[j];
        for (int b = 0; b < 32; b++) {
            uint8_t v = desc.at<uchar>(j, b);
            if (k > 0 && b >= 2) {           // bytes 0 / 1 pick the synthetic vocabulary node: kept, so that the keyframes share nodes
                lcg = lcg * 1664525u + 1013904223u;
                if ((lcg >> 24) < 82) v ^= (uint8_t)(1u << ((lcg >> 16) & 7));      // one bit in ~ a third of the bytes
            }
            dd.at<uchar>(i, b) = v;
        }
    }
    fill_keyframe(kf, kk, dd, ur, store, 4 + k % 5, 3 + k % 3, sf, s2);
}

static void dump_keyframe(FILE *fo, const char *pre, KeyFrame &kf)
{
    char nm[64];
    snprintf(nm, sizeof nm, "%skeys", pre); dump_keys(fo, nm, kf.mvKeysUn);
    snprintf(nm, sizeof nm, "%sdesc", pre); dump_mat(fo, nm, kf.mDescriptors);
    snprintf(nm, sizeof nm, "%suright", pre); dump_f(fo, nm, kf.mvuRight);
    std::vector<int> good(kf.N), has(kf.N);
    for (int i = 0; i < kf.N; i++) { has[i] = kf.mvpMapPoints[i] ? 1 : 0; good[i] = kf.mvpMapPoints[i] && !kf.mvpMapPoints[i]->isBad() ? 1 : 0; }
    snprintf(nm, sizeof nm, "%sgood", pre); dump_i(fo, nm, good);
    snprintf(nm, sizeof nm, "%shas", pre); dump_i(fo, nm, has);
}

static std::vector<int> held_in(const std::vector<MapPoint *> &v, const std::vector<MapPoint> &store)
{
    std::vector<int> o(v.size(), -1);
    for (size_t i = 0; i < v.size(); i++) if (v[i]) o[i] = (int)(v[i] - &store[0]);
    return o;
}

static long long median_ns(std::vector<long long> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

static int mode_batch(const char *in_path, const char *out_path, int reps)
{
    FILE *fi = fopen(in_path, "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in_path); return 2; }
    int hdr[2];
    if (fread(hdr, 4, 2, fi) != 2) return 2;
    const int w = hdr[0], h = hdr[1];
    cv::Mat imL(h, w, CV_8UC1), imR(h, w, CV_8UC1);
    if (fread(imL.data, 1, (size_t)w * h, fi) != (size_t)w * h || fread(imR.data, 1, (size_t)w * h, fi) != (size_t)w * h) return 2;
    fclose(fi);
    FILE *fo = fopen(out_path, "w");
    if (!fo) return 2;
    ORBextractor exL(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kl, kr; cv::Mat dl, dr; std::vector<float> ur, dp;
    exL.ExtractStereo(imL, imR, 386.1448f, 386.1448f / Frame::fx, kl, dl, kr, dr, ur, dp);
    const std::vector<float> sf = exL.GetScaleFactors(), s2 = exL.GetScaleSigmaSquares();
    dump_f(fo, "scaleFactors", sf); dump_f(fo, "levelSigma2", s2);

    const int NK = 20;
    g_fv_nodes = 100;
    KeyFrame cur;
    std::vector<MapPoint> storeC;
    derived_keyframe(cur, storeC, 0, kl, dl, ur, sf, s2);
    std::vector<KeyFrame> nb(NK);
    std::vector<std::vector<MapPoint> > storeN(NK);
    std::vector<KeyFrame *> vpN(NK);
    std::vector<cv::Mat> vF12(NK);
    cur.Ow = cv::Mat(3, 1, CV_32F); cur.Ow.at<float>(0) = 0.5372f; cur.Ow.at<float>(1) = 0.01f; cur.Ow.at<float>(2) = 0.02f;
    std::vector<float> f12all;
    for (int k = 0; k < NK; k++) {
        derived_keyframe(nb[k], storeN[k], k + 1, (k & 1) ? kl : kr, (k & 1) ? dl : dr, (k & 1) ? ur : std::vector<float>(), sf, s2);
        vpN[k] = &nb[k];
        nb[k].Rcw = cv::Mat(3, 3, CV_32F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) nb[k].Rcw.at<float>(r, c) = r == c ? 1.f : 0.f;
        nb[k].tcw = cv::Mat(3, 1, CV_32F);
        nb[k].tcw.at<float>(0) = 0.05f * (float)k; nb[k].tcw.at<float>(1) = 0.f; nb[k].tcw.at<float>(2) = 0.25f + 0.01f * (float)k;
        vF12[k] = cv::Mat(3, 3, CV_32F);
        const float g = 1.f / (718.856f + 3.f * (float)k);       // a family of near-rectified epipolar geometries
        const float f12[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, -g, 0.f, g, 0.0001f * (float)k };
        for (int i = 0; i < 9; i++) { vF12[k].at<float>(i / 3, i % 3) = f12[i]; f12all.push_back(f12[i]); }
        char pre[32];
        snprintf(pre, sizeof pre, "nb%d_", k);
        dump_keyframe(fo, pre, nb[k]);
        std::vector<float> pose(3);
        for (int i = 0; i < 3; i++) pose[i] = nb[k].tcw.at<float>(i);
        snprintf(pre, sizeof pre, "nb%d_tcw", k); dump_f(fo, pre, pose);
    }
    dump_keyframe(fo, "cur_", cur);
    dump_f(fo, "F12s", f12all);
    std::vector<float> ow(3);
    for (int i = 0; i < 3; i++) ow[i] = cur.Ow.at<float>(i);
    dump_f(fo, "cur_Ow", ow);
    // the frame of the relocalisation loop: the left eye as extracted
    Frame F;
    F.N = (int)kl.size(); F.mvKeys = kl; F.mvKeysUn = kl; F.mDescriptors = dl;
    feature_vector(F.mDescriptors, F.mFeatVec);

    orbx_adapter::KeyFrameCache &cache = orbx_adapter::KeyFrameCache::instance();
    // ---- single-pair adaptors BEFORE anything is resident: host-pointer path (reference for "cache on == cache off")
    std::vector<std::pair<size_t, size_t> > pairs_h;
    ORBmatcher tri(0.6f, false), bow(0.75f, true);
    tri.SearchForTriangulation(&cur, vpN[3], vF12[3], pairs_h, false);
    std::vector<MapPoint *> vp12_h;
    const int n12_h = bow.SearchByBoW(&cur, vpN[4], vp12_h);
    if (cache.size() != 0) { fprintf(stderr, "cache not empty\n"); return 4; }

    // ---- LocalMapping::CreateNewMapPoints: all neighbours in one call
    std::vector<std::vector<std::pair<size_t, size_t> > > vvPairs;
    const int total_tri = orbx_adapter::SearchForTriangulationBatch(&cur, vpN, vF12, vvPairs, false);
    for (int k = 0; k < NK; k++) {
        std::vector<long long> flat;
        for (size_t i = 0; i < vvPairs[k].size(); i++) { flat.push_back((long long)vvPairs[k][i].first); flat.push_back((long long)vvPairs[k][i].second); }
        char nm[32]; snprintf(nm, sizeof nm, "tri%d", k); dump_i(fo, nm, flat);
    }
    std::vector<std::vector<std::pair<size_t, size_t> > > vvPairsS;
    orbx_adapter::SearchForTriangulationBatch(&cur, vpN, vF12, vvPairsS, true);        // bOnlyStereo
    for (int k = 0; k < NK; k += 7) {
        std::vector<long long> flat;
        for (size_t i = 0; i < vvPairsS[k].size(); i++) { flat.push_back((long long)vvPairsS[k][i].first); flat.push_back((long long)vvPairsS[k][i].second); }
        char nm[32]; snprintf(nm, sizeof nm, "triStereo%d", k); dump_i(fo, nm, flat);
    }
    // ---- LoopClosing::ComputeSim3: all candidates in one call
    std::vector<std::vector<MapPoint *> > vv12; std::vector<int> vn12;
    orbx_adapter::SearchByBoWBatch(&cur, vpN, vv12, vn12);
    for (int k = 0; k < NK; k++) { char nm[32]; snprintf(nm, sizeof nm, "bow12_%d", k); dump_i(fo, nm, held_in(vv12[k], storeN[k])); }
    dump_i(fo, "bow12_n", vn12);
    // ---- Tracking::Relocalization: all candidates against the frame in one call
    std::vector<std::vector<MapPoint *> > vvF; std::vector<int> vnF;
    orbx_adapter::SearchByBoWBatch(vpN, F, vvF, vnF);
    for (int k = 0; k < NK; k++) { char nm[32]; snprintf(nm, sizeof nm, "bowF_%d", k); dump_i(fo, nm, held_in(vvF[k], storeN[k])); }
    dump_i(fo, "bowF_n", vnF);
    dump_i(fo, "cache_size", std::vector<int>(1, (int)cache.size()));

    // ---- the single-pair adaptors again, now on resident keyframes: the same answers as the host-pointer path and as the batch
    std::vector<std::pair<size_t, size_t> > pairs_r;
    tri.SearchForTriangulation(&cur, vpN[3], vF12[3], pairs_r, false);
    std::vector<MapPoint *> vp12_r;
    const int n12_r = bow.SearchByBoW(&cur, vpN[4], vp12_r);
    const int same = (pairs_r == pairs_h && pairs_r == vvPairs[3] && vp12_r == vp12_h && vp12_r == vv12[4] && n12_r == n12_h && n12_r == vn12[4]) ? 1 : 0;
    dump_i(fo, "single_equals_batch", std::vector<int>(1, same));
    cache.drop(vpN[4]);
    std::vector<MapPoint *> vp12_d;
    const int n12_d = bow.SearchByBoW(&cur, vpN[4], vp12_d);                            // dropped: host-pointer path again
    dump_i(fo, "after_drop_equal", std::vector<int>(1, (vp12_d == vp12_h && n12_d == n12_h && cache.size() == (size_t)NK) ? 1 : 0));
    cache.get(vpN[4]);

    // ---- adaptor-inclusive timing: what ORB-SLAM2's threads would see (flags under the keyframe mutexes, epipoles, call, map-back)
    typedef std::chrono::steady_clock clk;
    std::vector<long long> t_tri_b, t_bow_b, t_rel_b, t_tri_1, t_bow_1, t_tri_h, t_bow_h, t_bowF_h;
    for (int r = 0; r < reps; r++) {
        clk::time_point t0 = clk::now();
        orbx_adapter::SearchForTriangulationBatch(&cur, vpN, vF12, vvPairs, false);
        clk::time_point t1 = clk::now();
        orbx_adapter::SearchByBoWBatch(&cur, vpN, vv12, vn12);
        clk::time_point t2 = clk::now();
        orbx_adapter::SearchByBoWBatch(vpN, F, vvF, vnF);
        clk::time_point t3 = clk::now();
        tri.SearchForTriangulation(&cur, vpN[r % NK], vF12[r % NK], pairs_r, false);      // resident single pair
        clk::time_point t4 = clk::now();
        bow.SearchByBoW(&cur, vpN[r % NK], vp12_r);
        clk::time_point t5 = clk::now();
        t_tri_b.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
        t_bow_b.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count());
        t_rel_b.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count());
        t_tri_1.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t4 - t3).count());
        t_bow_1.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t5 - t4).count());
    }
    cache.clear();                                                                       // host-pointer single pairs (round-3 adaptor path)
    for (int r = 0; r < reps; r++) {
        clk::time_point t0 = clk::now();
        tri.SearchForTriangulation(&cur, vpN[r % NK], vF12[r % NK], pairs_r, false);
        clk::time_point t1 = clk::now();
        bow.SearchByBoW(&cur, vpN[r % NK], vp12_r);
        clk::time_point t2 = clk::now();
        bow.SearchByBoW(vpN[r % NK], F, vp12_r);
        clk::time_point t3 = clk::now();
        t_tri_h.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
        t_bow_h.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count());
        t_bowF_h.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count());
    }
    std::vector<long long> tm;
    tm.push_back(median_ns(t_tri_b)); tm.push_back(median_ns(t_bow_b)); tm.push_back(median_ns(t_rel_b)); tm.push_back(median_ns(t_tri_1)); tm.push_back(median_ns(t_bow_1));
    tm.push_back(median_ns(t_tri_h)); tm.push_back(median_ns(t_bow_h)); tm.push_back(median_ns(t_bowF_h));
    dump_i(fo, "time_ns", tm);       // [tri batch of 20, bow(KF,KF) batch of 20, bow(KFs,F) batch of 20, tri resident single, bow resident single, tri host, bow(KF,KF) host, bow(KF,F) host]
    dump_i(fo, "time_reps", std::vector<int>(1, reps));
    dump_i(fo, "features", std::vector<int>(1, cur.N));
    fclose(fo);
    printf("adaptor batch ok: %d keyframes x %d features, %d triangulation pairs; us per pair: tri batch %.2f, bow12 batch %.2f, bowF batch %.2f; single resident tri %.1f bow %.1f; "
           "single host tri %.1f bow12 %.1f bowF %.1f\n", NK, cur.N, total_tri, tm[0] / 1e3 / NK, tm[1] / 1e3 / NK, tm[2] / 1e3 / NK, tm[3] / 1e3, tm[4] / 1e3, tm[5] / 1e3, tm[6] / 1e3, tm[7] / 1e3);
    return 0;
}


int main(int argc, char **argv)
{
    try {
        if (argc >= 2 && !strcmp(argv[1], "csr")) return mode_csr();
        if (argc >= 4 && !strcmp(argv[1], "run")) return mode_run(argv[2], argv[3]);
        if (argc >= 4 && !strcmp(argv[1], "batch")) return mode_batch(argv[2], argv[3], argc >= 5 ? atoi(argv[4]) : 50);
#ifdef ORBX_ADAPTER_CAPTURE
        if (argc >= 5 && !strcmp(argv[1], "track")) return mode_track(argv[2], argv[3], argv[4]);
        if (argc >= 4 && !strcmp(argv[1], "map")) return mode_map(argv[2], argv[3]);
#endif
    } catch (const std::exception &e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    fprintf(stderr, "usage: adapter_driver csr | run in.bin out.txt\n");
    return 1;
}
