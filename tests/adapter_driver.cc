// tests/adapter_driver.cc -- drives the compiled C++ adaptor (adapter/*.cc, the code a maintainer drops into ORB-SLAM2)
// the way the reference's Tracking / LocalMapping / LoopClosing threads drive ORBextractor, Frame::ComputeStereoMatches
// and ORBmatcher.  Built by tests/test_adapter.py with g++ against tests/cvstub (OpenCV is not installed) and linked
// with liborbx.so.
//
//   adapter_driver csr                 FeatureVector -> flatten() -> orbx_featset round trip (no GPU needed)
//   adapter_driver run in.bin out.txt  stereo pair through the adaptor classes; every result is written as
//                                      "name n v0 v1 ..." lines (floats as their uint32 bit patterns)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <utility>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "orbx_adapter.h"

using namespace ORB_SLAM2;

float Frame::fx = 718.856f;

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
static void feature_vector(const cv::Mat &desc, DBoW2::FeatureVector &fv)
{
    for (int i = 0; i < desc.rows; i++)
        if (desc.at<uchar>(i, 1) % 8 != 0) fv.addFeature(100 + (desc.at<uchar>(i, 0) & 15), (unsigned)i);
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

int main(int argc, char **argv)
{
    try {
        if (argc >= 2 && !strcmp(argv[1], "csr")) return mode_csr();
        if (argc >= 4 && !strcmp(argv[1], "run")) return mode_run(argv[2], argv[3]);
    } catch (const std::exception &e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    fprintf(stderr, "usage: adapter_driver csr | run in.bin out.txt\n");
    return 1;
}
