// adapter/ORBmatcher_bow.cc -- the three vocabulary-guided searches of ORB_SLAM2::ORBmatcher over liborbx: replaces
// SearchByBoW(KeyFrame*, Frame&) (reference src/ORBmatcher.cc:171-303), SearchByBoW(KeyFrame*, KeyFrame*) (:568-702) and
// SearchForTriangulation (:704-871, with CheckDistEpipolarLine :147-164 and ComputeThreeMaxima :1687-1728 running on the
// device).  include/ORBmatcher.h stays the reference's; the other methods of src/ORBmatcher.cc stay compiled as they are
// (or move to the orbx_search_by_projection_* entry points).
#include "ORBmatcher.h"

#include <stdexcept>

#include "orbx_adapter.h"
#include "orbx_batch.h"

using namespace std;

namespace ORB_SLAM2
{

using orbx_adapter::Side;

static void keys_to_angles(const vector<cv::KeyPoint> &keys, vector<float> &angle)
{
    angle.resize(keys.size());
    for (size_t i = 0; i < keys.size(); i++)
        angle[i] = keys[i].angle;
}

// flag[i] = feature i holds a MapPoint that is not bad (:204-210, :605-626)
static void good_points(const vector<MapPoint *> &vp, vector<uint8_t> &flag)
{
    flag.resize(vp.size());
    for (size_t i = 0; i < vp.size(); i++)
        flag[i] = (vp[i] && !vp[i]->isBad()) ? 1 : 0;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint *> &vpMapPointMatches)
{
    const vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();          // copy under the keyframe's mutex, :173
    Side kf, f;
    kf.csr = orbx_adapter::flatten(pKF->mFeatVec);
    f.csr = orbx_adapter::flatten(F.mFeatVec);
    good_points(vpMapPointsKF, kf.flag);
    f.flag.assign(F.N, 0);
    keys_to_angles(pKF->mvKeysUn, kf.angle);                                       // :253: pKF->mvKeysUn vs F.mvKeys
    keys_to_angles(F.mvKeys, f.angle);
    kf.bind(orbx_adapter::dense_descriptors(pKF->mDescriptors, pKF->N), pKF->N);
    f.bind(orbx_adapter::dense_descriptors(F.mDescriptors, F.N), F.N);
    vector<int32_t> match(F.N > 0 ? F.N : 1);
    int nmatches = 0;
    if (orbx_search_by_bow_kf_f(orbx_adapter::Device(), &kf.fs, &f.fs, mfNNratio, mbCheckOrientation ? 1 : 0, &match[0], &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    vpMapPointMatches = vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));   // :175
    for (int i = 0; i < F.N; i++)
        if (match[i] >= 0)
            vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12)
{
    // both keyframes resident in HBM (orbx_adapter::KeyFrameCache: a batched call has seen them, or the KeyFrame constructor registered
    // them): only the map-point flags, the node intersection and the result row cross PCIe
    orbx_adapter::KeyFrameCache &cache = orbx_adapter::KeyFrameCache::instance();
    if (const orbx_kf *r1 = cache.find(pKF1))
        if (const orbx_kf *r2 = cache.find(pKF2)) {
            vector<MapPoint *> vp1, vp2;
            vector<uint8_t> fl1, fl2;
            orbx_adapter::GoodPointFlags(pKF1, vp1, fl1);
            orbx_adapter::GoodPointFlags(pKF2, vp2, fl2);
            static const uint8_t none = 0;
            const uint8_t *f2 = fl2.empty() ? &none : &fl2[0];
            vector<int32_t> m12(pKF1->N > 0 ? pKF1->N : 1);
            int n = 0;
            if (orbx_kf_search_by_bow_kf_kf(r1, fl1.empty() ? &none : &fl1[0], &r2, &f2, 1, mfNNratio, mbCheckOrientation ? 1 : 0, &m12[0], &n) != ORBX_OK)
                throw std::runtime_error(orbx_last_error());
            vpMatches12 = vector<MapPoint *>(vp1.size(), static_cast<MapPoint *>(NULL));
            for (int i = 0; i < pKF1->N; i++)
                if (m12[i] >= 0) vpMatches12[i] = vp2[m12[i]];
            return n;
        }
    const vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    Side k1, k2;
    k1.csr = orbx_adapter::flatten(pKF1->mFeatVec);
    k2.csr = orbx_adapter::flatten(pKF2->mFeatVec);
    good_points(vpMapPoints1, k1.flag);
    good_points(vpMapPoints2, k2.flag);
    keys_to_angles(pKF1->mvKeysUn, k1.angle);                                      // :654
    keys_to_angles(pKF2->mvKeysUn, k2.angle);
    k1.bind(orbx_adapter::dense_descriptors(pKF1->mDescriptors, pKF1->N), pKF1->N);
    k2.bind(orbx_adapter::dense_descriptors(pKF2->mDescriptors, pKF2->N), pKF2->N);
    vector<int32_t> match12(pKF1->N > 0 ? pKF1->N : 1);
    int nmatches = 0;
    if (orbx_search_by_bow_kf_kf(orbx_adapter::Device(), &k1.fs, &k2.fs, mfNNratio, mbCheckOrientation ? 1 : 0, &match12[0], &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    vpMatches12 = vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));   // :580
    for (int i = 0; i < pKF1->N; i++)
        if (match12[i] >= 0)
            vpMatches12[i] = vpMapPoints2[match12[i]];
    return nmatches;
}

static void triangulation_side(KeyFrame *pKF, Side &s)
{
    s.csr = orbx_adapter::flatten(pKF->mFeatVec);
    const int n = pKF->N;
    s.flag.resize(n); s.angle.resize(n); s.x.resize(n); s.y.resize(n); s.octave.resize(n); s.u_right.resize(n);
    for (int i = 0; i < n; i++) {
        const cv::KeyPoint &kp = pKF->mvKeysUn[i];
        s.flag[i] = pKF->GetMapPoint(i) ? 1 : 0;                                    // :750, :773: features that already have a point are skipped
        s.angle[i] = kp.angle;
        s.x[i] = kp.pt.x;
        s.y[i] = kp.pt.y;
        s.octave[i] = kp.octave;
        s.u_right[i] = pKF->mvuRight[i];
    }
    s.bind(orbx_adapter::dense_descriptors(pKF->mDescriptors, n), n);
}

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo)
{
    // epipole of camera 1 in image 2, exactly the reference's lines :712-718 (real OpenCV arithmetic on the host)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    float f12[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            f12[3 * r + c] = F12.at<float>(r, c);
    {   // both keyframes resident (orbx_adapter::KeyFrameCache): see SearchByBoW(KeyFrame*, KeyFrame*)
        orbx_adapter::KeyFrameCache &cache = orbx_adapter::KeyFrameCache::instance();
        const orbx_kf *r1 = cache.find(pKF1), *r2 = r1 ? cache.find(pKF2) : NULL;
        if (r1 && r2) {
            vector<uint8_t> fl1, fl2;
            orbx_adapter::HasPointFlags(pKF1, fl1);
            orbx_adapter::HasPointFlags(pKF2, fl2);
            const uint8_t *f2 = fl2.empty() ? NULL : &fl2[0];
            const float ep[2] = { ex, ey };
            const int cap = pKF1->N > 0 ? pKF1->N : 1;
            vector<int32_t> pairs(2 * (size_t)cap);
            int npairs = 0;
            if (orbx_kf_search_for_triangulation(r1, fl1.empty() ? NULL : &fl1[0], &r2, &f2, 1, f12, ep, &pKF2->mvScaleFactors[0], &pKF2->mvLevelSigma2[0],
                                                 (int)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, &pairs[0], cap, &npairs) != ORBX_OK)
                throw std::runtime_error(orbx_last_error());
            vMatchedPairs.clear();
            vMatchedPairs.reserve(npairs);
            for (int i = 0; i < npairs; i++)
                vMatchedPairs.push_back(make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
            return npairs;
        }
    }
    Side k1, k2;
    triangulation_side(pKF1, k1);
    triangulation_side(pKF2, k2);
    const int cap = pKF1->N > 0 ? pKF1->N : 1;
    vector<int32_t> pairs(2 * (size_t)cap);
    int npairs = 0;
    if (orbx_search_for_triangulation(orbx_adapter::Device(), &k1.fs, &k2.fs, f12, ex, ey, &pKF2->mvScaleFactors[0], &pKF2->mvLevelSigma2[0],
                                      (int)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, &pairs[0], cap,
                                      &npairs) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    vMatchedPairs.clear();                                                          // :858-868
    vMatchedPairs.reserve(npairs);
    for (int i = 0; i < npairs; i++)
        vMatchedPairs.push_back(make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
    return npairs;
}

} // namespace ORB_SLAM2
