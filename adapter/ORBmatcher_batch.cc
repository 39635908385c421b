// adapter/ORBmatcher_batch.cc -- resident keyframes (orbx_adapter::KeyFrameCache) and the three loops of the reference that run a
// vocabulary-guided search per neighbour / candidate keyframe, each as ONE call into liborbx (adapter/orbx_batch.h).
// LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:241-309), LoopClosing::ComputeSim3 (src/LoopClosing.cc:293-323),
// Tracking::Relocalization (src/Tracking.cc:1661-1682).
#include "orbx_batch.h"

#include <stdexcept>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "orbx_adapter.h"

using namespace std;
using ORB_SLAM2::Frame;
using ORB_SLAM2::KeyFrame;
using ORB_SLAM2::MapPoint;

namespace orbx_adapter
{

// ---------------------------------------------------------------- resident keyframes

KeyFrameCache &KeyFrameCache::instance()
{
    static KeyFrameCache c;
    return c;
}

KeyFrameCache::~KeyFrameCache() { clear(); }

// the immutable matching data of a keyframe behind an orbx_featset: descriptors, FeatureVector, undistorted keypoints, mvuRight
static void immutable_side(KeyFrame *pKF, Side &s)
{
    s.csr = flatten(pKF->mFeatVec);
    const int n = pKF->N;
    s.angle.resize(n); s.x.resize(n); s.y.resize(n); s.octave.resize(n); s.u_right.resize(n);
    for (int i = 0; i < n; i++) {
        const cv::KeyPoint &kp = pKF->mvKeysUn[i];
        s.angle[i] = kp.angle; s.x[i] = kp.pt.x; s.y[i] = kp.pt.y; s.octave[i] = kp.octave;
        s.u_right[i] = i < (int)pKF->mvuRight.size() ? pKF->mvuRight[i] : -1.0f;
    }
    s.bind(dense_descriptors(pKF->mDescriptors, n), n);
}

const orbx_kf *KeyFrameCache::get(KeyFrame *pKF)
{
    {
        unique_lock<mutex> lock(mMutex);
        map<const KeyFrame *, orbx_kf *>::iterator it = mKFs.find(pKF);
        if (it != mKFs.end()) return it->second;
    }
    Side s;                                   // built outside the lock: a 1000-feature keyframe is a 60 KB upload
    immutable_side(pKF, s);
    orbx_kf *k = NULL;
    if (orbx_kf_create(Device(), &s.fs, &k) != ORBX_OK) throw std::runtime_error(orbx_last_error());
    unique_lock<mutex> lock(mMutex);
    pair<map<const KeyFrame *, orbx_kf *>::iterator, bool> ins = mKFs.insert(make_pair((const KeyFrame *)pKF, k));
    if (!ins.second) orbx_kf_destroy(k);      // another thread was faster
    return ins.first->second;
}

const orbx_kf *KeyFrameCache::find(const KeyFrame *pKF)
{
    unique_lock<mutex> lock(mMutex);
    map<const KeyFrame *, orbx_kf *>::iterator it = mKFs.find(pKF);
    return it == mKFs.end() ? NULL : it->second;
}

void KeyFrameCache::drop(const KeyFrame *pKF)
{
    orbx_kf *k = NULL;
    {
        unique_lock<mutex> lock(mMutex);
        map<const KeyFrame *, orbx_kf *>::iterator it = mKFs.find(pKF);
        if (it == mKFs.end()) return;
        k = it->second;
        mKFs.erase(it);
    }
    orbx_kf_destroy(k);
}

void KeyFrameCache::clear()
{
    unique_lock<mutex> lock(mMutex);
    for (map<const KeyFrame *, orbx_kf *>::iterator it = mKFs.begin(); it != mKFs.end(); ++it) orbx_kf_destroy(it->second);
    mKFs.clear();
}

size_t KeyFrameCache::size()
{
    unique_lock<mutex> lock(mMutex);
    return mKFs.size();
}

// ---------------------------------------------------------------- per-call inputs

// flag[i] = feature i holds a MapPoint that is not bad (src/ORBmatcher.cc:204-210, :605-626); the copy of the keyframe's MapPoint
// vector the reference takes under the keyframe's mutex (:173, :572-576) is returned for mapping the indices back
void GoodPointFlags(KeyFrame *pKF, vector<MapPoint *> &vpMapPoints, vector<uint8_t> &flag)
{
    vpMapPoints = pKF->GetMapPointMatches();
    flag.resize(vpMapPoints.size());
    for (size_t i = 0; i < vpMapPoints.size(); i++)
        flag[i] = (vpMapPoints[i] && !vpMapPoints[i]->isBad()) ? 1 : 0;
}

// flag[i] = feature i already has a MapPoint (src/ORBmatcher.cc:750, :773: such features are skipped)
void HasPointFlags(KeyFrame *pKF, vector<uint8_t> &flag)
{
    flag.resize(pKF->N);
    for (int i = 0; i < pKF->N; i++)
        flag[i] = pKF->GetMapPoint(i) ? 1 : 0;
}

// src/ORBmatcher.cc:712-718, the reference's own cv::Mat arithmetic on the host
void Epipole(KeyFrame *pKF1, KeyFrame *pKF2, float &ex, float &ey)
{
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
}

// ---------------------------------------------------------------- the three loops

int SearchForTriangulationBatch(KeyFrame *pKF1, const vector<KeyFrame *> &vpKF2, const vector<cv::Mat> &vF12,
                                vector<vector<pair<size_t, size_t> > > &vvMatchedPairs, bool bOnlyStereo, float nnratio, bool checkOri)
{
    (void)nnratio;                            // SearchForTriangulation never reads mfNNratio (src/ORBmatcher.cc:704-871)
    const size_t n2 = vpKF2.size();
    vvMatchedPairs.assign(n2, vector<pair<size_t, size_t> >());
    if (n2 == 0) return 0;
    if (vF12.size() != n2) throw std::runtime_error("SearchForTriangulationBatch: one F12 per neighbour keyframe");
    KeyFrameCache &cache = KeyFrameCache::instance();
    const orbx_kf *k1 = cache.get(pKF1);
    vector<const orbx_kf *> k2s(n2);
    vector<vector<uint8_t> > flags2(n2);
    vector<const uint8_t *> fp2(n2);
    vector<uint8_t> flag1;
    HasPointFlags(pKF1, flag1);
    vector<float> f12(9 * n2), ep(2 * n2);
    for (size_t i = 0; i < n2; i++) {
        k2s[i] = cache.get(vpKF2[i]);
        HasPointFlags(vpKF2[i], flags2[i]);
        fp2[i] = flags2[i].empty() ? NULL : &flags2[i][0];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) f12[9 * i + 3 * r + c] = vF12[i].at<float>(r, c);
        Epipole(pKF1, vpKF2[i], ep[2 * i], ep[2 * i + 1]);
    }
    const int cap = pKF1->N > 0 ? pKF1->N : 1;
    vector<int32_t> pairs(2 * (size_t)cap * n2);
    vector<int> npairs(n2, 0);
    KeyFrame *any2 = vpKF2[0];                // all keyframes of a map share one extractor's level tables (src/KeyFrame.cc:37-39)
    if (orbx_kf_search_for_triangulation(k1, flag1.empty() ? NULL : &flag1[0], &k2s[0], &fp2[0], (int)n2, &f12[0], &ep[0], &any2->mvScaleFactors[0],
                                         &any2->mvLevelSigma2[0], (int)any2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, checkOri ? 1 : 0, &pairs[0], cap,
                                         &npairs[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    int total = 0;
    for (size_t i = 0; i < n2; i++) {
        const int32_t *p = &pairs[2 * (size_t)cap * i];
        vvMatchedPairs[i].reserve(npairs[i]);                                       // :858-868
        for (int j = 0; j < npairs[i]; j++) vvMatchedPairs[i].push_back(make_pair((size_t)p[2 * j], (size_t)p[2 * j + 1]));
        total += npairs[i];
    }
    return total;
}

void SearchByBoWBatch(KeyFrame *pKF1, const vector<KeyFrame *> &vpKF2, vector<vector<MapPoint *> > &vvpMatches12, vector<int> &vnMatches,
                      float nnratio, bool checkOri)
{
    const size_t n2 = vpKF2.size();
    vvpMatches12.assign(n2, vector<MapPoint *>());
    vnMatches.assign(n2, 0);
    if (n2 == 0) return;
    KeyFrameCache &cache = KeyFrameCache::instance();
    const orbx_kf *k1 = cache.get(pKF1);
    vector<MapPoint *> vp1;
    vector<uint8_t> flag1;
    GoodPointFlags(pKF1, vp1, flag1);
    vector<const orbx_kf *> k2s(n2);
    vector<vector<MapPoint *> > vp2(n2);
    vector<vector<uint8_t> > flags2(n2);
    vector<const uint8_t *> fp2(n2);
    static const uint8_t none = 0;
    for (size_t i = 0; i < n2; i++) {
        k2s[i] = cache.get(vpKF2[i]);
        GoodPointFlags(vpKF2[i], vp2[i], flags2[i]);
        fp2[i] = flags2[i].empty() ? &none : &flags2[i][0];
    }
    const size_t n1 = (size_t)(pKF1->N > 0 ? pKF1->N : 1);
    vector<int32_t> match12(n1 * n2);
    if (orbx_kf_search_by_bow_kf_kf(k1, flag1.empty() ? &none : &flag1[0], &k2s[0], &fp2[0], (int)n2, nnratio, checkOri ? 1 : 0, &match12[0], &vnMatches[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (size_t i = 0; i < n2; i++) {
        vvpMatches12[i] = vector<MapPoint *>(vp1.size(), static_cast<MapPoint *>(NULL));   // :580
        const int32_t *m = &match12[n1 * i];
        for (int j = 0; j < pKF1->N; j++)
            if (m[j] >= 0) vvpMatches12[i][j] = vp2[i][m[j]];
    }
}

void SearchByBoWBatch(const vector<KeyFrame *> &vpKFs, Frame &F, vector<vector<MapPoint *> > &vvpMapPointMatches, vector<int> &vnMatches,
                      float nnratio, bool checkOri)
{
    const size_t nkf = vpKFs.size();
    vvpMapPointMatches.assign(nkf, vector<MapPoint *>());
    vnMatches.assign(nkf, 0);
    if (nkf == 0) return;
    KeyFrameCache &cache = KeyFrameCache::instance();
    // the frame side: descriptors, FeatureVector and the angles of F.mvKeys (src/ORBmatcher.cc:253 compares pKF->mvKeysUn with F.mvKeys)
    Side f;
    f.csr = flatten(F.mFeatVec);
    f.angle.resize(F.N);
    for (int i = 0; i < F.N; i++) f.angle[i] = F.mvKeys[i].angle;
    f.flag.assign(F.N, 0);
    f.bind(dense_descriptors(F.mDescriptors, F.N), F.N);
    vector<const orbx_kf *> ks(nkf);
    vector<vector<MapPoint *> > vp(nkf);
    vector<vector<uint8_t> > flags(nkf);
    vector<const uint8_t *> fp(nkf);
    static const uint8_t none = 0;
    const size_t nf = (size_t)(F.N > 0 ? F.N : 1);
    vector<int32_t> match(nf * nkf);
    for (size_t i = 0; i < nkf; i++) {
        ks[i] = cache.get(vpKFs[i]);
        GoodPointFlags(vpKFs[i], vp[i], flags[i]);
        fp[i] = flags[i].empty() ? &none : &flags[i][0];
    }
    if (orbx_kf_search_by_bow_kfs_f(&ks[0], &fp[0], (int)nkf, &f.fs, nnratio, checkOri ? 1 : 0, &match[0], &vnMatches[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (size_t i = 0; i < nkf; i++) {
        vvpMapPointMatches[i] = vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));     // :175
        const int32_t *m = &match[nf * i];
        for (int j = 0; j < F.N; j++)
            if (m[j] >= 0) vvpMapPointMatches[i][j] = vp[i][m[j]];
    }
}

} // namespace orbx_adapter
