// adapter/ORBmatcher_fuse.cc -- the projection searches of relocalisation, local mapping and loop closing over liborbx: replaces
//   SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)    reference src/ORBmatcher.cc:1555-1685
//   SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th)                 :305-415
//   Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th)                            :873-1038
//   Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint)                           :1040-1164
//   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)                                 :1166-1394
// Same split as adapter/ORBmatcher_proj.cc: the host reads the map under the reference's accessors and projects every point with
// the reference's own cv::Mat expressions (poses, depth, distance range, viewing angle, PredictScale); the window search, the level
// gates, the chi-square gates of Fuse, the Hamming distances and the claiming order run on the device.  What changes the map --
// Replace / AddObservation / AddMapPoint / vpReplacePoint -- is applied here, in the reference's order, from the device's answer.
#include "ORBmatcher.h"

#include <math.h>
#include <set>
#include <stdexcept>

#include "orbx_adapter.h"

using namespace std;

namespace ORB_SLAM2
{

using orbx_adapter::FrameSide;
using orbx_adapter::PointSide;

// a KeyFrame's undistorted keypoints behind an orbx_frame_feats; bounds = the keyframe's image bounds (IsInImage runs on the device)
static void keyframe_side(KeyFrame *pKF, FrameSide &s)
{
    const size_t n = (size_t)pKF->N;
    s.x.resize(n); s.y.resize(n); s.angle.resize(n); s.octave.resize(n); s.occupied.assign(n, 0);
    for (size_t i = 0; i < n; i++) {
        const cv::KeyPoint &kp = pKF->mvKeysUn[i];
        s.x[i] = kp.pt.x; s.y[i] = kp.pt.y; s.angle[i] = kp.angle; s.octave[i] = kp.octave;
    }
    memset(&s.ff, 0, sizeof s.ff);
    s.ff.n = pKF->N;
    if (n) {
        s.ff.x = &s.x[0]; s.ff.y = &s.y[0]; s.ff.octave = &s.octave[0]; s.ff.angle = &s.angle[0];
        s.ff.u_right = &pKF->mvuRight[0]; s.ff.desc = orbx_adapter::dense_descriptors(pKF->mDescriptors, pKF->N); s.ff.occupied = &s.occupied[0];
    }
    s.ff.min_x = (float)pKF->mnMinX; s.ff.min_y = (float)pKF->mnMinY; s.ff.max_x = (float)pKF->mnMaxX; s.ff.max_y = (float)pKF->mnMaxY;
}

static void frame_side_any_point(const Frame &F, FrameSide &s)      // occupied[i] = the feature holds a point (:1617-1618)
{
    const size_t n = (size_t)F.N;
    s.x.resize(n); s.y.resize(n); s.angle.resize(n); s.octave.resize(n); s.occupied.assign(n, 0);
    for (size_t i = 0; i < n; i++) {
        const cv::KeyPoint &kp = F.mvKeysUn[i];
        s.x[i] = kp.pt.x; s.y[i] = kp.pt.y; s.angle[i] = kp.angle; s.octave[i] = kp.octave;
        s.occupied[i] = F.mvpMapPoints[i] ? 1 : 0;
    }
    memset(&s.ff, 0, sizeof s.ff);
    s.ff.n = F.N;
    if (n) {
        s.ff.x = &s.x[0]; s.ff.y = &s.y[0]; s.ff.octave = &s.octave[0]; s.ff.angle = &s.angle[0];
        s.ff.u_right = &F.mvuRight[0]; s.ff.desc = orbx_adapter::dense_descriptors(F.mDescriptors, F.N); s.ff.occupied = &s.occupied[0];
    }
    s.ff.min_x = Frame::mnMinX; s.ff.min_y = Frame::mnMinY; s.ff.max_x = Frame::mnMaxX; s.ff.max_y = Frame::mnMaxY;
}

static void put_desc(MapPoint *pMP, uint8_t *dst)
{
    const cv::Mat d = pMP->GetDescriptor();
    if (!d.empty())
        memcpy(dst, d.data, 32);
}

// world point -> pixel of a pinhole camera at (Rcw, tcw); false if it lies behind the camera.  The reference's statements, in its order.
static bool project(const cv::Mat &Rcw, const cv::Mat &tcw, const cv::Mat &p3Dw, float fx, float fy, float cx, float cy, float &u, float &v, float &invz)
{
    const cv::Mat p3Dc = Rcw * p3Dw + tcw;
    if (p3Dc.at<float>(2) < 0.0f)
        return false;
    invz = 1.0 / p3Dc.at<float>(2);
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    u = fx * x + cx;
    v = fy * y + cy;
    return true;
}

// distance range + viewing angle of a point seen from camera centre Ow (:359-372, :917-945); dist3D out
static bool in_range_and_facing(MapPoint *pMP, const cv::Mat &p3Dw, const cv::Mat &Ow, float &dist3D)
{
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const cv::Mat PO = p3Dw - Ow;
    dist3D = (float)cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance)
        return false;
    const cv::Mat Pn = pMP->GetNormal();
    return !(PO.dot(Pn) < 0.5 * dist3D);
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    FrameSide cur;
    frame_side_any_point(CurrentFrame, cur);
    PointSide pts(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP))       // :1573-1577
            continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        pts.u[i] = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;   // the image-bounds test (:1591-1594) runs on the device
        pts.v[i] = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        const cv::Mat PO = x3Dw - Ow;
        const float dist3D = (float)cv::norm(PO);
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance())
            continue;
        pts.level[i] = pMP->PredictScale(dist3D, &CurrentFrame);
        pts.angle[i] = pKF->mvKeysUn[i].angle;
        pts.valid[i] = 1;
        put_desc(pMP, &pts.desc[32 * i]);
    }
    vector<int32_t> match((size_t)(CurrentFrame.N > 0 ? CurrentFrame.N : 1));
    int nmatches = 0;
    ORBX_CAPTURE(cur.ff, &pts.pp);
    if (orbx_search_by_projection_keyframe(orbx_adapter::Device(), &cur.ff, &pts.pp, &CurrentFrame.mvScaleFactors[0], (int)CurrentFrame.mvScaleFactors.size(), th, ORBdist,
                                           mbCheckOrientation ? 3 : 0, &match[0], &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int f = 0; f < CurrentFrame.N; f++) {
        if (match[f] >= 0)
            CurrentFrame.mvpMapPoints[f] = vpMPs[match[f]];          // :1626
        else if (match[f] == -2)
            CurrentFrame.mvpMapPoints[f] = NULL;                     // matched, then cleared by the rotation filter (:1672)
    }
    return nmatches;
}

// Sim3 -> (Rcw, tcw, Ow) exactly as :312-316 / :1047-1051
static void decompose_sim3(const cv::Mat &Scw, cv::Mat &Rcw, cv::Mat &tcw, cv::Mat &Ow)
{
    const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    Rcw = sRcw / scw;
    tcw = Scw.rowRange(0, 3).col(3) / scw;
    Ow = -Rcw.t() * tcw;
}

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, vector<MapPoint *> &vpMatched, int th)
{
    cv::Mat Rcw, tcw, Ow;
    decompose_sim3(Scw, Rcw, tcw, Ow);
    set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    FrameSide kf;
    keyframe_side(pKF, kf);
    for (size_t i = 0; i < kf.occupied.size(); i++)
        kf.occupied[i] = vpMatched[i] ? 1 : 0;                      // :386-387
    PointSide pts(vpPoints.size());
    for (size_t i = 0; i < vpPoints.size(); i++) {
        MapPoint *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP))              // :331-332
            continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        float invz, dist;
        if (!project(Rcw, tcw, p3Dw, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pts.u[i], pts.v[i], invz))
            continue;
        if (!in_range_and_facing(pMP, p3Dw, Ow, dist))
            continue;
        pts.level[i] = pMP->PredictScale(dist, pKF);
        pts.valid[i] = 1;
        put_desc(pMP, &pts.desc[32 * i]);
    }
    vector<int32_t> match((size_t)(pKF->N > 0 ? pKF->N : 1));
    int nmatches = 0;
    ORBX_CAPTURE(kf.ff, &pts.pp);
    if (orbx_search_by_projection_sim3(orbx_adapter::Device(), &kf.ff, &pts.pp, &pKF->mvScaleFactors[0], (int)pKF->mvScaleFactors.size(), (float)th, &match[0], &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int idx = 0; idx < pKF->N; idx++)
        if (match[idx] >= 0)
            vpMatched[idx] = vpPoints[match[idx]];                  // :408
    return nmatches;
}

int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th)
{
    const cv::Mat Rcw = pKF->GetRotation();
    const cv::Mat tcw = pKF->GetTranslation();
    const float bf = pKF->mbf;
    const cv::Mat Ow = pKF->GetCameraCenter();
    FrameSide kf;
    keyframe_side(pKF, kf);
    PointSide pts(vpMapPoints.size());
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF))         // :889-893 (re-checked below: earlier Replace calls can change it)
            continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        float invz, dist3D;
        if (!project(Rcw, tcw, p3Dw, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pts.u[i], pts.v[i], invz))
            continue;
        pts.aux[i] = pts.u[i] - bf * invz;                          // ur, :912
        if (!in_range_and_facing(pMP, p3Dw, Ow, dist3D))
            continue;
        pts.level[i] = pMP->PredictScale(dist3D, pKF);
        pts.valid[i] = 1;
        put_desc(pMP, &pts.desc[32 * i]);
    }
    const size_t np = vpMapPoints.size() ? vpMapPoints.size() : 1;
    vector<int32_t> best(np, -1);
    int nfound = 0;
    ORBX_CAPTURE(kf.ff, &pts.pp);
    if (orbx_window_best(orbx_adapter::Device(), &kf.ff, &pts.pp, &pKF->mvScaleFactors[0], &pKF->mvInvLevelSigma2[0], (int)pKF->mvScaleFactors.size(), th, 1, TH_LOW,
                         &best[0], NULL, &nfound) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    // the map surgery of :1011-1033, in the reference's order.  A point's best keypoint does not depend on the map state, but whether
    // the point is still good and not yet in the keyframe does: an earlier Replace / AddObservation of this loop can have changed it.
    int nFused = 0;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (best[i] < 0 || !pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF))
            continue;
        MapPoint *pMPinKF = pKF->GetMapPoint((size_t)best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations())
                    pMP->Replace(pMPinKF);
                else
                    pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, (size_t)best[i]);
            pKF->AddMapPoint(pMP, (size_t)best[i]);
        }
        nFused++;
    }
    return nFused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th, vector<MapPoint *> &vpReplacePoint)
{
    cv::Mat Rcw, tcw, Ow;
    decompose_sim3(Scw, Rcw, tcw, Ow);
    const set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
    FrameSide kf;
    keyframe_side(pKF, kf);
    PointSide pts(vpPoints.size());
    for (size_t i = 0; i < vpPoints.size(); i++) {
        MapPoint *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP))              // :1066-1067
            continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        float invz, dist3D;
        if (!project(Rcw, tcw, p3Dw, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pts.u[i], pts.v[i], invz))
            continue;
        if (!in_range_and_facing(pMP, p3Dw, Ow, dist3D))
            continue;
        pts.level[i] = pMP->PredictScale(dist3D, pKF);
        pts.valid[i] = 1;
        put_desc(pMP, &pts.desc[32 * i]);
    }
    const size_t np = vpPoints.size() ? vpPoints.size() : 1;
    vector<int32_t> best(np, -1);
    int nfound = 0;
    ORBX_CAPTURE(kf.ff, &pts.pp);
    if (orbx_window_best(orbx_adapter::Device(), &kf.ff, &pts.pp, &pKF->mvScaleFactors[0], NULL, (int)pKF->mvScaleFactors.size(), th, 0, TH_LOW, &best[0], NULL, &nfound) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    int nFused = 0;
    for (size_t i = 0; i < vpPoints.size(); i++) {                   // :1140-1157
        if (best[i] < 0)
            continue;
        MapPoint *pMP = vpPoints[i];
        MapPoint *pMPinKF = pKF->GetMapPoint((size_t)best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad())
                vpReplacePoint[i] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, (size_t)best[i]);
            pKF->AddMapPoint(pMP, (size_t)best[i]);
        }
        nFused++;
    }
    return nFused;
}

int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                             const float th)
{
    // the reference projects into BOTH keyframes with pKF1's intrinsics (:1169-1172): kept
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
    const cv::Mat R1w = pKF1->GetRotation();
    const cv::Mat t1w = pKF1->GetTranslation();
    const cv::Mat R2w = pKF2->GetRotation();
    const cv::Mat t2w = pKF2->GetTranslation();
    const cv::Mat sR12 = s12 * R12;
    const cv::Mat sR21 = (1.0 / s12) * R12.t();
    const cv::Mat t21 = -sR21 * t12;
    const vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                   // :1191-1201
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2)
                vbAlreadyMatched2[idx2] = true;
        }
    }
    FrameSide kf1, kf2;
    keyframe_side(pKF1, kf1);
    keyframe_side(pKF2, kf2);
    PointSide p12((size_t)N1), p21((size_t)N2);
    for (int dir = 0; dir < 2; dir++) {
        // dir 0: KF1's points into KF2 (:1207-1262); dir 1: the mirror image (:1289-1344)
        const vector<MapPoint *> &vp = dir == 0 ? vpMapPoints1 : vpMapPoints2;
        const vector<bool> &done = dir == 0 ? vbAlreadyMatched1 : vbAlreadyMatched2;
        const cv::Mat &Ra = dir == 0 ? R1w : R2w, &ta = dir == 0 ? t1w : t2w, &sRb = dir == 0 ? sR21 : sR12, &tb = dir == 0 ? t21 : t12;
        KeyFrame *target = dir == 0 ? pKF2 : pKF1;
        PointSide &pts = dir == 0 ? p12 : p21;
        for (size_t i = 0; i < vp.size(); i++) {
            MapPoint *pMP = vp[i];
            if (!pMP || done[i] || pMP->isBad())
                continue;
            const cv::Mat p3Dw = pMP->GetWorldPos();
            const cv::Mat p3Da = Ra * p3Dw + ta;
            const cv::Mat p3Db = sRb * p3Da + tb;
            if (p3Db.at<float>(2) < 0.0)
                continue;
            const float invz = 1.0 / p3Db.at<float>(2);
            const float x = p3Db.at<float>(0) * invz;
            const float y = p3Db.at<float>(1) * invz;
            pts.u[i] = fx * x + cx;
            pts.v[i] = fy * y + cy;                                  // IsInImage (:1233, :1315) runs on the device
            const float dist3D = (float)cv::norm(p3Db);
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance())
                continue;
            pts.level[i] = pMP->PredictScale(dist3D, target);
            pts.valid[i] = 1;
            put_desc(pMP, &pts.desc[32 * i]);
        }
    }
    vector<int32_t> m12((size_t)(N1 > 0 ? N1 : 1));
    int nFound = 0;
    ORBX_CAPTURE(kf2.ff, &p12.pp);
    ORBX_CAPTURE2(kf1.ff, &p21.pp);
    if (orbx_search_by_sim3(orbx_adapter::Device(), &kf1.ff, &kf2.ff, &p12.pp, &p21.pp, &pKF1->mvScaleFactors[0], &pKF2->mvScaleFactors[0], (int)pKF1->mvScaleFactors.size(), th,
                            &m12[0], &nFound) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int i1 = 0; i1 < N1; i1++)
        if (m12[i1] >= 0)
            vpMatches12[i1] = vpMapPoints2[m12[i1]];                 // :1383
    return nFound;
}

} // namespace ORB_SLAM2
