// adapter/ORBmatcher_proj.cc -- the per-frame projection searches of ORB_SLAM2::ORBmatcher over liborbx: replaces
//   SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)            reference src/ORBmatcher.cc:48-129
//       (Tracking::SearchLocalPoints, every tracked frame)
//   SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)         :1396-1553
//       (Tracking::TrackWithMotionModel, every tracked frame)
//   SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)            :430-556
//       (Tracking::MonocularInitialization)
// The adaptor does what only the host can do -- reads the MapPoint / Frame objects under their accessors and, for the motion-model
// search, projects the last frame's points with the current pose using the reference's own cv::Mat expressions (so the arithmetic is
// OpenCV's) -- and hands the window search, the level / uRight gates, the Hamming distances, the order-dependent claiming of
// features and the rotation-consistency filter to the device.  The other five projection-type searches (relocalisation, Sim3,
// both Fuse overloads, SearchBySim3) follow the same split; INTEGRATION.md section 3c lists their field mappings.
#include "ORBmatcher.h"

#include <stdexcept>

#include "orbx_adapter.h"

using namespace std;

namespace ORB_SLAM2
{

using orbx_adapter::FrameSide;
using orbx_adapter::PointSide;

// occupied[i]: the feature already holds a MapPoint that blocks it for this search
//   with_observations = true : mvpMapPoints[i] && mvpMapPoints[i]->Observations() > 0   (:85-87, :1468-1470)
static void frame_side(const Frame &F, FrameSide &s)
{
    const size_t n = (size_t)F.N;
    s.x.resize(n); s.y.resize(n); s.angle.resize(n); s.octave.resize(n); s.occupied.assign(n, 0);
    for (size_t i = 0; i < n; i++) {
        const cv::KeyPoint &kp = F.mvKeysUn[i];
        s.x[i] = kp.pt.x; s.y[i] = kp.pt.y; s.angle[i] = kp.angle; s.octave[i] = kp.octave;
        MapPoint *pMP = i < F.mvpMapPoints.size() ? F.mvpMapPoints[i] : static_cast<MapPoint *>(NULL);
        s.occupied[i] = (pMP && pMP->Observations() > 0) ? 1 : 0;
    }
    memset(&s.ff, 0, sizeof s.ff);
    s.ff.n = F.N;
    if (n) {
        s.ff.x = &s.x[0]; s.ff.y = &s.y[0]; s.ff.octave = &s.octave[0]; s.ff.angle = &s.angle[0];
        s.ff.u_right = &F.mvuRight[0]; s.ff.desc = orbx_adapter::dense_descriptors(F.mDescriptors, F.N); s.ff.occupied = &s.occupied[0];
    }
    s.ff.min_x = Frame::mnMinX; s.ff.min_y = Frame::mnMinY; s.ff.max_x = Frame::mnMaxX; s.ff.max_y = Frame::mnMaxY;
}

static void put_descriptor(MapPoint *pMP, uint8_t *dst)
{
    const cv::Mat d = pMP->GetDescriptor();            // a clone taken under the point's mutex, as the reference reads it (:70, :1458)
    if (!d.empty())
        memcpy(dst, d.data, 32);
}

int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint *> &vpMapPoints, const float th)
{
    FrameSide cur;
    frame_side(F, cur);
    PointSide pts(vpMapPoints.size());
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP->mbTrackInView || pMP->isBad())         // :56-60
            continue;
        pts.valid[i] = 1;
        pts.u[i] = pMP->mTrackProjX; pts.v[i] = pMP->mTrackProjY; pts.aux[i] = pMP->mTrackProjXR;
        pts.level[i] = pMP->mnTrackScaleLevel;
        pts.view_cos[i] = pMP->mTrackViewCos;
        pts.has_obs[i] = pMP->Observations() > 0 ? 1 : 0;
        put_descriptor(pMP, &pts.desc[32 * i]);
    }
    vector<int32_t> match((size_t)(F.N > 0 ? F.N : 1));
    int nmatches = 0;
    ORBX_CAPTURE(cur.ff, &pts.pp);
    if (orbx_search_by_projection_map_points(orbx_adapter::Device(), &cur.ff, &pts.pp, &F.mvScaleFactors[0], (int)F.mvScaleFactors.size(), th, mfNNratio, &match[0],
                                             &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int f = 0; f < F.N; f++)
        if (match[f] >= 0)
            F.mvpMapPoints[f] = vpMapPoints[match[f]];    // :120
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    // poses and the forward / backward test, in the reference's own matrix expressions (:1404-1416)
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;

    FrameSide cur;
    frame_side(CurrentFrame, cur);
    PointSide pts((size_t)LastFrame.N);
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i])             // :1419-1423
            continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);     // double division, rounded to float: :1434
        pts.valid[i] = 1;
        pts.aux[i] = invzc;                              // the device drops invzc < 0 and points outside the image bounds (:1436-1449)
        pts.u[i] = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        pts.v[i] = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        pts.level[i] = LastFrame.mvKeys[i].octave;
        pts.angle[i] = LastFrame.mvKeysUn[i].angle;
        pts.has_obs[i] = pMP->Observations() > 0 ? 1 : 0;
        put_descriptor(pMP, &pts.desc[32 * i]);
    }
    vector<int32_t> match((size_t)(CurrentFrame.N > 0 ? CurrentFrame.N : 1));
    int nmatches = 0;
    ORBX_CAPTURE(cur.ff, &pts.pp);
    if (orbx_search_by_projection_last_frame(orbx_adapter::Device(), &cur.ff, &pts.pp, &CurrentFrame.mvScaleFactors[0], (int)CurrentFrame.mvScaleFactors.size(), th,
                                             bForward ? 1 : bBackward ? 2 : 0, CurrentFrame.mbf, mbCheckOrientation ? 3 : 0, &match[0],
                                             &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int f = 0; f < CurrentFrame.N; f++) {
        if (match[f] >= 0)
            CurrentFrame.mvpMapPoints[f] = LastFrame.mvpMapPoints[match[f]];   // :1500
        else if (match[f] == -2)
            CurrentFrame.mvpMapPoints[f] = static_cast<MapPoint *>(NULL);      // matched, then cleared by the rotation filter (:1526-1545)
    }
    return nmatches;
}

int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
{
    FrameSide f1, f2;
    frame_side(F1, f1);
    frame_side(F2, f2);
    const size_t n1 = F1.mvKeysUn.size();
    vector<float> prev(2 * (n1 ? n1 : 1));
    for (size_t i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    vector<int32_t> m12(n1 ? n1 : 1);
    int nmatches = 0;
    ORBX_CAPTURE(f2.ff, NULL);
    if (orbx_search_for_initialization(orbx_adapter::Device(), &f1.ff, &f2.ff, &prev[0], windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, &m12[0], &nmatches) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    vnMatches12 = vector<int>(n1, -1);                  // :433
    for (size_t i1 = 0; i1 < n1; i1++) {
        vnMatches12[i1] = m12[i1];
        if (m12[i1] >= 0)
            vbPrevMatched[i1] = F2.mvKeysUn[m12[i1]].pt; // :550-552
    }
    return nmatches;
}

} // namespace ORB_SLAM2
