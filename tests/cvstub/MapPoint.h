// stand-in for the reference's include/MapPoint.h: the members the matcher adaptors (adapter/ORBmatcher_bow.cc,
// adapter/ORBmatcher_proj.cc) call and the ones adapter/MapPoint_distinctive.cc defines / touches (same names and types)
#ifndef CVSTUB_MAPPOINT_H
#define CVSTUB_MAPPOINT_H
#include <map>
#include <mutex>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class MapPoint
{
public:
    explicit MapPoint(bool bad = false) : mTrackProjX(0), mTrackProjY(0), mTrackProjXR(0), mbTrackInView(false), mnTrackScaleLevel(0),
                                          mTrackViewCos(0), nObs(0), mbBad(bad) {}
    MapPoint(const MapPoint &o) : mTrackProjX(o.mTrackProjX), mTrackProjY(o.mTrackProjY), mTrackProjXR(o.mTrackProjXR), mbTrackInView(o.mbTrackInView),
                                  mnTrackScaleLevel(o.mnTrackScaleLevel), mTrackViewCos(o.mTrackViewCos), mWorldPos(o.mWorldPos), mObservations(o.mObservations),
                                  mDescriptor(o.mDescriptor), nObs(o.nObs), mbBad(o.mbBad) {}
    bool isBad() { return mbBad; }                       // include/MapPoint.h: bool isBad();
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return nObs; }
    void ComputeDistinctiveDescriptors();               // defined by adapter/MapPoint_distinctive.cc

    // the variables Tracking::SearchLocalPoints / Frame::isInFrustum leave for SearchByProjection (include/MapPoint.h:89-95)
    float mTrackProjX, mTrackProjY, mTrackProjXR;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;

    cv::Mat mWorldPos;                                  // protected in the reference
    std::map<KeyFrame *, size_t> mObservations;         // protected in the reference
    cv::Mat mDescriptor;                                // protected in the reference
    int nObs;
    bool mbBad;                                         // protected in the reference
    std::mutex mMutexFeatures;                          // protected in the reference
};
}
#endif
