// stand-in for the reference's include/MapPoint.h: the members the matcher adaptors (adapter/ORBmatcher_bow.cc, ORBmatcher_proj.cc,
// ORBmatcher_fuse.cc) call and the ones adapter/MapPoint_distinctive.cc defines / touches (same names and types); plain data behind
// the accessors, no map bookkeeping
#ifndef CVSTUB_MAPPOINT_H
#define CVSTUB_MAPPOINT_H
#include <map>
#include <mutex>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class Frame;
class MapPoint
{
public:
    explicit MapPoint(bool bad = false) { init(); mbBad = bad; }
    MapPoint(const MapPoint &o) { copy(o); }
    MapPoint &operator=(const MapPoint &o) { copy(o); return *this; }

    bool isBad() { return mbBad; }                       // include/MapPoint.h: bool isBad();
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    int Observations() { return nObs; }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, KeyFrame *pKF);     // src/MapPoint.cc:393-415; defined in tests/adapter_driver.cc
    int PredictScale(const float &currentDist, Frame *pF);
    void AddObservation(KeyFrame *pKF, size_t idx) { if (!mObservations.count(pKF)) { mObservations[pKF] = idx; nObs++; } }
    void Replace(MapPoint *pMP) { mbBad = true; mpReplaced = pMP; }
    void ComputeDistinctiveDescriptors();               // defined by adapter/MapPoint_distinctive.cc

    // the variables Tracking::SearchLocalPoints / Frame::isInFrustum leave for SearchByProjection (include/MapPoint.h:89-95)
    float mTrackProjX, mTrackProjY, mTrackProjXR;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;

    // protected in the reference
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    float mfMinDistance, mfMaxDistance;
    MapPoint *mpReplaced;
    int nObs;
    bool mbBad;
    std::mutex mMutexFeatures;

private:
    void init()
    {
        mTrackProjX = mTrackProjY = mTrackProjXR = 0.f; mbTrackInView = false; mnTrackScaleLevel = 0; mTrackViewCos = 0.f;
        mfMinDistance = 0.f; mfMaxDistance = 1e9f; mpReplaced = NULL; nObs = 0; mbBad = false;
    }
    void copy(const MapPoint &o)
    {
        mTrackProjX = o.mTrackProjX; mTrackProjY = o.mTrackProjY; mTrackProjXR = o.mTrackProjXR; mbTrackInView = o.mbTrackInView;
        mnTrackScaleLevel = o.mnTrackScaleLevel; mTrackViewCos = o.mTrackViewCos; mWorldPos = o.mWorldPos; mNormalVector = o.mNormalVector;
        mDescriptor = o.mDescriptor; mObservations = o.mObservations; mfMinDistance = o.mfMinDistance; mfMaxDistance = o.mfMaxDistance;
        mpReplaced = o.mpReplaced; nObs = o.nObs; mbBad = o.mbBad;
    }
};
}
#endif
