// adapter/orbx_batch.h -- the vocabulary-guided searches in the form the GPU wins with: keyframes RESIDENT in HBM and the loops the
// reference runs these searches in as ONE call each.
//
//   reference loop                                                            one call here
//   LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:241-309             SearchForTriangulationBatch(cur, neighbours, F12s, ...)
//     for each of 10-20 neighbour keyframes: ComputeF12 + matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false)
//   LoopClosing::ComputeSim3, src/LoopClosing.cc:293-323                      SearchByBoWBatch(cur, candidates, ...)
//     for each loop candidate: matcher.SearchByBoW(mpCurrentKF, pKF, vvpMapPointMatches[i])
//   Tracking::Relocalization, src/Tracking.cc:1661-1682                       SearchByBoWBatch(candidates, mCurrentFrame, ...)
//     for each relocalisation candidate: matcher.SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i])
//
// A single SearchByBoW / SearchForTriangulation call costs 20-31 us on the GPU (one launch + one PCIe round trip) against 11-16 us on
// a host core; twenty pairs in one call cost 3-4.5 us per pair, this file's work included (matchers).  What makes the batch cheap is that a keyframe's
// descriptors, FeatureVector and undistorted keypoints never change once it exists (src/KeyFrame.cc:29-60): KeyFrameCache keeps them
// in HBM (orbx_kf), and a call moves only the map-point flags, the node intersection and the results.
//
// The reference's include/ORBmatcher.h is not touched: mfNNratio / mbCheckOrientation are protected there, so these free functions
// take the two constructor arguments of the `ORBmatcher matcher(ratio, checkOri)` the loop declares (0.6 / false in
// CreateNewMapPoints, 0.75 / true in ComputeSim3 and Relocalization).
#ifndef ORBX_ADAPTER_BATCH_H
#define ORBX_ADAPTER_BATCH_H

#include <stddef.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include <opencv2/core/core.hpp>

#include <orbx.h>

namespace ORB_SLAM2
{
class KeyFrame;
class Frame;
class MapPoint;
}

namespace orbx_adapter
{

// One orbx_kf per KeyFrame*: made on first use from mDescriptors / mFeatVec / mvKeysUn / mvuRight (KeyFrame::ComputeBoW must have
// run: src/LocalMapping.cc:118, :246 guarantee it before any of the three loops), dropped when the keyframe goes away.
// Hook: call KeyFrameCache::instance().drop(this) from KeyFrame::SetBadFlag (src/KeyFrame.cc:459-503, after mbBad = true).
// Thread-safe (Tracking, LocalMapping and LoopClosing all match concurrently); the orbx_kf objects are immutable.
class KeyFrameCache
{
public:
    static KeyFrameCache &instance();
    const orbx_kf *get(ORB_SLAM2::KeyFrame *pKF);            // creates the resident copy if there is none yet; throws on failure
    const orbx_kf *find(const ORB_SLAM2::KeyFrame *pKF);     // NULL when pKF is not resident
    void drop(const ORB_SLAM2::KeyFrame *pKF);
    void clear();
    size_t size();
    ~KeyFrameCache();

private:
    KeyFrameCache() {}
    KeyFrameCache(const KeyFrameCache &);
    KeyFrameCache &operator=(const KeyFrameCache &);
    std::mutex mMutex;
    std::map<const ORB_SLAM2::KeyFrame *, orbx_kf *> mKFs;
};

// src/LocalMapping.cc:241-309 as one call: vF12[i] = ComputeF12(pKF1, vpKF2[i]) (3x3 CV_32F), vvMatchedPairs[i] = what
// matcher.SearchForTriangulation(pKF1, vpKF2[i], vF12[i], vMatchedIndices, bOnlyStereo) would have returned.  Returns the total.
int SearchForTriangulationBatch(ORB_SLAM2::KeyFrame *pKF1, const std::vector<ORB_SLAM2::KeyFrame *> &vpKF2, const std::vector<cv::Mat> &vF12,
                                std::vector<std::vector<std::pair<size_t, size_t> > > &vvMatchedPairs, bool bOnlyStereo,
                                float nnratio = 0.6f, bool checkOri = false);

// src/LoopClosing.cc:293-323 as one call: vvpMatches12[i] and vnMatches[i] = what matcher.SearchByBoW(pKF1, vpKF2[i], vvpMatches12[i])
// would have produced / returned.
void SearchByBoWBatch(ORB_SLAM2::KeyFrame *pKF1, const std::vector<ORB_SLAM2::KeyFrame *> &vpKF2,
                      std::vector<std::vector<ORB_SLAM2::MapPoint *> > &vvpMatches12, std::vector<int> &vnMatches,
                      float nnratio = 0.75f, bool checkOri = true);

// src/Tracking.cc:1661-1682 as one call: vvpMapPointMatches[i] and vnMatches[i] = what matcher.SearchByBoW(vpKFs[i], F,
// vvpMapPointMatches[i]) would have produced / returned.  Keyframes resident, the frame as host pointers (it lives one frame time).
void SearchByBoWBatch(const std::vector<ORB_SLAM2::KeyFrame *> &vpKFs, ORB_SLAM2::Frame &F,
                      std::vector<std::vector<ORB_SLAM2::MapPoint *> > &vvpMapPointMatches, std::vector<int> &vnMatches,
                      float nnratio = 0.75f, bool checkOri = true);

// flag arrays the resident searches take per call (exactly what adapter/ORBmatcher_bow.cc hands the host-pointer entries)
void GoodPointFlags(ORB_SLAM2::KeyFrame *pKF, std::vector<ORB_SLAM2::MapPoint *> &vpMapPoints, std::vector<uint8_t> &flag);   // SearchByBoW: non-bad MapPoint
void HasPointFlags(ORB_SLAM2::KeyFrame *pKF, std::vector<uint8_t> &flag);                                                      // SearchForTriangulation: any MapPoint
// epipole of camera 1 in image 2 (src/ORBmatcher.cc:712-718)
void Epipole(ORB_SLAM2::KeyFrame *pKF1, ORB_SLAM2::KeyFrame *pKF2, float &ex, float &ey);

} // namespace orbx_adapter

#endif
