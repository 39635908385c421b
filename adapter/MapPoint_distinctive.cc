// adapter/MapPoint_distinctive.cc -- MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:266-340) over liborbx: the
// N x N Hamming matrix, the per-row medians and the arg-min run on the device (orbx_distinctive_descriptors); gathering the
// observations under the point's mutex and cloning the winner stay here.  orbx_adapter::DistinctiveDescriptors() is the batched form
// for LocalMapping::ProcessNewKeyFrame (src/LocalMapping.cc:131-160), which updates every map point the new keyframe sees: one call,
// one launch for all of them instead of one per point.
#include "MapPoint.h"

#include <stdexcept>
#include <vector>

#include "KeyFrame.h"
#include "orbx_adapter.h"

using namespace std;

namespace ORB_SLAM2
{

// the descriptors of the point's non-bad observations, in the order of the reference's std::map walk (:283-290)
static void gather(MapPoint *pMP, vector<cv::Mat> &vDescriptors)
{
    map<KeyFrame *, size_t> observations;
    {
        unique_lock<mutex> lock1(pMP->mMutexFeatures);
        if (pMP->mbBad)
            return;
        observations = pMP->mObservations;
    }
    vDescriptors.reserve(observations.size());
    for (map<KeyFrame *, size_t>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
        KeyFrame *pKF = mit->first;
        if (!pKF->isBad())
            vDescriptors.push_back(pKF->mDescriptors.row((int)mit->second));
    }
}

void MapPoint::ComputeDistinctiveDescriptors()
{
    vector<cv::Mat> vDescriptors;
    gather(this, vDescriptors);
    if (vDescriptors.empty())                            // :277-278, :292-293
        return;
    vector<uint8_t> flat(32 * vDescriptors.size());
    for (size_t i = 0; i < vDescriptors.size(); i++)
        memcpy(&flat[32 * i], vDescriptors[i].data, 32);
    const int32_t off[2] = { 0, (int32_t)vDescriptors.size() };
    int32_t best = -1;
    if (orbx_distinctive_descriptors(orbx_adapter::Device(), &flat[0], off, 1, &best) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    {
        unique_lock<mutex> lock(mMutexFeatures);
        mDescriptor = vDescriptors[best].clone();         // :336-339
    }
}

} // namespace ORB_SLAM2

namespace orbx_adapter
{

// every point of the list in one device call
void DistinctiveDescriptors(const std::vector<ORB_SLAM2::MapPoint *> &points)
{
    std::vector<std::vector<cv::Mat> > descs(points.size());
    std::vector<int32_t> off(points.size() + 1, 0);
    std::vector<uint8_t> flat;
    for (size_t p = 0; p < points.size(); p++) {
        ORB_SLAM2::gather(points[p], descs[p]);
        for (size_t i = 0; i < descs[p].size(); i++)
            flat.insert(flat.end(), descs[p][i].data, descs[p][i].data + 32);
        off[p + 1] = off[p] + (int32_t)descs[p].size();
    }
    if (points.empty())
        return;
    std::vector<int32_t> best(points.size(), -1);
    if (orbx_distinctive_descriptors(orbx_adapter::Device(), flat.empty() ? NULL : &flat[0], &off[0], (int)points.size(), &best[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (size_t p = 0; p < points.size(); p++)
        if (best[p] >= 0) {
            std::unique_lock<std::mutex> lock(points[p]->mMutexFeatures);
            points[p]->mDescriptor = descs[p][best[p]].clone();
        }
}

} // namespace orbx_adapter
