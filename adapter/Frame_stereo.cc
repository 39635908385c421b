// adapter/Frame_stereo.cc -- Frame::ComputeStereoMatches over liborbx (replaces the reference's src/Frame.cc:577-751;
// the rest of src/Frame.cc is compiled as it is, with this method's body removed or #if'd out).
#include "Frame.h"

#include <stdexcept>

#include <orbx.h>

#include "orbx_adapter.h"

namespace ORB_SLAM2
{

void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);     // :579-580
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0)
        return;
    // The reference reads the member mb at :607 before the constructor assigns it (:118); the intended value is
    // mb = mbf / fx, which makes maxD = mbf / minZ = fx.  The ABI takes it explicitly.
    const float b = mbf / fx;
    if (orbx_stereo_match(mpORBextractorLeft->Handle(), mpORBextractorRight->Handle(), reinterpret_cast<const orbx_keypoint *>(&mvKeys[0]),
                          orbx_adapter::dense_descriptors(mDescriptors, N), N,
                          mvKeysRight.empty() ? NULL : reinterpret_cast<const orbx_keypoint *>(&mvKeysRight[0]),
                          orbx_adapter::dense_descriptors(mDescriptorsRight, (int)mvKeysRight.size()), (int)mvKeysRight.size(), mbf, b, &mvuRight[0],
                          &mvDepth[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
}

} // namespace ORB_SLAM2
