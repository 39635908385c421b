// adapter/orbx_device.h -- which GPU the adaptors run on (shared by ORBextractor.cc and the matcher / Frame / MapPoint adaptors).
#ifndef ORBX_ADAPTER_DEVICE_H
#define ORBX_ADAPTER_DEVICE_H

#include <stdlib.h>

namespace orbx_adapter
{

// The device every adaptor call runs on.  ORB-SLAM2 is one process per camera stream (Examples/Stereo/stereo_kitti.cc:68-117); on a multi-GPU
// node each process picks its card once, before the first ORBextractor is constructed: SetDevice(i), or ORBX_DEVICE=i in the environment
// (default 0, which is also right under one-process-per-GPU with HIP_VISIBLE_DEVICES).  An index the runtime does not know makes every ABI
// call return ORBX_E_NO_DEVICE, and the adaptors throw.
inline int &device_slot()
{
    static int d = -1;
    return d;
}
inline int Device()
{
    int &d = device_slot();
    if (d < 0) {
        const char *env = getenv("ORBX_DEVICE");
        d = env && *env ? atoi(env) : 0;
        if (d < 0) d = 0;
    }
    return d;
}
inline void SetDevice(int device) { device_slot() = device < 0 ? 0 : device; }

} // namespace orbx_adapter

#endif
