// stand-in for the reference's include/MapPoint.h: the one member the matcher adaptor calls
#ifndef CVSTUB_MAPPOINT_H
#define CVSTUB_MAPPOINT_H
namespace ORB_SLAM2 {
class MapPoint
{
public:
    explicit MapPoint(bool bad = false) : mbBad(bad) {}
    bool isBad() { return mbBad; }       // include/MapPoint.h: bool isBad();
private:
    bool mbBad;
};
}
#endif
