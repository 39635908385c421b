// stand-in for the reference's Thirdparty/DBoW2/DBoW2/FeatureVector.h (:21-22): std::map<NodeId, std::vector<unsigned>>.
// tests/test_adapter.py also compiles the driver against the REAL header when the reference checkout is present.
#ifndef CVSTUB_FEATURE_VECTOR_H
#define CVSTUB_FEATURE_VECTOR_H
#include <map>
#include <vector>
#include "BowVector.h"
namespace DBoW2 {
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> >
{
public:
    void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
};
}
#endif
