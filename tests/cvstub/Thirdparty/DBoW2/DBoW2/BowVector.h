// stand-in for the reference's Thirdparty/DBoW2/DBoW2/BowVector.h (:20-26, :58): the typedefs the adaptor touches
#ifndef CVSTUB_BOW_VECTOR_H
#define CVSTUB_BOW_VECTOR_H
#include <map>
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
}
#endif
