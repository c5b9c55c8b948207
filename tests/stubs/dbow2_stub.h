// dbow2_stub.h -- DECLARATIONS-ONLY stand-in for the DBoW2 names the reference's headers mention (see cv_stub.h: syntax check only).
#ifndef CMS_TEST_DBOW2_STUB_H
#define CMS_TEST_DBOW2_STUB_H
#include <map>
#include <string>
#include <vector>
#include "cv_stub.h"
using namespace std;      // (the real TemplatedVocabulary.h:36 says so at global scope, and the reference's headers lean on it)
namespace DBoW2 {
typedef unsigned int WordId; typedef double WordValue; typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> { public: BowVector(); ~BowVector(); };
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > { public: FeatureVector(); ~FeatureVector(); };
class FORB { public: typedef cv::Mat TDescriptor; typedef const TDescriptor* pDescriptor; static const int L = 32; static int distance(const TDescriptor&, const TDescriptor&); };
template <class TDescriptor, class F> class TemplatedVocabulary {
 public:
  TemplatedVocabulary(); virtual ~TemplatedVocabulary();
  bool loadFromTextFile(const std::string&); void saveToTextFile(const std::string&) const;
  virtual void transform(const std::vector<TDescriptor>&, BowVector&, FeatureVector&, int) const;
  double score(const BowVector&, const BowVector&) const; unsigned int size() const; bool empty() const;
};
}  // namespace DBoW2
#endif
