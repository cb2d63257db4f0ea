// blob.h — host-side mirror of the blob types the tracker's modules exchange (opencv2/legacy/blobtrack.hpp, not in this image):
// CvBlob, cvBlob(), CvBlobSeq with the members ustc_src/trackingMain.cpp uses (GetBlobNum / GetBlob, :186-215) plus the ones a
// detector needs to fill a list (AddBlob, Clear, DelBlob, GetBlobByID).  Same names, same meaning; a std::vector underneath.
#pragma once
#include <vector>

#include "blob_convert.h"

namespace bgs_hip {

struct CvBlob {
  float x, y;  // blob position (centre)
  float w, h;  // blob sizes
  int ID;      // blob ID
};
inline CvBlob cvBlob(float x, float y, float w, float h) {
  CvBlob B = {x, y, w, h, 0};
  return B;
}
const int CV_BLOB_MINW = 5, CV_BLOB_MINH = 5;  // blobtrack.hpp: smallest blob the detectors report

class CvBlobSeq {
 public:
  int GetBlobNum() const { return (int)v_.size(); }
  CvBlob* GetBlob(int BlobIndex) { return (BlobIndex >= 0 && BlobIndex < (int)v_.size()) ? &v_[BlobIndex] : 0; }
  CvBlob* GetBlobByID(int BlobID) {
    for (size_t i = 0; i < v_.size(); ++i)
      if (v_[i].ID == BlobID) return &v_[i];
    return 0;
  }
  void DelBlob(int BlobIndex) {
    if (BlobIndex >= 0 && BlobIndex < (int)v_.size()) v_.erase(v_.begin() + BlobIndex);
  }
  void Clear() { v_.clear(); }
  void AddBlob(CvBlob* pB) { v_.push_back(*pB); }

 private:
  std::vector<CvBlob> v_;
};

}  // namespace bgs_hip
