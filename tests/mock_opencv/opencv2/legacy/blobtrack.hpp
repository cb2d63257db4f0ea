// declaration-only mock of the parts of OpenCV 2.4's opencv2/legacy/blobtrack.hpp the adapters touch, see ../../README.md
#pragma once
#include "../opencv.hpp"
struct CvBlob {
  float x, y; /* blob position */
  float w, h; /* blob sizes    */
  int ID;     /* blob ID       */
};
inline CvBlob cvBlob(float x, float y, float w, float h) {
  CvBlob B = {x, y, w, h, 0};
  return B;
}
#define CV_BLOB_MINW 5
#define CV_BLOB_MINH 5
class CvBlobSeq {
 public:
  CvBlobSeq(int BlobSize = sizeof(CvBlob));
  virtual ~CvBlobSeq();
  virtual CvBlob* GetBlob(int BlobIndex);
  virtual CvBlob* GetBlobByID(int BlobID);
  virtual void DelBlob(int BlobIndex);
  virtual void Clear();
  virtual void AddBlob(CvBlob* pB);
  virtual int GetBlobNum();
};
class CvVSModule {
 public:
  CvVSModule();
  virtual ~CvVSModule();
  void SetNickName(const char* pStr);
  virtual void Release() = 0;

 protected:
  void SetTypeName(const char* name);
};
class CvFGDetector : public CvVSModule {
 public:
  CvFGDetector();
  virtual IplImage* GetMask() = 0;
  virtual void Process(IplImage* pImg) = 0;
  virtual void Release() = 0;
};
