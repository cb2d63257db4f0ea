// HipFGDetector.h — reference-side FG detector over libbgs_hip with the blob list hand-off.  Add next to HipBGS.h in the
// USTC-Computer-Vision/tracking tree; replaces USTC_BGS (ustc_src/ustc_bgs.{h,cpp}) in ustc_src/trackingMain.cpp:
//
//     HipFGDetector* bgs = new HipFGDetector(type);      // trackingMain.cpp:35   was: USTC_BGS* bgs = new USTC_BGS(type);
//     param.pFG = bgs;                                    // :613                  unchanged
//
// Process / GetMask / Release are USTC_BGS's, statement for statement (ustc_bgs.cpp:75-113), over the hipbgs:: classes of HipBGS.h
// for the types on the hot path (0-13 except the three dp classes outside it, 35-37).  GetBlobs() is the addition: the foreground
// regions of the last mask as CvBlob {x, y, w, h, ID}, found on the device copy of the mask, so a blob-detection module can seed
// from rectangles instead of re-scanning the mask on the CPU (CvBlobDetector::DetectNewBlob receives the mask at :166).
//
// Needs OpenCV 2.4 with the legacy module (opencv2/legacy/blobtrack.hpp), which this repository's build image does not have:
// checked for syntax against the declaration-only mock under tests/mock_opencv (tests/test_capi_cpu.py), as gnu++0x like the
// reference's CMakeLists.txt:5.  The logic is shared with the tested mirror tracking_amd/host/ustc_bgs.h.
#pragma once
#include <vector>

#include "opencv2/legacy/blobtrack.hpp"
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "HipBGS.h"
#include "blob_convert.h"

class HipFGDetector : public CvFGDetector {
 public:
  int frameNum;
  IplImage* c_mask;
  IBGS* bgs;
  cv::Mat img_mask;
  cv::Mat img_bkgmodel;
  cv::Mat img_input;
  IplImage b;
  int nextBlobID;

  explicit HipFGDetector(int type) : frameNum(0), c_mask(0), bgs(0), nextBlobID(0) {  // ustc_bgs.cpp:3-69
    const int i = type;
    CV_Assert(i >= 0 && i <= 37);
    if (i == 0) bgs = new hipbgs::FrameDifferenceBGS;
    if (i == 1) bgs = new hipbgs::StaticFrameDifferenceBGS;
    if (i == 2) bgs = new hipbgs::WeightedMovingMeanBGS;
    if (i == 3) bgs = new hipbgs::WeightedMovingVarianceBGS;
    if (i == 4) bgs = new hipbgs::MixtureOfGaussianV1BGS;
    if (i == 5) bgs = new hipbgs::MixtureOfGaussianV2BGS;
    if (i == 6) bgs = new hipbgs::AdaptiveBackgroundLearning;
    if (i == 7) bgs = new hipbgs::AdaptiveSelectiveBackgroundLearning;
    if (i == 8) bgs = new hipbgs::GMG;
    if (i == 9) bgs = new hipbgs::DPAdaptiveMedianBGS;
    if (i == 10) bgs = new hipbgs::DPGrimsonGMMBGS;
    if (i == 11) bgs = new hipbgs::DPZivkovicAGMMBGS;
    if (i == 12) bgs = new hipbgs::DPMeanBGS;
    if (i == 13) bgs = new hipbgs::DPWrenGABGS;
    if (i == 35) bgs = new hipbgs::SigmaDeltaBGS;
    if (i == 36) bgs = new hipbgs::SuBSENSEBGS();
    if (i == 37) bgs = new hipbgs::LOBSTERBGS();
    if (!bgs) CV_Error(CV_StsBadArg, "HipFGDetector: this type is outside the package_bgs hot path libbgs_hip covers");
  }
  ~HipFGDetector() {}
  void Release() { delete bgs, bgs = 0; }  // :75-77

  IplImage* GetMask() {  // :79-85
    if (frameNum == 0) return NULL;
    return c_mask;
  }

  void Process(IplImage* pImg) {  // :87-113
    img_input = cv::Mat(pImg);
    bgs->process(img_input, img_mask, img_bkgmodel);
    if (!img_mask.empty()) {
      b = img_mask.operator IplImage();
      c_mask = &b;
      frameNum++;
    } else {
      std::cout << "img_mask is empty " << frameNum << std::endl;
      frameNum++;
    }
  }

  // One CvBlob per 8- (or 4-) connected foreground region of the last mask that is at least CV_BLOB_MINW x CV_BLOB_MINH pixels,
  // in raster order of the regions' first pixels; IDs count up from nextBlobID.  fromMoments: centre = centroid, size = 4 sigma
  // (what the legacy detectors compute from cvMoments); otherwise centre and size of the bounding rectangle.
  int GetBlobs(CvBlobSeq* pBlobs, int connectivity = 8, bool fromMoments = true) {
    pBlobs->Clear();
    if (frameNum == 0 || img_mask.empty()) return 0;
    hipbgs::HipBGSBase* h = dynamic_cast<hipbgs::HipBGSBase*>(bgs);
    CV_Assert(h != 0);
    h->lastMaskBlobs(connectivity, CV_BLOB_MINW, CV_BLOB_MINH, boxes_, moments_);
    for (size_t k = 0; k < boxes_.size(); ++k) {
      CvBlob B = cvBlob(0, 0, 0, 0);
      if (fromMoments)
        bgs_hip_convert::blob_from_moments(boxes_[k], moments_[k], B);
      else
        bgs_hip_convert::blob_from_box(boxes_[k], B);
      B.ID = nextBlobID++;
      pBlobs->AddBlob(&B);
    }
    return pBlobs->GetBlobNum();
  }

 private:
  std::vector<bgs_box> boxes_;
  std::vector<bgs_moments> moments_;
};
