// ustc_bgs.h — host-side mirror of USTC_BGS (ustc_src/ustc_bgs.{h,cpp}), the CvFGDetector the tracker plugs into
// cvCreateBlobTrackerAuto1 (ustc_src/trackingMain.cpp:33-35, :613-618; the shipped build uses type 36 = SuBSENSE).
// Same type table, same Process / GetMask / Release protocol; OpenCV-legacy's CvFGDetector base and IplImage are not in this
// image, so the mask is handed out as a bgs_hip::Image (INTEGRATION.md shows the IplImage-returning version for the reference).
// Types whose class is outside the hot path (ustc_bgs.cpp:23-58: dp Prati/Eigen/Texture, tb/, jmo/, lb/, ck/, av/, ae/, db/, sjn/)
// throw instead of silently running something else.
#pragma once
#include "bgs_host.h"
#include "blob.h"

namespace bgs_hip {

class USTC_BGS {
 public:
  int frameNum;
  IBGS* bgs;
  Image img_mask, img_bkgmodel, img_input;

  explicit USTC_BGS(int type) : frameNum(0), bgs(nullptr) {  // ustc_bgs.cpp:3-69
    const int i = type;
    if (!(i >= 0 && i <= 37)) throw Exception(BGS_ERR_INVALID, "USTC_BGS: type must be 0..37");  // CV_Assert(i>=0&&i<=37)
    if (i == 0) bgs = new FrameDifferenceBGS;
    if (i == 1) bgs = new StaticFrameDifferenceBGS;
    if (i == 2) bgs = new WeightedMovingMeanBGS;
    if (i == 3) bgs = new WeightedMovingVarianceBGS;
    if (i == 4) bgs = new MixtureOfGaussianV1BGS;
    if (i == 5) bgs = new MixtureOfGaussianV2BGS;
    if (i == 6) bgs = new AdaptiveBackgroundLearning;
    if (i == 7) bgs = new AdaptiveSelectiveBackgroundLearning;
    if (i == 8) bgs = new GMG;
    if (i == 9) bgs = new DPAdaptiveMedianBGS;
    if (i == 10) bgs = new DPGrimsonGMMBGS;
    if (i == 11) bgs = new DPZivkovicAGMMBGS;
    if (i == 12) bgs = new DPMeanBGS;
    if (i == 13) bgs = new DPWrenGABGS;
    if (i == 35) bgs = new SigmaDeltaBGS;
    if (i == 36) bgs = new SuBSENSEBGS();
    if (i == 37) bgs = new LOBSTERBGS();
    if (!bgs) throw Exception(BGS_ERR_UNSUPPORTED, "USTC_BGS: type " + std::to_string(i) + " is outside the package_bgs hot path built here");
  }
  ~USTC_BGS() {}
  void Release() { delete bgs, bgs = nullptr; }  // :75-77

  // the mask of the last processed frame; NULL before the first one (:79-85)
  const Image* GetMask() const { return frameNum == 0 ? nullptr : &img_mask; }

  void Process(const Image& pImg) {  // :87-113
    img_input = pImg;
    bgs->process(img_input, img_mask, img_bkgmodel);
    if (img_mask.empty()) std::cout << "img_mask is empty " << frameNum << std::endl;
    frameNum++;
  }
};

// N2: the FG detector with the blob list hand-off.  USTC_BGS hands the tracker a full mask (GetMask) and OpenCV-legacy's
// CvBlobDetector extracts the foreground regions from it on the CPU (ustc_src/trackingMain.cpp:56-57, :166).  With the mask
// already in HBM the regions are found there (kernel_cc.h) and only their rectangles and moments come back: GetBlobs() fills a
// CvBlobSeq with one CvBlob {x, y, w, h, ID} per region of at least CV_BLOB_MINW x CV_BLOB_MINH pixels, in raster order of the
// regions' first pixels, IDs counting up from the detector's own counter (CvBlobTrackerAuto1 numbers new blobs the same way).
class HipFGDetector : public USTC_BGS {
 public:
  explicit HipFGDetector(int type) : USTC_BGS(type), nextBlobID(0) {}
  int nextBlobID;

  // fromMoments: centre = centroid, size = 4 sigma (blob_from_moments); otherwise the bounding rectangle (blob_from_box)
  int GetBlobs(CvBlobSeq* pBlobs, int connectivity = 8, bool fromMoments = true) {
    pBlobs->Clear();
    if (frameNum == 0 || img_mask.empty()) return 0;
    HipBGSBase* h = dynamic_cast<HipBGSBase*>(bgs);
    if (!h) throw Exception(BGS_ERR_STATE, "HipFGDetector: the IBGS object is not a libbgs_hip class");
    h->lastMaskBlobs(connectivity, CV_BLOB_MINW, CV_BLOB_MINH, boxes_, moments_);
    for (size_t i = 0; i < boxes_.size(); ++i) {
      CvBlob b = cvBlob(0, 0, 0, 0);
      if (fromMoments)
        bgs_hip_convert::blob_from_moments(boxes_[i], moments_[i], b);
      else
        bgs_hip_convert::blob_from_box(boxes_[i], b);
      b.ID = nextBlobID++;
      pBlobs->AddBlob(&b);
    }
    return pBlobs->GetBlobNum();
  }
  const std::vector<bgs_box>& lastBoxes() const { return boxes_; }

 private:
  std::vector<bgs_box> boxes_;
  std::vector<bgs_moments> moments_;
};

}  // namespace bgs_hip
