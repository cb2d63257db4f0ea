// HipBGS.h — reference-side binding of libbgs_hip (add to the USTC-Computer-Vision/tracking tree, e.g. as package_bgs/hip/HipBGS.h).
//
// One IBGS class per reference class on the hot path, all in namespace hipbgs so that nothing clashes with the CPU classes:
//   hipbgs::FrameDifferenceBGS, StaticFrameDifferenceBGS, WeightedMovingMeanBGS, WeightedMovingVarianceBGS,
//   AdaptiveBackgroundLearning, AdaptiveSelectiveBackgroundLearning, MixtureOfGaussianV1BGS, MixtureOfGaussianV2BGS, GMG,
//   SigmaDeltaBGS, SuBSENSEBGS, LOBSTERBGS, DPZivkovicAGMMBGS, DPGrimsonGMMBGS, DPWrenGABGS, DPMeanBGS, DPAdaptiveMedianBGS
// Each reads / writes the same ./config/<Class>.xml with the same keys and defaults as the class it stands in for
// (the list is bgs_classes.inc, shared verbatim with this repository's tested host mirror, tracking_amd/host/bgs_host.h).
//
// Use:   mixtureOfGaussianV2BGS = new hipbgs::MixtureOfGaussianV2BGS;      // FrameProcessor.cpp:59-60
//        if(i==36) bgs = new hipbgs::SuBSENSEBGS();                          // ustc_src/ustc_bgs.cpp:68
// Build: -I<repo>/include -I<repo>/tracking_amd/host  -L<repo>/tracking_amd/lib -lbgs_hip
//
// This file needs OpenCV 2.4 headers, which the build image of this repository does not have: it is checked for syntax against
// a declaration-only mock (tests/mock_opencv, test_reference_side_adapters_compile in tests/test_capi_cpu.py, -std=gnu++0x like the
// reference's CMakeLists.txt:5); the logic it shares with bgs_host.h is what the tests exercise.
#pragma once
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include <opencv2/opencv.hpp>

#include "package_bgs/IBGS.h"  // the reference's own interface (package_bgs/IBGS.h:21-33)
#include "bgs_hip.h"

namespace hipbgs {

// the flat CvFileStorage XML of the reference, through OpenCV's own C API (cvReadIntByName on a NULL storage returns the default)
class XmlConfig {
 public:
  XmlConfig() : fs_(0) {}
  ~XmlConfig() { close(); }
  bool load(const std::string& path) {
    close();
    fs_ = cvOpenFileStorage(path.c_str(), 0, CV_STORAGE_READ);
    return fs_ != 0;
  }
  int readInt(const std::string& k, int def) const { return cvReadIntByName(fs_, 0, k.c_str(), def); }
  double readReal(const std::string& k, double def) const { return cvReadRealByName(fs_, 0, k.c_str(), def); }
  void beginWrite() { ints_.clear(), reals_.clear(), order_.clear(); }
  void writeInt(const std::string& k, int v) { order_.push_back(std::make_pair(k, (int)ints_.size())), ints_.push_back(v); }
  void writeReal(const std::string& k, double v) { order_.push_back(std::make_pair(k, -1 - (int)reals_.size())), reals_.push_back(v); }
  bool save(const std::string& path) {
    CvFileStorage* fs = cvOpenFileStorage(path.c_str(), 0, CV_STORAGE_WRITE);
    if (!fs) return false;
    for (size_t i = 0; i < order_.size(); ++i) {
      if (order_[i].second >= 0)
        cvWriteInt(fs, order_[i].first.c_str(), ints_[order_[i].second]);
      else
        cvWriteReal(fs, order_[i].first.c_str(), reals_[-1 - order_[i].second]);
    }
    cvReleaseFileStorage(&fs);
    return true;
  }

 private:
  void close() {
    if (fs_) cvReleaseFileStorage(&fs_);
    fs_ = 0;
  }
  CvFileStorage* fs_;
  std::vector<int> ints_;
  std::vector<double> reals_;
  std::vector<std::pair<std::string, int> > order_;
};

// Common machinery: one single-stream engine, per-frame config reload, the reference's output conventions
// (outputs left untouched on warm-up frames and for classes that never write a background).
class HipBGSBase : public IBGS {
 public:
  virtual ~HipBGSBase() {
    if (engine_) bgs_destroy(engine_);
  }
  void process(const cv::Mat& img_input, cv::Mat& img_output, cv::Mat& img_bgmodel) {
    if (img_input.empty()) return;  // first line of every reference process()
    loadConfig();
    if (firstTime) saveConfig();
    if (!engine_) {
      if (bgs_create(algo_, &params_, device_, 1, &engine_)) fail();
    } else if (bgs_set_params(engine_, &params_)) {
      fail();
    }
    const int bg_ch = (algo_ == BGS_ASBL) ? 1 : img_input.channels();
    fg_.create(img_input.size(), CV_8UC1);
    bg_.create(img_input.size(), CV_MAKETYPE(CV_8U, bg_ch));
    uint32_t flags = 0;
    if (bgs_process(engine_, 0, img_input.data, img_input.rows, img_input.cols, img_input.channels(), img_input.step, fg_.data, fg_.step, bg_.data, bg_.step, &flags))
      fail();
    if (flags & BGS_FG_VALID) fg_.copyTo(img_output);  // img_foreground.copyTo(img_output)
    if (flags & BGS_BG_VALID)
      bg_.copyTo(img_bgmodel);  // img_background.copyTo(img_bgmodel)
    else if (clears_bg_)
      img_bgmodel.release();    // MixtureOfGaussianV1BGS.cpp:68: copyTo of an empty Mat
    firstTime = false;
  }
  void setDevice(int d) { device_ = d; }  // which HIP device the lazily created engine uses (default 0)
  // N2 blob hand-off: connected components of the mask the last process() call produced, found on the device copy of that
  // mask (bgs_last_mask_blobs); components smaller than min_w x min_h are dropped.  Returns how many there are.
  int lastMaskBlobs(int connectivity, int min_w, int min_h, std::vector<bgs_box>& boxes, std::vector<bgs_moments>& moments) {
    if (!engine_) fail();
    int32_t n = 0;
    boxes.resize(256), moments.resize(256);
    for (int pass = 0; pass < 2; ++pass) {  // second pass only if the first buffer was too small
      if (bgs_last_mask_blobs(engine_, 0, connectivity, min_w, min_h, &boxes[0], &moments[0], (int)boxes.size(), &n)) fail();
      if (n <= (int)boxes.size()) break;
      boxes.resize(n), moments.resize(n);
    }
    boxes.resize(n), moments.resize(n);
    return n;
  }

 protected:
  HipBGSBase(bgs_algo algo, const char* name, bool clears_bg = false) : firstTime(true), algo_(algo), name_(name), clears_bg_(clears_bg), device_(0), engine_(0) {
    params_ = bgs_params();
    params_.struct_size = sizeof(params_);
    bgs_default_params(algo, &params_);
    std::cout << name_ << "()" << std::endl;
  }
  std::string configPath() const { return std::string("./config/") + name_ + ".xml"; }
  bool firstTime;
  bgs_params params_;

 private:
  virtual void saveConfig() = 0;  // private pure virtuals of IBGS, re-declared so process() above may call them
  virtual void loadConfig() = 0;
  void fail() { CV_Error(CV_StsError, std::string(name_) + ": " + bgs_last_error()); }  // -> cv::Exception, caught at Main.cpp:63-72
  bgs_algo algo_;
  const char* name_;
  bool clears_bg_;
  int device_;
  bgs_engine* engine_;
  cv::Mat fg_, bg_;
};

#define BGS_HIP_BANNER_DTOR(Class) \
  ~Class() { std::cout << "~" #Class "()" << std::endl; }
#ifndef override
#define BGS_HIP_DEFINED_OVERRIDE
#define override  /* the reference builds as C++03 */
#endif
#include "bgs_classes.inc"
#ifdef BGS_HIP_DEFINED_OVERRIDE
#undef override
#undef BGS_HIP_DEFINED_OVERRIDE
#endif
#undef BGS_HIP_BANNER_DTOR

}  // namespace hipbgs
