// PreProcessor.h — host-side mirror of the two steps in front of the path:
//   bgslibrary::PreProcessor   PreProcessor.{h,cpp}: process() = copy + optional cv::equalizeHist + optional cv::GaussianBlur(7x7, 1.5),
//                              ./config/PreProcessor.xml (equalizeHist, gaussianBlur, enableShow), re-read on every call (:46-77, :128-150)
//   FramePrep                  what VideoCapture::start does to a captured frame before FrameProcessor sees it (VideoCapture.cpp:158-207:
//                              cvResize to input_resize_percent, cvFlip(mode 0) when enableFlip, the ROI view), keys of ./config/VideoCapture.xml
// Both go through bgs_ingest_host (include/bgs_hip.h): the arithmetic runs on the device; with nothing enabled they are plain copies.
// getGrayScale / rotate / applyCanny of the reference's PreProcessor are not on the path (no caller in the tree uses their results).
#pragma once
#include "bgs_host.h"

namespace bgs_hip {

class PreProcessor {
 public:
  PreProcessor() : firstTime(true), equalizeHist(false), gaussianBlur(false), enableShow(true) { std::cout << "PreProcessor()" << std::endl; }
  ~PreProcessor() { std::cout << "~PreProcessor()" << std::endl; }
  void setEqualizeHist(bool value) { equalizeHist = value; }
  void setGaussianBlur(bool value) { gaussianBlur = value; }

  void process(const Image& img_input, Image& img_output) {  // PreProcessor.cpp:46-77
    if (img_input.empty()) return;
    loadConfig();
    if (firstTime) saveConfig();
    if (!equalizeHist && !gaussianBlur) {
      img_input.copyTo(img_output);  // :56
    } else {
      bgs_ingest cfg;
      bgs_ingest_default(&cfg);
      cfg.equalize_hist = equalizeHist, cfg.gaussian_blur = gaussianBlur;
      img_output.create(img_input.rows, img_input.cols, img_input.channels());
      const int rc = bgs_ingest_host(0, &cfg, img_input.data, img_input.rows, img_input.cols, img_input.channels(), img_input.step, img_output.data, img_output.step);
      if (rc) throw Exception(rc, std::string("PreProcessor: ") + bgs_last_error());  // e.g. equalizeHist of a BGR frame: cv::equalizeHist's CV_Assert
    }
    firstTime = false;
  }

 private:
  bool firstTime, equalizeHist, gaussianBlur, enableShow;
  void saveConfig() {  // :128-137
    XmlConfig fs;
    fs.beginWrite();
    fs.writeInt("equalizeHist", equalizeHist);
    fs.writeInt("gaussianBlur", gaussianBlur);
    fs.writeInt("enableShow", enableShow);
    fs.save("./config/PreProcessor.xml");
  }
  void loadConfig() {  // :139-148
    XmlConfig fs;
    fs.load("./config/PreProcessor.xml");
    equalizeHist = fs.readInt("equalizeHist", false);
    gaussianBlur = fs.readInt("gaussianBlur", false);
    enableShow = fs.readInt("enableShow", true);
  }
};

// the frame preparation of VideoCapture::start (the capture, GUI and ROI-picking parts of that class are out of scope)
class FramePrep {
 public:
  FramePrep() : input_resize_percent(100), enableFlip(false), use_roi(false), roi_defined(false), roi_x0(0), roi_y0(0), roi_x1(0), roi_y1(0) { loadConfig(); }
  int input_resize_percent;
  bool enableFlip, use_roi, roi_defined;
  int roi_x0, roi_y0, roi_x1, roi_y1;

  void loadConfig() {  // VideoCapture.cpp:266-283
    XmlConfig fs;
    fs.load("./config/VideoCapture.xml");
    input_resize_percent = fs.readInt("input_resize_percent", 100);
    enableFlip = fs.readInt("enableFlip", false);
    use_roi = fs.readInt("use_roi", false), roi_defined = fs.readInt("roi_defined", false);
    roi_x0 = fs.readInt("roi_x0", 0), roi_y0 = fs.readInt("roi_y0", 0), roi_x1 = fs.readInt("roi_x1", 0), roi_y1 = fs.readInt("roi_y1", 0);
  }
  bgs_ingest config() const {
    bgs_ingest cfg;
    bgs_ingest_default(&cfg);
    cfg.resize_percent = input_resize_percent, cfg.flip = enableFlip;
    if (use_roi && roi_defined) cfg.roi_x0 = roi_x0, cfg.roi_y0 = roi_y0, cfg.roi_x1 = roi_x1, cfg.roi_y1 = roi_y1;  // :197-201
    return cfg;
  }
  // frame1 -> img_input (VideoCapture.cpp:164-203)
  void process(const Image& frame1, Image& img_input) const {
    const bgs_ingest cfg = config();
    if (cfg.resize_percent == 100 && !cfg.flip && !(cfg.roi_x1 > cfg.roi_x0 && cfg.roi_y1 > cfg.roi_y0)) {
      frame1.copyTo(img_input);  // cvResize to the same size is the identity
      return;
    }
    int rows = 0, cols = 0;
    int rc = bgs_ingest_size(&cfg, frame1.rows, frame1.cols, &rows, &cols);
    if (!rc) {
      img_input.create(rows, cols, frame1.channels());
      rc = bgs_ingest_host(0, &cfg, frame1.data, frame1.rows, frame1.cols, frame1.channels(), frame1.step, img_input.data, img_input.step);
    }
    if (rc) throw Exception(rc, std::string("FramePrep: ") + bgs_last_error());
  }
};

}  // namespace bgs_hip
