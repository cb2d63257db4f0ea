// The reference's plugin interface (package_bgs/IBGS.h:21-33) restated for the syntax check only, see ../README.md: the adapters
// include the real one when they live in the reference tree.
#pragma once
#include <opencv2/opencv.hpp>
class IBGS {
 public:
  virtual void process(const cv::Mat& img_input, cv::Mat& img_foreground, cv::Mat& img_background) = 0;
  virtual ~IBGS() {}

 private:
  virtual void saveConfig() = 0;
  virtual void loadConfig() = 0;
};
