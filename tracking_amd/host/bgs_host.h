// bgs_host.h — host-side C++ mirror of the reference's plugin surface, above the C ABI (include/bgs_hip.h).
//
// Mirrors, name for name:
//   class IBGS                      package_bgs/IBGS.h:21-33          process(in, fg, bg); private saveConfig/loadConfig
//   FrameDifferenceBGS ...          package_bgs/<Class>.{h,cpp}        same class names, same ./config/<Class>.xml keys,
//                                                                      loadConfig() on every process(), saveConfig() on the first
// so a caller written against the reference (FrameProcessor.cpp:157-167, Demo.cpp:179, ustc_src/ustc_bgs.cpp:94) changes
// one `new` expression.  The image type is bgs_hip::Image — the four things IBGS::process reads from a cv::Mat
// (data, rows, cols, channels, step) plus copyTo()-style reallocation; the cv::Mat overloads a maintainer adds where
// OpenCV exists are in INTEGRATION.md (OpenCV is not in this image, so they are not compiled here).
//
// Behaviour kept from the reference:
//   * empty input  -> silent return, outputs untouched
//   * warm-up frames / classes that never write a background -> that output is left untouched
//   * hard failures -> exception derived from std::exception (the reference: CV_Assert -> cv::Exception), nothing else
//   * imshow side effects are dropped; the showOutput keys are still read and written so the XML files stay compatible
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bgs_hip.h"

namespace bgs_hip {

// ---------------------------------------------------------------------------------------------- Image
// Owning or viewing 8-bit interleaved image; the subset of cv::Mat that IBGS::process touches.
class Image {
 public:
  uint8_t* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;

  Image() {}
  Image(int r, int c, int ch) { create(r, c, ch); }
  // non-owning view (what `cv::Mat img_input(frame)` is for an IplImage, VideoCapture.cpp:209)
  Image(int r, int c, int ch, uint8_t* p, size_t step_bytes) : data(p), rows(r), cols(c), step(step_bytes), ch_(ch) {}
  int channels() const { return ch_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step == (size_t)cols * ch_; }
  void create(int r, int c, int ch) {
    if (owned_.size() == (size_t)r * c * ch && rows == r && cols == c && ch_ == ch && data == owned_.data()) return;
    owned_.assign((size_t)r * c * ch, 0);
    data = owned_.data(), rows = r, cols = c, ch_ = ch, step = (size_t)c * ch;
  }
  void copyTo(Image& dst) const {
    if (empty()) {
      dst = Image();
      return;
    }
    dst.create(rows, cols, ch_);
    for (int y = 0; y < rows; ++y) std::memcpy(dst.data + (size_t)y * dst.step, data + (size_t)y * step, (size_t)cols * ch_);
  }
  uint8_t* ptr(int y) { return data + (size_t)y * step; }
  const uint8_t* ptr(int y) const { return data + (size_t)y * step; }

 private:
  int ch_ = 0;
  std::vector<uint8_t> owned_;
};

// ---------------------------------------------------------------------------------------------- XML config
// Flat CvFileStorage XML as the reference reads/writes it with cvReadIntByName / cvWriteInt etc.:
//   <?xml version="1.0"?>\n<opencv_storage>\n<key>value</key>\n...</opencv_storage>
// Missing file or key -> the in-code default, exactly like cvRead*ByName(fs, 0, key, default) on a NULL storage.
class XmlConfig {
 public:
  bool load(const std::string& path) {
    kv_.clear();
    std::ifstream f(path.c_str());
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string t = ss.str();
    size_t pos = 0;
    while ((pos = t.find('<', pos)) != std::string::npos) {
      size_t end = t.find('>', pos);
      if (end == std::string::npos) break;
      std::string tag = t.substr(pos + 1, end - pos - 1);
      pos = end + 1;
      if (tag.empty() || tag[0] == '?' || tag[0] == '/' || tag[0] == '!' || tag == "opencv_storage") continue;
      const std::string close = "</" + tag + ">";
      size_t c = t.find(close, pos);
      if (c == std::string::npos) continue;
      std::string val = t.substr(pos, c - pos);
      size_t a = val.find_first_not_of(" \t\r\n"), b = val.find_last_not_of(" \t\r\n");
      val = (a == std::string::npos) ? "" : val.substr(a, b - a + 1);
      if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
      kv_[tag] = val;
      pos = c + close.size();
    }
    return true;
  }
  int readInt(const std::string& k, int def) const {
    auto it = kv_.find(k);
    if (it == kv_.end() || it->second.empty()) return def;
    return (int)std::strtod(it->second.c_str(), nullptr);  // cvReadInt rounds a real node; integers pass through
  }
  double readReal(const std::string& k, double def) const {
    auto it = kv_.find(k);
    if (it == kv_.end() || it->second.empty()) return def;
    return std::strtod(it->second.c_str(), nullptr);
  }
  std::string readString(const std::string& k, const std::string& def) const {
    auto it = kv_.find(k);
    return it == kv_.end() ? def : it->second;
  }

  // writer: keys in insertion order, like consecutive cvWrite* calls
  void beginWrite() { out_.clear(); }
  void writeInt(const std::string& k, int v) { out_.push_back(k + ">" + std::to_string(v)); }
  void writeReal(const std::string& k, double v) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.16g", v);
    std::string s = buf;
    if (s.find_first_of(".eE") == std::string::npos) s += ".";  // CvFileStorage writes reals with a decimal point
    out_.push_back(k + ">" + s);
  }
  void writeString(const std::string& k, const std::string& v) { out_.push_back(k + ">\"" + v + "\""); }
  bool save(const std::string& path) const {
    std::ofstream f(path.c_str());
    if (!f) return false;  // the reference's cvOpenFileStorage also fails silently when ./config is missing
    f << "<?xml version=\"1.0\"?>\n<opencv_storage>\n";
    for (const std::string& e : out_) {
      const size_t gt = e.find('>');
      f << "<" << e.substr(0, gt) << ">" << e.substr(gt + 1) << "</" << e.substr(0, gt) << ">\n";
    }
    f << "</opencv_storage>\n";
    return true;
  }

 private:
  std::map<std::string, std::string> kv_;
  std::vector<std::string> out_;
};

// ---------------------------------------------------------------------------------------------- errors
// What a failed CV_Assert is in the reference: an exception derived from std::exception, caught at Main.cpp:63-72.
class Exception : public std::runtime_error {
 public:
  Exception(int code_, const std::string& what_) : std::runtime_error(what_), code(code_) {}
  int code;
};

// ---------------------------------------------------------------------------------------------- IBGS
class IBGS {  // package_bgs/IBGS.h:21-33
 public:
  virtual void process(const Image& img_input, Image& img_foreground, Image& img_background) = 0;
  virtual ~IBGS() {}

 private:
  virtual void saveConfig() = 0;
  virtual void loadConfig() = 0;
};

// Common machinery of every class below: one single-stream engine, per-frame config reload, output conventions.
class HipBGSBase : public IBGS {
 public:
  ~HipBGSBase() override {
    if (engine_) bgs_destroy(engine_);
  }
  void process(const Image& img_input, Image& img_output, Image& img_bgmodel) override {
    if (img_input.empty()) return;  // first line of every reference process()
    if (group_) {  // this frame already went through the group's one fused launch (FrameProcessor::process): hand the results over
      if (!group_ready_) throw Exception(BGS_ERR_STATE, std::string(name_) + ": grouped with other classes - FrameProcessor::process runs the group first");
      group_ready_ = false;
      deliver(group_flags_, img_output, img_bgmodel);
      return;
    }
    loadConfig();
    if (firstTime) saveConfig();
    if (!engine_) {
      int rc = bgs_create(algo_, &params_, device_, 1, &engine_);
      if (rc) throw Exception(rc, std::string(name_) + ": " + bgs_last_error());
      // fg_ / bg_ below are this object's own images: they live as long as the engine and keep their storage while the frame size
      // stays, so the DMA engine may write them in place (bit 1 mask, bit 2 background).  The caller's input is staged: this class
      // cannot know how long img_input's buffer lives.
      (void)bgs_set_option(engine_, BGS_OPT_HOST_REGISTER, 6);
    } else {
      int rc = bgs_set_params(engine_, &params_);
      if (rc) throw Exception(rc, std::string(name_) + ": " + bgs_last_error());
    }
    const int bg_ch = (algo_ == BGS_ASBL) ? 1 : img_input.channels();
    fg_.create(img_input.rows, img_input.cols, 1);
    bg_.create(img_input.rows, img_input.cols, bg_ch);
    uint32_t flags = 0;
    int rc = bgs_process(engine_, 0, img_input.data, img_input.rows, img_input.cols, img_input.channels(), img_input.step, fg_.data, fg_.step, bg_.data,
                         bg_.step, &flags);
    if (rc) throw Exception(rc, std::string(name_) + ": " + bgs_last_error());
    deliver(flags, img_output, img_bgmodel);
  }
  // ---- several classes on one frame (bgs_group, include/bgs_hip.h): FrameProcessor attaches the byte-stream classes it enables to
  // one group; per frame it calls groupPrepare() on each (the per-frame XML reload + parameter hand-over every process() starts
  // with), runs the group's single launch into the buffers groupBuffers() names, and the class's own process() call then only
  // delivers what that launch produced - same outputs, same conventions, one read of the frame.
  bgs_algo algo() const { return algo_; }
  void attachGroup(bgs_group* g, int index) { group_ = g, group_index_ = index; }
  void groupPrepare(const Image& img_input) {
    loadConfig();
    if (firstTime) saveConfig();
    int rc = bgs_group_set_params(group_, group_index_, &params_);
    if (rc) throw Exception(rc, std::string(name_) + ": " + bgs_last_error());
    fg_.create(img_input.rows, img_input.cols, 1);
    bg_.create(img_input.rows, img_input.cols, img_input.channels());
  }
  void groupBuffers(uint8_t** fg, size_t* fg_step, uint8_t** bg, size_t* bg_step) { *fg = fg_.data, *fg_step = fg_.step, *bg = bg_.data, *bg_step = bg_.step; }
  void groupDone(uint32_t flags) { group_flags_ = flags, group_ready_ = true; }
  // which HIP device the lazily created engine uses (default 0); the reference has no such notion
  void setDevice(int d) { device_ = d; }
  // N2 blob hand-off: connected components of the mask the last process() call produced, found on the device copy of that
  // mask (bgs_last_mask_blobs); components smaller than min_w x min_h are dropped.  Returns how many there are.
  int lastMaskBlobs(int connectivity, int min_w, int min_h, std::vector<bgs_box>& boxes, std::vector<bgs_moments>& moments) {
    if (!engine_) throw Exception(BGS_ERR_STATE, std::string(name_) + ": no frame processed yet");
    int32_t n = 0;
    boxes.resize(256), moments.resize(256);
    for (int pass = 0; pass < 2; ++pass) {  // second pass only if the first buffer was too small
      int rc = bgs_last_mask_blobs(engine_, 0, connectivity, min_w, min_h, boxes.data(), moments.data(), (int)boxes.size(), &n);
      if (rc) throw Exception(rc, std::string(name_) + ": " + bgs_last_error());
      if (n <= (int)boxes.size()) break;
      boxes.resize(n), moments.resize(n);
    }
    boxes.resize(n), moments.resize(n);
    return n;
  }

 protected:
  HipBGSBase(bgs_algo algo, const char* name, bool clears_bg = false) : firstTime(true), algo_(algo), name_(name), clears_bg_(clears_bg) {
    std::memset(&params_, 0, sizeof(params_));
    params_.struct_size = sizeof(params_);
    bgs_default_params(algo, &params_);
    std::cout << name_ << "()" << std::endl;  // the reference's ctor banner
  }
  std::string configPath() const { return std::string("./config/") + name_ + ".xml"; }
  bool firstTime;
  bgs_params params_;

 private:
  void saveConfig() override = 0;  // private pure virtuals of IBGS, re-declared so process() above may call them
  void loadConfig() override = 0;
  bgs_algo algo_;
  const char* name_;
  bool clears_bg_;
  int device_ = 0;
  bgs_engine* engine_ = nullptr;
  Image fg_, bg_;
  bgs_group* group_ = nullptr;  // not owned
  int group_index_ = -1;
  bool group_ready_ = false;
  uint32_t group_flags_ = 0;
  void deliver(uint32_t flags, Image& img_output, Image& img_bgmodel) {
    if (flags & BGS_FG_VALID) fg_.copyTo(img_output);  // img_foreground.copyTo(img_output)
    if (flags & BGS_BG_VALID)
      bg_.copyTo(img_bgmodel);  // img_background.copyTo(img_bgmodel)
    else if (clears_bg_)
      img_bgmodel = Image();    // MixtureOfGaussianV1BGS.cpp:68: copyTo of an empty Mat releases the destination
    firstTime = false;
  }
};

#define BGS_HIP_BANNER_DTOR(Class) \
  ~Class() override { std::cout << "~" #Class "()" << std::endl; }

#include "bgs_classes.inc"

#undef BGS_HIP_BANNER_DTOR

}  // namespace bgs_hip
