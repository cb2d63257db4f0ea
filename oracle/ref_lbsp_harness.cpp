// ref_lbsp_harness.cpp — TEST INFRASTRUCTURE.  Compiles the REFERENCE's own LBSP pattern fragments
// (package_bgs/pl/LBSP_16bits_dbcross_{3ch3t,3ch1t,s3ch,1ch}.i) from where they lie under /root/reference
// (include path given by oracle/Makefile; nothing is copied) into oracle/_ref/libref_lbsp.so.
// The fragments include no header; they document the names that "must be defined externally"
// (_t, _ref, _data, _y, _x, _step_row, _res, L1dist) — exactly what LBSP.h:50-95 provides around them,
// and what this harness provides with plain C types.  L1dist follows DistanceUtils.h:6-8.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

static inline size_t L1dist(unsigned char a, unsigned char b) { return (size_t)abs((int)a - (int)b); }

extern "C" {

// LBSP::computeRGBDescriptor(img, ref[3], x, y, t[3], res[3])   LBSP.h:62-71
void ref_lbsp_3ch3t(const unsigned char* _data, size_t _step_row, int _x, int _y, const unsigned char* _ref, const size_t* _t, unsigned short* _res) {
#include "LBSP_16bits_dbcross_3ch3t.i"
}
// LBSP::computeRGBDescriptor(img, ref[3], x, y, t, res[3])      LBSP.h:74-83
void ref_lbsp_3ch1t(const unsigned char* _data, size_t _step_row, int _x, int _y, const unsigned char* _ref, size_t _t, unsigned short* _res) {
#include "LBSP_16bits_dbcross_3ch1t.i"
}
// LBSP::computeSingleRGBDescriptor(img, ref, x, y, c, t, res)   LBSP.h:86-95
unsigned short ref_lbsp_s3ch(const unsigned char* _data, size_t _step_row, int _x, int _y, size_t _c, unsigned char _ref, size_t _t) {
  unsigned short _res;
#include "LBSP_16bits_dbcross_s3ch.i"
  return _res;
}
// LBSP::computeGrayscaleDescriptor(img, ref, x, y, t, res)      LBSP.h:50-59
unsigned short ref_lbsp_1ch(const unsigned char* _data, size_t _step_row, int _x, int _y, unsigned char _ref, size_t _t) {
  unsigned short _res;
#include "LBSP_16bits_dbcross_1ch.i"
  return _res;
}

// Whole image, intra-frame descriptors the way BackgroundSubtractorSuBSENSE::initialize fills m_oLastDescFrame
// (BackgroundSubtractorSuBSENSE.cpp:211-222 1ch, :229-243 3ch): centre = the pixel itself, t = lut[centre];
// the 2-px border stays 0 (LBSP::validateROI, LBSP.cpp:311-318).
void ref_lbsp_describe(const unsigned char* img, size_t step, int rows, int cols, int channels, const unsigned char* lut, unsigned short* desc) {
  memset(desc, 0, (size_t)rows * cols * channels * sizeof(unsigned short));
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) {
      if (channels == 3) {
        const unsigned char* ref = img + (size_t)y * step + 3 * (size_t)x;
        const size_t t[3] = {lut[ref[0]], lut[ref[1]], lut[ref[2]]};
        ref_lbsp_3ch3t(img, step, x, y, ref, t, desc + ((size_t)y * cols + x) * 3);
      } else {
        const unsigned char ref = img[(size_t)y * step + x];
        desc[(size_t)y * cols + x] = ref_lbsp_1ch(img, step, x, y, ref, lut[ref]);
      }
    }
}
}
