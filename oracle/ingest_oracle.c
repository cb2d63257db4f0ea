/*
 * ingest_oracle.c — CPU restatement of the frame preparation in front of the package_bgs hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see bgs_oracle.h).
 *
 * What the reference does to a captured frame before any IBGS::process sees it:
 *   VideoCapture::start   VideoCapture.cpp:158-207   cvResize(frame1, frame) to (w*pct/100, h*pct/100); cvFlip(frame, frame, 0) when
 *                                                   enableFlip; cvSetImageROI(frame, rect(x0, y0, x1-x0, y1-y0)) when a ROI is defined
 *   PreProcessor::process PreProcessor.cpp:46-77     copy; cv::equalizeHist when equalizeHist (asserts 8UC1); cv::GaussianBlur(7x7, 1.5)
 *                                                   when gaussianBlur
 *
 * PARITY STATUS: flip and ROI crop are exact by definition.  cv::resize, cv::equalizeHist and cv::GaussianBlur are OpenCV (not in
 * /root/reference, not in the image): their 8-bit fixed-point arithmetic is RECALLED from OpenCV 2.4.x and is UNPINNED:
 *   R1 cv::resize INTER_LINEAR 8U (imgwarp.cpp): scale = src/dst (double); fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx;
 *      sx < 0 -> (0, fx 0); sx >= w-1 -> (w-1, fx 0); coefficients saturate_cast<short>(c * 2048) of (1-fx, fx); horizontal pass
 *      S[sx]*a0 + S[sx+1]*a1 in int; rows sy, sy+1 clipped to the image; vertical pass
 *      uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).
 *      Exactly 2x down in both directions is switched to INTER_AREA's fast path: (a + b + c + d + 2) >> 2 of the 2x2 block.
 *   R2 cv::equalizeHist (histogram.cpp, 2.4.4+): i = first non-empty bin; all pixels there -> constant image i; scale = 255.f/(total - hist[i]);
 *      lut[i] = 0; lut[j] = saturate_cast<uchar>(sum_{i<k<=j} hist[k] * scale)   (float product, cvRound = half-to-even)
 *   R3 cv::GaussianBlur(7x7, sigma 1.5) 8U (smooth.cpp + filter.cpp): float kernel cf[i] = (float)exp(-0.5 x^2 / sigma^2), normalised by the
 *      double sum of the floats, cf[i] = (float)(cf[i] * (1/sum)); smooth + symmetrical 8U -> 8U runs in fixed point: integer kernel
 *      cvRound(cf[i] * 256.f) for rows and columns, row pass exact in int, column pass (sum + (1 << 15)) >> 16 saturated;
 *      border BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba).
 */
#include "bgs_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static int sat_short(float v) {
  long r = lrintf(v);
  return (int)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}
static uint8_t sat_u8_float(float v) {
  long r = lrintf(v);
  return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

void orc_ingest_size(const bgs_ingest* c, int src_rows, int src_cols, int* rows, int* cols) {
  int w = (int)((src_cols * c->resize_percent) / 100), h = (int)((src_rows * c->resize_percent) / 100); /* VideoCapture.cpp:142 */
  if (c->roi_x1 > c->roi_x0 && c->roi_y1 > c->roi_y0) w = c->roi_x1 - c->roi_x0, h = c->roi_y1 - c->roi_y0;
  *rows = h, *cols = w;
}

/* R1 */
void orc_resize_linear_u8(const uint8_t* src, int srows, int scols, int ch, size_t sstep, uint8_t* dst, int drows, int dcols) {
  const double inv_sx = (double)dcols / scols, inv_sy = (double)drows / srows;
  const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
  const int area_fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
  if (area_fast && isx == 2 && isy == 2) {
    for (int y = 0; y < drows; ++y)
      for (int x = 0; x < dcols; ++x)
        for (int k = 0; k < ch; ++k) {
          const uint8_t* s = src + (size_t)(2 * y) * sstep + (size_t)(2 * x) * ch + k;
          dst[((size_t)y * dcols + x) * ch + k] = (uint8_t)((s[0] + s[ch] + s[sstep] + s[sstep + ch] + 2) >> 2);
        }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * dcols);
  int* ia = (int*)malloc(sizeof(int) * dcols * 2);
  for (int dx = 0; dx < dcols; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= scols - 1) fx = 0, sx = scols - 1;
    xofs[dx] = sx;
    ia[2 * dx] = sat_short((1.f - fx) * 2048.f), ia[2 * dx + 1] = sat_short(fx * 2048.f);
  }
  int* rowbuf[2];
  rowbuf[0] = (int*)malloc(sizeof(int) * dcols * ch), rowbuf[1] = (int*)malloc(sizeof(int) * dcols * ch);
  for (int dy = 0; dy < drows; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
    for (int k = 0; k < 2; ++k) {
      const uint8_t* S = src + (size_t)clampi(sy + k, 0, srows - 1) * sstep;
      for (int dx = 0; dx < dcols; ++dx)
        for (int c = 0; c < ch; ++c) {
          const int sx = xofs[dx], sx1 = sx + 1 < scols ? sx + 1 : sx; /* the coefficient of a clamped neighbour is 0 */
          rowbuf[k][dx * ch + c] = S[sx * ch + c] * ia[2 * dx] + S[sx1 * ch + c] * ia[2 * dx + 1];
        }
    }
    for (int i = 0; i < dcols * ch; ++i)
      dst[(size_t)dy * dcols * ch + i] = (uint8_t)((((b0 * (rowbuf[0][i] >> 4)) >> 16) + ((b1 * (rowbuf[1][i] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs), free(ia), free(rowbuf[0]), free(rowbuf[1]);
}

/* R2, in place on a contiguous 1-channel image */
void orc_equalize_hist_u8(uint8_t* img, size_t total) {
  int hist[256] = {0};
  for (size_t i = 0; i < total; ++i) hist[img[i]]++;
  int i = 0;
  while (!hist[i]) ++i;
  if ((size_t)hist[i] == total) {
    memset(img, i, total);
    return;
  }
  const float scale = (256 - 1.f) / (float)(total - (size_t)hist[i]);
  int sum = 0;
  uint8_t lut[256];
  memset(lut, 0, sizeof(lut));
  for (lut[i++] = 0; i < 256; ++i) {
    sum += hist[i];
    lut[i] = sat_u8_float((float)sum * scale);
  }
  for (size_t k = 0; k < total; ++k) img[k] = lut[img[k]];
}

/* R3: the integer kernel; returns 0 if OpenCV's getKernelType would NOT call the float kernel smooth (then the fixed-point path
 * would not be taken and this restatement would not apply) */
int orc_gaussian7_kernel(int ik[7]) {
  float cf[7];
  const double sigma = 1.5, scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < 7; ++i) {
    const double x = i - 3.0;
    cf[i] = (float)exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  double check = 0;
  for (int i = 0; i < 7; ++i) {
    cf[i] = (float)(cf[i] * sum);
    check += cf[i];
    ik[i] = (int)lrintf(cf[i] * 256.f);
  }
  return fabs(check - 1) <= FLT_EPSILON * (fabs(check) + 1);
}

static int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

/* contiguous src -> contiguous dst (may not alias) */
void orc_gaussian_blur7_u8(const uint8_t* src, uint8_t* dst, int rows, int cols, int ch) {
  int ik[7];
  (void)orc_gaussian7_kernel(ik);
  int* tmp = (int*)malloc(sizeof(int) * (size_t)rows * cols * ch);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
      for (int c = 0; c < ch; ++c) {
        int s = 0;
        for (int j = -3; j <= 3; ++j) s += ik[j + 3] * src[((size_t)y * cols + reflect101(x + j, cols)) * ch + c];
        tmp[((size_t)y * cols + x) * ch + c] = s;
      }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
      for (int c = 0; c < ch; ++c) {
        int s = 0;
        for (int j = -3; j <= 3; ++j) s += ik[j + 3] * tmp[((size_t)reflect101(y + j, rows) * cols + x) * ch + c];
        const int v = (s + (1 << 15)) >> 16;
        dst[((size_t)y * cols + x) * ch + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
  free(tmp);
}

/* the whole chain; dst is contiguous [rows][cols][ch] of orc_ingest_size.  Returns 0, or -1 for a configuration the reference
 * itself fails on (equalizeHist of a 3-channel frame: cv::equalizeHist asserts CV_8UC1; ROI outside the resized frame). */
int orc_ingest(const bgs_ingest* c, const uint8_t* src, int src_rows, int src_cols, int ch, size_t src_step, uint8_t* dst) {
  const int rw = (int)((src_cols * c->resize_percent) / 100), rh = (int)((src_rows * c->resize_percent) / 100);
  if (rw < 1 || rh < 1) return -1;
  if (c->equalize_hist && ch != 1) return -1;
  int rows, cols;
  orc_ingest_size(c, src_rows, src_cols, &rows, &cols);
  const int roi = c->roi_x1 > c->roi_x0 && c->roi_y1 > c->roi_y0;
  const int x0 = roi ? c->roi_x0 : 0, y0 = roi ? c->roi_y0 : 0;
  if (x0 < 0 || y0 < 0 || x0 + cols > rw || y0 + rows > rh) return -1;
  uint8_t* resized = (uint8_t*)malloc((size_t)rw * rh * ch);
  orc_resize_linear_u8(src, src_rows, src_cols, ch, src_step, resized, rh, rw); /* cvResize(frame1, frame) */
  uint8_t* cur = (uint8_t*)malloc((size_t)rows * cols * ch);
  for (int y = 0; y < rows; ++y) {
    const int Y = y + y0, Ys = c->flip ? rh - 1 - Y : Y; /* cvFlip(frame, frame, 0), then the ROI view */
    memcpy(cur + (size_t)y * cols * ch, resized + ((size_t)Ys * rw + x0) * ch, (size_t)cols * ch);
  }
  free(resized);
  if (c->equalize_hist) orc_equalize_hist_u8(cur, (size_t)rows * cols);
  if (c->gaussian_blur)
    orc_gaussian_blur7_u8(cur, dst, rows, cols, ch);
  else
    memcpy(dst, cur, (size_t)rows * cols * ch);
  free(cur);
  return 0;
}
