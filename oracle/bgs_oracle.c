/*
 * bgs_oracle.c — CPU restatement of the reference's package_bgs hot path (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY (see bgs_oracle.h).  Build: oracle/Makefile, flags
 * -O2 -ffp-contract=off -fno-fast-math: every float expression below must round
 * exactly where it is written, because the HIP kernels are compared bit-for-bit.
 *
 * PARITY STATUS
 *   pinned   : LBSP descriptors (vs oracle/_ref built from the reference's own .i files)
 *   unpinned : every cv:: primitive (OpenCV 2.4 is not in /root/reference, not in the
 *              image; the semantics below are recalled) — that covers
 *              BGR2GRAY, convertTo, MatExpr folding, addWeighted, MOG2, MOG.
 *
 * Recalled OpenCV 2.4 semantics used throughout (one place, so they can be re-pinned):
 *   P1 cvtColor(BGR2GRAY) u8 : (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14
 *   P2 threshold(BINARY) u8   : src > thr ? 255 : 0
 *   P3 convertTo(32F, 1/255.) : (float)u8 * (float)(1./255.)            [cvtScale_<uchar,float,float>]
 *   P4 convertTo(8U, 255, -0) : saturate_u8(cvRound(f * 255.f))          [cvtScale_<float,uchar,float>; cvRound = half-to-even]
 *   P5 addWeighted f32        : (float)((double)a*alpha + (double)b*beta + gamma)   [addWeighted_<float,double>]
 *   P6 scaleAdd f32           : a*(float)alpha + b   in float
 *   P7 MatExpr  A*a + B*b     : folds to ONE addWeighted(A,a,B,b,0)      [MatOp::add over two unary AddEx]
 *      MatExpr (A*a+B*b)+C*c  : t = addWeighted(A,a,B,b,0);  then AddEx(t,C,1,c) assigns as scaleAdd(C,c,t)  [MatOp_AddEx::assign, alpha==1]
 *      MatExpr (A+B+C)/3.0    : t = add(A,B); AddEx(t,C,1,1)*(1/3.) = addWeighted(t,1/3.,C,1/3.,0)
 *      (SURVEY.md App. A guessed addWeighted for the second step and a float scale for /3.0; the
 *       MatOp_AddEx::assign / ::multiply paths recalled here differ by < 1 ulp and matter only at u8 rounding ties.)
 *   P8 pow(x,2) = x*x ; w*M = M*(float)w in float ; sqrt = sqrtf ; absdiff f32 = fabsf(a-b)
 *   P9 medianBlur k : true median, BORDER_REPLICATE
 */
#include "bgs_oracle.h"
#include "subsense_oracle.h"
#include "dp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_engine {
  bgs_algo algo;
  bgs_params p;
  int threads;
  int rows, cols, ch;
  size_t n;
  int64_t nframes; /* frames consumed */
  uint8_t *cur, *prev1, *prev2;
  int have1, have2;
  uint8_t* bgimg; /* SFD/ABL/ASBL background (u8) */
  int64_t counter;
  /* MOG2 (reference layout: GMM{weight,variance}[N*K] then mean[N*K*C]) */
  float *gmm, *mean;
  uint8_t* modes;
  /* MOG1: MixData{sortKey, weight, mean[C], var[C]}[N*K] */
  float* mix;
  /* SigmaDelta: Mt = bgimg, Vt */
  uint8_t* vt;
  ss_state* ss; /* SuBSENSE (subsense_oracle.c) */
  dp_state* dp; /* package_bgs/dp models (dp_oracle.c) */
  /* GMG: per pixel up to maxFeatures {colour, weight} + count */
  int32_t* gmg_colors;
  float* gmg_weights;
  int32_t* gmg_nfeat;
  /* scratch */
  uint8_t *tmp8a, *tmp8b;
  float* tmpf;
};

/* ---------------------------------------------------------------- primitives */

static inline int cv_round(double v) { return (int)lrint(v); } /* half-to-even in the default rounding mode */
static inline uint8_t sat_u8_i(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline uint8_t sat_u8_f(float v) { return sat_u8_i(cv_round((double)v)); }
static inline uint8_t gray_bgr(int b, int g, int r) { return (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14); } /* P1 */
static inline uint8_t thr_bin(int v, int thr) { return (uint8_t)(v > thr ? 255 : 0); }                                           /* P2 */
static inline float aw(float a, double alpha, float b, double beta) { return (float)((double)a * alpha + (double)b * beta + 0.0); } /* P5 */
static inline float scale_add(float a, double alpha, float b) { return a * (float)alpha + b; }                                     /* P6 */

void orc_bgr2gray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const uint8_t* s = src + (size_t)y * sstep + 3 * (size_t)x;
      dst[(size_t)y * dstep + x] = gray_bgr(s[0], s[1], s[2]);
    }
}

static int cmp_u8(const void* a, const void* b) { return (int)*(const uint8_t*)a - (int)*(const uint8_t*)b; }

void orc_median_blur_u8(const uint8_t* src, uint8_t* dst, int rows, int cols, int ksize) { /* P9 */
  const int r = ksize / 2;
  uint8_t win[15 * 15];
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int m = 0;
      for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
          int yy = y + dy, xx = x + dx;
          yy = yy < 0 ? 0 : yy >= rows ? rows - 1 : yy;
          xx = xx < 0 ? 0 : xx >= cols ? cols - 1 : xx;
          win[m++] = src[(size_t)yy * cols + xx];
        }
      qsort(win, (size_t)m, 1, cmp_u8);
      dst[(size_t)y * cols + x] = win[m / 2];
    }
}

/* 3x3 rectangular erode/dilate; border pixels outside the image do not take part
 * (BORDER_CONSTANT with morphologyDefaultBorderValue = +inf for erode, -inf for dilate). */
static void morph3x3_once(const uint8_t* src, uint8_t* dst, int rows, int cols, int dilate) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int v = dilate ? 0 : 255;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
          int s = src[(size_t)yy * cols + xx];
          v = dilate ? (s > v ? s : v) : (s < v ? s : v);
        }
      dst[(size_t)y * cols + x] = (uint8_t)v;
    }
}
static void morph3x3(const uint8_t* src, uint8_t* dst, int rows, int cols, int iterations, int dilate) {
  size_t n = (size_t)rows * cols;
  uint8_t* a = (uint8_t*)malloc(n);
  uint8_t* b = (uint8_t*)malloc(n);
  memcpy(a, src, n);
  for (int i = 0; i < iterations; ++i) {
    morph3x3_once(a, b, rows, cols, dilate);
    uint8_t* t = a;
    a = b;
    b = t;
  }
  memcpy(dst, a, n);
  free(a);
  free(b);
}
void orc_erode3x3(const uint8_t* src, uint8_t* dst, int rows, int cols, int iterations) { morph3x3(src, dst, rows, cols, iterations, 0); }
void orc_dilate3x3(const uint8_t* src, uint8_t* dst, int rows, int cols, int iterations) { morph3x3(src, dst, rows, cols, iterations, 1); }

/* cv::floodFill(img, Point(0,0), newval) with lo=up=0: 4-connected region of pixels equal to the seed value */
void orc_floodfill_from_origin(uint8_t* img, int rows, int cols, uint8_t newval) {
  const uint8_t seed = img[0];
  if (seed == newval) return;
  size_t n = (size_t)rows * cols, head = 0, tail = 0;
  int32_t* q = (int32_t*)malloc(n * sizeof(int32_t));
  img[0] = newval;
  q[tail++] = 0;
  while (head < tail) {
    int32_t i = q[head++];
    int y = i / cols, x = i % cols;
    const int ny[4] = {y - 1, y + 1, y, y}, nx[4] = {x, x, x - 1, x + 1};
    for (int k = 0; k < 4; ++k) {
      if (ny[k] < 0 || ny[k] >= rows || nx[k] < 0 || nx[k] >= cols) continue;
      size_t j = (size_t)ny[k] * cols + nx[k];
      if (img[j] == seed) {
        img[j] = newval;
        q[tail++] = (int32_t)j;
      }
    }
  }
  free(q);
}

/* ---------------------------------------------------------------- LBSP (a10) */

/* BackgroundSubtractorSuBSENSE.cpp:209-210 (1ch: /3) and :227-228 (3ch) */
void orc_lbsp_lut(float rel_threshold, int offset, int channels, uint8_t lut[256]) {
  for (int t = 0; t < 256; ++t) {
    float v = (float)(size_t)offset + (float)(size_t)t * rel_threshold;
    if (channels == 1) v = v / 3;
    lut[t] = sat_u8_f(v);
  }
}

/* LBSP_16bits_dbcross_3ch3t.i:27-43 / LBSP_16bits_dbcross_1ch.i:26-41: bit 15..0 = offsets (x,y) below */
static const int8_t LBSP_DX[16] = {-1, 1, 1, -1, 1, 0, -1, 0, -2, 2, 2, -2, 0, 0, 2, -2};
static const int8_t LBSP_DY[16] = {1, -1, 1, -1, 0, -1, 0, 1, -2, 2, -2, 2, 2, -2, 0, 0};

void orc_lbsp_describe(const uint8_t* img, size_t step, int rows, int cols, int channels, const uint8_t* t_lut, uint16_t* desc) {
  memset(desc, 0, (size_t)rows * cols * channels * sizeof(uint16_t));
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x)
      for (int c = 0; c < channels; ++c) {
        const int ref = img[(size_t)y * step + (size_t)x * channels + c];
        const int t = t_lut[ref];
        unsigned r = 0;
        for (int b = 0; b < 16; ++b) {
          const int v = img[(size_t)(y + LBSP_DY[b]) * step + (size_t)(x + LBSP_DX[b]) * channels + c];
          r |= (unsigned)(abs(v - ref) > t) << (15 - b);
        }
        desc[((size_t)y * cols + x) * channels + c] = (uint16_t)r;
      }
}

/* ---------------------------------------------------------------- engine plumbing */

/* N1: connected components of a byte mask (the definition the GPU kernels are checked against; the reference's consumer,
 * OpenCV-legacy's CvBlobDetectorCC -> cvFindContours + bounding rects, ustc_src/trackingMain.cpp:56-57, is not in the
 * tree).  Raster scan; every unlabelled non-zero pixel starts a component named by its own raster index (root) and is
 * flooded with an explicit stack, so components come out in increasing root order.  labels: root per pixel, -1 for
 * background (may be NULL).  Returns the number of components; writes the first max_boxes. */
int orc_components(const uint8_t* mask, int rows, int cols, int connectivity, int32_t* labels, bgs_box* boxes, int max_boxes) {
  const size_t n = (size_t)rows * cols;
  int32_t* L = labels ? labels : (int32_t*)malloc(n * sizeof(int32_t));
  int32_t* stack = (int32_t*)malloc(n * sizeof(int32_t));
  for (size_t i = 0; i < n; ++i) L[i] = -1;
  int count = 0;
  for (size_t p = 0; p < n; ++p) {
    if (!mask[p] || L[p] >= 0) continue;
    int minx = cols, miny = rows, maxx = -1, maxy = -1, area = 0;
    size_t top = 0;
    stack[top++] = (int32_t)p;
    L[p] = (int32_t)p;
    while (top) {
      const int32_t q = stack[--top];
      const int y = q / cols, x = q - y * cols;
      if (x < minx) minx = x;
      if (x > maxx) maxx = x;
      if (y < miny) miny = y;
      if (y > maxy) maxy = y;
      area++;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dy && !dx) continue;
          if (connectivity == 4 && dy && dx) continue;
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
          const size_t r = (size_t)yy * cols + xx;
          if (mask[r] && L[r] < 0) {
            L[r] = (int32_t)p;
            stack[top++] = (int32_t)r;
          }
        }
    }
    if (count < max_boxes) {
      bgs_box b = {minx, miny, maxx - minx + 1, maxy - miny + 1, area, (int32_t)p};
      boxes[count] = b;
    }
    count++;
  }
  free(stack);
  if (!labels) free(L);
  return count;
}

int orc_default_params(bgs_algo algo, bgs_params* p) {
  if (!p) return BGS_ERR_INVALID;
  uint32_t sz = p->struct_size;
  memset(p, 0, sizeof(*p));
  p->struct_size = sz ? sz : (uint32_t)sizeof(*p);
  p->enable_threshold = 1;
  p->threshold = (algo == BGS_ASBL) ? 25 : 15; /* AdaptiveSelectiveBackgroundLearning.cpp:127 */
  p->enable_weight = 1;
  p->alpha = 0.05;
  p->limit = -1;
  p->learning_frames = 90;
  p->alpha_learn = 0.05;
  p->alpha_detection = 0.05;
  p->mog2_history = 500;
  p->mog2_nmixtures = 5;
  p->mog2_var_threshold = 16.f;
  p->mog2_background_ratio = 0.9f;
  p->mog2_var_threshold_gen = 9.f;
  p->mog2_var_init = 15.f;
  p->mog2_var_min = 4.f;
  p->mog2_var_max = 75.f; /* 5*var_init */
  p->mog2_ct = 0.05f;
  p->mog2_tau = 0.5f;
  p->mog2_detect_shadows = 1;
  p->mog2_shadow_value = 127;
  p->mog1_history = 200;
  p->mog1_nmixtures = 5;
  p->mog1_background_ratio = 0.7;
  p->mog1_var_threshold = 2.5 * 2.5;
  p->mog1_noise_sigma = 30 * 0.5;
  p->lbsp_rel_threshold = 0.333f;
  p->lbsp_threshold_offset = 0;
  p->subsense_min_color_dist_threshold = 30;
  p->subsense_n_samples = 50;
  p->subsense_n_required = 2;
  p->subsense_samples_for_moving_avgs = 100;
  p->subsense_desc_dist_threshold_offset = 3;
  p->gmg_max_features = 64;
  p->gmg_init_frames = 20;
  p->gmg_quantization_levels = 16;
  p->gmg_smoothing_radius = 7;
  p->gmg_update_background_model = 1;
  p->gmg_learning_rate = 0.025;
  p->gmg_background_prior = 0.8;
  p->gmg_decision_threshold = 0.7;
  p->sd_amp_factor = 1;
  p->sd_min_var = 15;
  p->sd_max_var = 255;
  /* package_bgs/dp wrappers, DP*BGS.cpp:19 */
  p->dp_gaussians = 3;
  p->dp_sampling_rate = 7;
  switch (algo) {
    case BGS_DP_ZIVKOVIC_AGMM: p->dp_threshold = 25.0f, p->dp_alpha = 0.001f; break;
    case BGS_DP_GRIMSON_GMM: p->dp_threshold = 9.0f, p->dp_alpha = 0.01f; break;
    case BGS_DP_WREN_GA: p->dp_threshold = 12.25f, p->dp_alpha = 0.005f, p->learning_frames = 30; break;
    case BGS_DP_MEAN: p->dp_threshold = 2700.0f, p->dp_alpha = 1e-6f, p->learning_frames = 30; break;
    case BGS_DP_ADAPTIVE_MEDIAN: p->dp_threshold = 40.0f, p->learning_frames = 30; break;
    case BGS_LOBSTER: /* BackgroundSubtractorLOBSTER.h:6-16 */
      p->lbsp_rel_threshold = 0.365f, p->subsense_desc_dist_threshold_offset = 4, p->subsense_min_color_dist_threshold = 30;
      p->subsense_n_samples = 35, p->subsense_n_required = 2;
      break;
    default: break;
  }
  return BGS_OK;
}

int orc_create(bgs_algo algo, const bgs_params* params, orc_engine** out) {
  if (!out || (int)algo < 0 || algo >= BGS_ALGO_COUNT) return BGS_ERR_INVALID;
  orc_engine* e = (orc_engine*)calloc(1, sizeof(*e));
  if (!e) return BGS_ERR_NOMEM;
  e->algo = algo;
  e->threads = 1;
  if (params)
    e->p = *params;
  else
    orc_default_params(algo, &e->p);
  *out = e;
  return BGS_OK;
}

int orc_set_params(orc_engine* e, const bgs_params* params) {
  if (!e || !params) return BGS_ERR_INVALID;
  e->p = *params;
  return BGS_OK;
}

int orc_set_threads(orc_engine* e, int n) {
  if (!e || n < 1) return BGS_ERR_INVALID;
  e->threads = n;
  return BGS_OK;
}

void orc_destroy(orc_engine* e) {
  if (!e) return;
  free(e->cur);
  free(e->prev1);
  free(e->prev2);
  free(e->bgimg);
  free(e->gmm);
  free(e->mean);
  free(e->modes);
  free(e->mix);
  free(e->vt);
  ss_destroy(e->ss);
  dp_destroy(e->dp);
  free(e->gmg_colors);
  free(e->gmg_weights);
  free(e->gmg_nfeat);
  free(e->tmp8a);
  free(e->tmp8b);
  free(e->tmpf);
  free(e);
}

static int ensure_geometry(orc_engine* e, int rows, int cols, int ch) {
  if (e->n) return (rows == e->rows && cols == e->cols && ch == e->ch) ? BGS_OK : BGS_ERR_GEOMETRY;
  if (ch != 1 && ch != 3) return BGS_ERR_UNSUPPORTED;
  e->rows = rows;
  e->cols = cols;
  e->ch = ch;
  e->n = (size_t)rows * cols;
  const size_t nb = e->n * ch;
  e->cur = (uint8_t*)malloc(nb);
  e->prev1 = (uint8_t*)malloc(nb);
  e->prev2 = (uint8_t*)malloc(nb);
  e->bgimg = (uint8_t*)malloc(nb);
  e->tmp8a = (uint8_t*)malloc(nb);
  e->tmp8b = (uint8_t*)malloc(nb);
  e->tmpf = (float*)malloc(e->n * sizeof(float));
  if (e->algo == BGS_MOG2) {
    const int K = e->p.mog2_nmixtures;
    e->gmm = (float*)calloc(e->n * K * 2, sizeof(float));
    e->mean = (float*)calloc(e->n * K * ch, sizeof(float));
    e->modes = (uint8_t*)calloc(e->n, 1);
  }
  if (e->algo == BGS_MOG1) {
    const int K = e->p.mog1_nmixtures;
    e->mix = (float*)calloc(e->n * K * (2 + 2 * ch), sizeof(float));
  }
  return BGS_OK;
}

static void write_mask(const orc_engine* e, const uint8_t* m, uint8_t* fg, size_t fg_step) {
  if (!fg) return;
  for (int y = 0; y < e->rows; ++y) memcpy(fg + (size_t)y * fg_step, m + (size_t)y * e->cols, (size_t)e->cols);
}
static void write_img(const orc_engine* e, const uint8_t* img, int ch, uint8_t* bg, size_t bg_step) {
  if (!bg) return;
  for (int y = 0; y < e->rows; ++y) memcpy(bg + (size_t)y * bg_step, img + (size_t)y * e->cols * ch, (size_t)e->cols * ch);
}

/* gray (if 3ch) + optional threshold of a per-channel u8 image: the tail every wrapper shares,
 * e.g. FrameDifferenceBGS.cpp:47-51 */
static void gray_thr(const orc_engine* e, const uint8_t* img, uint8_t* mask) {
  const int thr = e->p.threshold, en = e->p.enable_threshold;
  for (size_t i = 0; i < e->n; ++i) {
    int g = e->ch == 3 ? gray_bgr(img[3 * i], img[3 * i + 1], img[3 * i + 2]) : img[i];
    mask[i] = en ? thr_bin(g, thr) : (uint8_t)g;
  }
}

/* ---------------------------------------------------------------- a1 / a2 */

/* FrameDifferenceBGS.cpp:29-61 */
static uint32_t fd_process(orc_engine* e, uint8_t* fg, size_t fg_step) {
  const size_t nb = e->n * e->ch;
  if (!e->have1) { /* :39-43 first frame: store, return, outputs untouched */
    memcpy(e->prev1, e->cur, nb);
    e->have1 = 1;
    return 0;
  }
  for (size_t i = 0; i < nb; ++i) e->tmp8a[i] = (uint8_t)abs((int)e->prev1[i] - (int)e->cur[i]); /* :45 */
  gray_thr(e, e->tmp8a, e->tmp8b);                                                               /* :47-51 */
  write_mask(e, e->tmp8b, fg, fg_step);                                                          /* :56 */
  memcpy(e->prev1, e->cur, nb);                                                                  /* :58 */
  return BGS_FG_VALID;
}

/* StaticFrameDifferenceBGS.cpp:29-57 */
static uint32_t sfd_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step) {
  const size_t nb = e->n * e->ch;
  if (!e->have1) { /* :34-35 */
    memcpy(e->bgimg, e->cur, nb);
    e->have1 = 1;
  }
  for (size_t i = 0; i < nb; ++i) e->tmp8a[i] = (uint8_t)abs((int)e->cur[i] - (int)e->bgimg[i]); /* :42 */
  gray_thr(e, e->tmp8a, e->tmp8b);
  write_mask(e, e->tmp8b, fg, fg_step);
  write_img(e, e->bgimg, e->ch, bg, bg_step); /* :54 */
  return BGS_FG_VALID | BGS_BG_VALID;
}

/* ---------------------------------------------------------------- a3 / a4 */

static int wm_warmup(orc_engine* e) { /* WeightedMovingMeanBGS.cpp:39-50 == WeightedMovingVarianceBGS.cpp:40-51 */
  const size_t nb = e->n * e->ch;
  if (!e->have1) {
    memcpy(e->prev1, e->cur, nb);
    e->have1 = 1;
    return 1;
  }
  if (!e->have2) {
    memcpy(e->prev2, e->prev1, nb);
    memcpy(e->prev1, e->cur, nb);
    e->have2 = 1;
    return 1;
  }
  return 0;
}
static void wm_shift(orc_engine* e) { /* WeightedMovingMeanBGS.cpp:92-93 */
  const size_t nb = e->n * e->ch;
  memcpy(e->prev2, e->prev1, nb);
  memcpy(e->prev1, e->cur, nb);
}

/* WeightedMovingMeanBGS.cpp:29-96 */
static uint32_t wmm_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step) {
  if (wm_warmup(e)) return 0;
  const size_t nb = e->n * e->ch;
  const float sf = (float)(1. / 255.); /* P3 */
  for (size_t i = 0; i < nb; ++i) {
    const float i0 = (float)e->cur[i] * sf, i1 = (float)e->prev1[i] * sf, i2 = (float)e->prev2[i] * sf; /* :52-59 */
    float bgf;
    if (e->p.enable_weight) {
      const float t = aw(i0, 0.5, i1, 0.3); /* :64, P7 */
      bgf = scale_add(i2, 0.2, t);
    } else {
      const float t = i0 + i1; /* :66, P7 */
      bgf = aw(t, 1. / 3.0, i2, 1. / 3.0);
    }
    const uint8_t b8 = sat_u8_f(bgf * 255.f); /* :72, P4 */
    e->bgimg[i] = b8;
    e->tmp8a[i] = (uint8_t)abs((int)e->cur[i] - (int)b8); /* :78 */
  }
  gray_thr(e, e->tmp8a, e->tmp8b); /* :80-84 */
  write_mask(e, e->tmp8b, fg, fg_step);
  write_img(e, e->bgimg, e->ch, bg, bg_step);
  wm_shift(e);
  return BGS_FG_VALID | BGS_BG_VALID;
}

/* WeightedMovingVarianceBGS.cpp:126-137 */
static inline float wvar(float x, float mean, double w) {
  const float d = fabsf(x - mean); /* :131 */
  const float p = d * d;           /* :133, P8 */
  return p * (float)w;             /* :134, P8 */
}

/* WeightedMovingVarianceBGS.cpp:30-117 */
static uint32_t wmv_process(orc_engine* e, uint8_t* fg, size_t fg_step) {
  if (wm_warmup(e)) return 0;
  const size_t nb = e->n * e->ch;
  const float sf = (float)(1. / 255.);
  const double w0 = e->p.enable_weight ? 0.5 : 0.3, w1 = 0.3, w2 = e->p.enable_weight ? 0.2 : 0.3; /* :68-70, :78-89 */
  for (size_t i = 0; i < nb; ++i) {
    const float i0 = (float)e->cur[i] * sf, i1 = (float)e->prev1[i] * sf, i2 = (float)e->prev2[i] * sf;
    const float t = aw(i0, w0, i1, w1);
    const float m = scale_add(i2, w2, t);
    const float v = (wvar(i0, m, w0) + wvar(i1, m, w1)) + wvar(i2, m, w2); /* :83 */
    const float sd = sqrtf(v);                                            /* :95 */
    e->tmp8a[i] = sat_u8_f(sd * 255.f);                                   /* :99 */
  }
  gray_thr(e, e->tmp8a, e->tmp8b); /* :102-106 */
  write_mask(e, e->tmp8b, fg, fg_step);
  wm_shift(e);
  return BGS_FG_VALID; /* bg never written, :111 */
}

/* ---------------------------------------------------------------- a5 / a6 */

/* AdaptiveBackgroundLearning.cpp:30-83 */
static uint32_t abl_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step) {
  const size_t nb = e->n * e->ch;
  if (!e->have1) { /* :40-41 */
    memcpy(e->bgimg, e->cur, nb);
    e->have1 = 1;
  }
  const float sf = (float)(1. / 255.);
  const double alpha = e->p.alpha, beta = 1 - e->p.alpha;
  const int limit = e->p.limit;
  const int update = (limit > 0 && limit < e->counter) || limit == -1; /* :52 */
  for (size_t i = 0; i < nb; ++i) {
    const float i_f = (float)e->cur[i] * sf, b_f = (float)e->bgimg[i] * sf; /* :43-47 */
    const float d = fabsf(i_f - b_f);                                       /* :50 */
    if (update) e->bgimg[i] = sat_u8_f(aw(i_f, alpha, b_f, beta) * 255.f); /* :54-58 */
    e->tmp8a[i] = sat_u8_f(d * 255.f);                                      /* :64-65 */
  }
  if (update && limit > 0 && limit < e->counter) e->counter++; /* :60-61 */
  gray_thr(e, e->tmp8a, e->tmp8b);                             /* :67-71 */
  write_mask(e, e->tmp8b, fg, fg_step);
  write_img(e, e->bgimg, e->ch, bg, bg_step);
  return BGS_FG_VALID | BGS_BG_VALID;
}

/* AdaptiveSelectiveBackgroundLearning.cpp:31-105 (state and outputs are single-channel) */
static uint32_t asbl_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step) {
  uint8_t* gray = e->tmp8a;
  for (size_t i = 0; i < e->n; ++i) gray[i] = e->ch == 3 ? gray_bgr(e->cur[3 * i], e->cur[3 * i + 1], e->cur[3 * i + 2]) : e->cur[i]; /* :37-40 */
  if (!e->have1) { /* :47-48 */
    memcpy(e->bgimg, gray, e->n);
    e->have1 = 1;
  }
  const float sf = (float)(1. / 255.);
  uint8_t* raw = e->tmp8b;
  for (size_t i = 0; i < e->n; ++i) {
    const float d = fabsf((float)gray[i] * sf - (float)e->bgimg[i] * sf); /* :50-57 */
    raw[i] = thr_bin(sat_u8_f(d * 255.f), e->p.threshold);                /* :59-62: always thresholded */
  }
  uint8_t* med = e->prev2; /* unused history buffer as scratch (>= n bytes) */
  orc_median_blur_u8(raw, med, e->rows, e->cols, 3); /* :63 */
  const int learn = e->p.learning_frames > 0 && e->counter <= e->p.learning_frames; /* :65 */
  const double aL = e->p.alpha_learn, aD = e->p.alpha_detection;
  for (size_t i = 0; i < e->n; ++i) {
    const float i_f = (float)gray[i] * sf;
    float b_f = (float)e->bgimg[i] * sf;
    if (learn)
      b_f = aw(i_f, aL, b_f, 1 - aL); /* :69 */
    else if (med[i] == 0)
      b_f = (float)(aD * (double)i_f + (1 - aD) * (double)b_f); /* :83-86 scalar double expression */
    e->bgimg[i] = sat_u8_f(b_f * 255.f);                         /* :92-94 */
  }
  if (learn) e->counter++; /* :70 */
  write_mask(e, med, fg, fg_step);
  write_img(e, e->bgimg, 1, bg, bg_step);
  return BGS_FG_VALID | BGS_BG_VALID;
}

/* ---------------------------------------------------------------- a7 MOG2 */

/* cv::BackgroundSubtractorMOG2 detectShadowGMM (OpenCV 2.4 bgfg_gaussmix2.cpp) */
static int mog2_shadow(const float* data, int nch, int nmodes, const float* gmm, const float* mean, float Tb, float TB, float tau) {
  float tWeight = 0;
  for (int mode = 0; mode < nmodes; ++mode, mean += nch) {
    const float gw = gmm[2 * mode], gvar = gmm[2 * mode + 1];
    float numerator = 0.0f, denominator = 0.0f;
    for (int c = 0; c < nch; ++c) {
      numerator += data[c] * mean[c];
      denominator += mean[c] * mean[c];
    }
    if (denominator == 0) return 0;
    if (numerator <= denominator && numerator >= tau * denominator) {
      const float a = numerator / denominator;
      float dist2a = 0.0f;
      for (int c = 0; c < nch; ++c) {
        const float dD = a * mean[c] - data[c];
        dist2a += dD * dD;
      }
      if (dist2a < Tb * gvar * a * a) return 1;
    }
    tWeight += gw;
    if (tWeight > TB) return 0;
  }
  return 0;
}

/* One pixel of MOG2Invoker::operator() (OpenCV 2.4 bgfg_gaussmix2.cpp).  gmm = {weight,variance}[K], mean = [K][nch]. */
static uint8_t mog2_pixel(const float* data, int nch, float* gmm, float* mean, uint8_t* modes_used, int K, float alphaT, float alpha1,
                          float prune, float Tb, float TB, float Tg, float varInit, float varMin, float varMax, int detect_shadows,
                          float tau, uint8_t shadow_val) {
  int background = 0, fitsPDF = 0;
  int nmodes = *modes_used;
  const int nNewModes = nmodes;
  float totalWeight = 0.f;
  float dData[3];
  for (int mode = 0; mode < nmodes; ++mode) { /* nmodes shrinks inside the loop when a mode is pruned */
    float* mean_m = mean + mode * nch;
    float weight = alpha1 * gmm[2 * mode] + prune;
    int swap_count = 0;
    if (!fitsPDF) {
      const float var = gmm[2 * mode + 1];
      float dist2;
      if (nch == 3) {
        dData[0] = mean_m[0] - data[0];
        dData[1] = mean_m[1] - data[1];
        dData[2] = mean_m[2] - data[2];
        dist2 = dData[0] * dData[0] + dData[1] * dData[1] + dData[2] * dData[2];
      } else {
        dist2 = 0.f;
        for (int c = 0; c < nch; ++c) {
          dData[c] = mean_m[c] - data[c];
          dist2 += dData[c] * dData[c];
        }
      }
      if (totalWeight < TB && dist2 < Tb * var) background = 1;
      if (dist2 < Tg * var) {
        fitsPDF = 1;
        weight += alphaT;
        const float k = alphaT / weight;
        for (int c = 0; c < nch; ++c) mean_m[c] -= k * dData[c];
        float varnew = var + k * (dist2 - var);
        varnew = varnew > varMin ? varnew : varMin; /* MAX(varnew, varMin) */
        varnew = varnew < varMax ? varnew : varMax; /* MIN(varnew, varMax) */
        gmm[2 * mode + 1] = varnew;
        for (int i = mode; i > 0; --i) {
          if (weight < gmm[2 * (i - 1)]) break;
          swap_count++;
          float t;
          t = gmm[2 * i], gmm[2 * i] = gmm[2 * (i - 1)], gmm[2 * (i - 1)] = t;
          t = gmm[2 * i + 1], gmm[2 * i + 1] = gmm[2 * (i - 1) + 1], gmm[2 * (i - 1) + 1] = t;
          for (int c = 0; c < nch; ++c) t = mean[i * nch + c], mean[i * nch + c] = mean[(i - 1) * nch + c], mean[(i - 1) * nch + c] = t;
        }
      }
    }
    if (weight < -prune) {
      weight = 0.0;
      nmodes--;
    }
    gmm[2 * (mode - swap_count)] = weight;
    totalWeight += weight;
  }
  totalWeight = 1.f / totalWeight;
  for (int mode = 0; mode < nmodes; ++mode) gmm[2 * mode] *= totalWeight;
  nmodes = nNewModes; /* sic: the pruned count is discarded */
  if (!fitsPDF) {
    const int mode = nmodes == K ? K - 1 : nmodes++;
    if (nmodes == 1)
      gmm[2 * mode] = 1.f;
    else {
      gmm[2 * mode] = alphaT;
      for (int i = 0; i < nmodes - 1; ++i) gmm[2 * i] *= alpha1;
    }
    for (int c = 0; c < nch; ++c) mean[mode * nch + c] = data[c];
    gmm[2 * mode + 1] = varInit;
    for (int i = nmodes - 1; i > 0; --i) {
      if (alphaT < gmm[2 * (i - 1)]) break;
      float t;
      t = gmm[2 * i], gmm[2 * i] = gmm[2 * (i - 1)], gmm[2 * (i - 1)] = t;
      t = gmm[2 * i + 1], gmm[2 * i + 1] = gmm[2 * (i - 1) + 1], gmm[2 * (i - 1) + 1] = t;
      for (int c = 0; c < nch; ++c) t = mean[i * nch + c], mean[i * nch + c] = mean[(i - 1) * nch + c], mean[(i - 1) * nch + c] = t;
    }
  }
  *modes_used = (uint8_t)nmodes;
  return background ? 0 : (detect_shadows && mog2_shadow(data, nch, nmodes, gmm, mean, Tb, TB, tau)) ? shadow_val : 255;
}

/* cv::BackgroundSubtractorMOG2::getBackgroundImage, one pixel (3 channels only) */
static void mog2_bg_pixel(const float* gmm, const float* mean, int nmodes, float TB, uint8_t* out) {
  float mv[3] = {0.f, 0.f, 0.f};
  float totalWeight = 0.f;
  for (int g = 0; g < nmodes; ++g) {
    const float w = gmm[2 * g];
    mv[0] += w * mean[3 * g + 0];
    mv[1] += w * mean[3 * g + 1];
    mv[2] += w * mean[3 * g + 2];
    totalWeight += w;
    if (totalWeight > TB) break;
  }
  const float inv = 1.f / totalWeight;
  for (int c = 0; c < 3; ++c) out[c] = sat_u8_f(mv[c] * inv);
}

/* MixtureOfGaussianV2BGS.cpp:29-74 */
static int mog2_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step, uint32_t* flags) {
  const bgs_params* p = &e->p;
  const int K = p->mog2_nmixtures, nch = e->ch;
  double lr = p->alpha;
  if (e->nframes == 0 || lr >= 1) { /* needToInitialize */
    memset(e->gmm, 0, e->n * K * 2 * sizeof(float));
    memset(e->mean, 0, e->n * K * nch * sizeof(float));
    memset(e->modes, 0, e->n);
    e->nframes = 0;
  }
  if (nch != 3) return BGS_ERR_UNSUPPORTED; /* getBackgroundImage CV_Assert(nchannels == 3), called every frame at :59 */
  ++e->nframes;
  const int64_t n2 = 2 * e->nframes;
  lr = (lr >= 0 && e->nframes > 1) ? lr : 1. / (double)(n2 < p->mog2_history ? n2 : p->mog2_history);
  const float alphaT = (float)lr, alpha1 = 1.f - alphaT;
  const float prune = (float)(-lr * (double)p->mog2_ct);
  const float Tb = p->mog2_var_threshold, TB = p->mog2_background_ratio, Tg = p->mog2_var_threshold_gen;
  uint8_t* mask = e->tmp8b;
#pragma omp parallel for num_threads(e->threads) schedule(static)
  for (int y = 0; y < e->rows; ++y) {
    for (int x = 0; x < e->cols; ++x) {
      const size_t i = (size_t)y * e->cols + x;
      float data[3];
      for (int c = 0; c < nch; ++c) data[c] = (float)e->cur[i * nch + c];
      uint8_t m = mog2_pixel(data, nch, e->gmm + i * K * 2, e->mean + i * K * nch, e->modes + i, K, alphaT, alpha1, prune, Tb, TB, Tg,
                             p->mog2_var_init, p->mog2_var_min, p->mog2_var_max, p->mog2_detect_shadows, p->mog2_tau,
                             (uint8_t)p->mog2_shadow_value);
      mask[i] = p->enable_threshold ? thr_bin(m, p->threshold) : m; /* :61-62 */
    }
  }
  write_mask(e, mask, fg, fg_step);
  if (bg) { /* :59 */
#pragma omp parallel for num_threads(e->threads) schedule(static)
    for (int y = 0; y < e->rows; ++y)
      for (int x = 0; x < e->cols; ++x) {
        const size_t i = (size_t)y * e->cols + x;
        mog2_bg_pixel(e->gmm + i * K * 2, e->mean + i * K * 3, e->modes[i], TB, bg + (size_t)y * bg_step + 3 * (size_t)x);
      }
  }
  *flags = BGS_FG_VALID | BGS_BG_VALID;
  return BGS_OK;
}

/* ---------------------------------------------------------------- a8 MOG (v1) */

/* cv::BackgroundSubtractorMOG process8uC3 / process8uC1 (OpenCV 2.4 bgfg_gaussmix.cpp).
 * One record = {sortKey, weight, mean[C], var[C]}. */
#define MIX_SK(m) ((m)[0])
#define MIX_W(m) ((m)[1])
static uint8_t mog1_pixel(const float* pix, int C, float* mptr, int K, float alpha, float T, float vT, float w0, float sk0, float var0,
                          float minVar) {
  const int R = 2 + 2 * C;
  int k, k1, kHit = -1, kForeground = -1;
  if (alpha > 0) {
    float wsum = 0;
    for (k = 0; k < K; ++k) {
      float* m = mptr + k * R;
      const float w = MIX_W(m);
      wsum += w;
      if (w < FLT_EPSILON) break;
      float diff[3], d2 = 0, vsum = 0;
      for (int c = 0; c < C; ++c) {
        diff[c] = pix[c] - m[2 + c];
        d2 += diff[c] * diff[c]; /* Vec::dot: s = 0; s += a[i]*b[i] */
      }
      if (C == 3)
        vsum = m[2 + C] + m[2 + C + 1] + m[2 + C + 2];
      else
        vsum = m[2 + C];
      if (d2 < vT * vsum) {
        wsum -= w;
        const float dw = alpha * (1.f - w);
        MIX_W(m) = w + dw;
        float vs2 = 0;
        for (int c = 0; c < C; ++c) {
          m[2 + c] = m[2 + c] + alpha * diff[c];
          const float v = m[2 + C + c] + alpha * (diff[c] * diff[c] - m[2 + C + c]);
          m[2 + C + c] = v > minVar ? v : minVar; /* max(v, minVar) */
        }
        if (C == 3)
          vs2 = m[2 + C] + m[2 + C + 1] + m[2 + C + 2];
        else
          vs2 = m[2 + C];
        MIX_SK(m) = w / sqrtf(vs2); /* sic: the OLD weight */
        for (k1 = k - 1; k1 >= 0; --k1) {
          float* a = mptr + k1 * R;
          float* b = mptr + (k1 + 1) * R;
          if (MIX_SK(a) >= MIX_SK(b)) break;
          for (int j = 0; j < R; ++j) {
            const float t = a[j];
            a[j] = b[j];
            b[j] = t;
          }
        }
        kHit = k1 + 1;
        break;
      }
    }
    if (kHit < 0) {
      kHit = k = (k < K - 1 ? k : K - 1);
      float* m = mptr + k * R;
      wsum += w0 - MIX_W(m);
      MIX_W(m) = w0;
      for (int c = 0; c < C; ++c) {
        m[2 + c] = pix[c];
        m[2 + C + c] = var0;
      }
      MIX_SK(m) = sk0;
    } else
      for (; k < K; ++k) wsum += MIX_W(mptr + k * R);
    const float wscale = 1.f / wsum;
    wsum = 0;
    for (k = 0; k < K; ++k) {
      float* m = mptr + k * R;
      MIX_W(m) *= wscale;
      wsum += MIX_W(m);
      MIX_SK(m) *= wscale;
      if (wsum > T && kForeground < 0) kForeground = k + 1;
    }
    return (uint8_t)(-(kHit >= kForeground));
  }
  /* alpha == 0: classify only */
  for (k = 0; k < K; ++k) {
    float* m = mptr + k * R;
    if (MIX_W(m) < FLT_EPSILON) break;
    float d2 = 0, vsum;
    for (int c = 0; c < C; ++c) {
      const float d = pix[c] - m[2 + c];
      d2 += d * d;
    }
    vsum = C == 3 ? m[2 + C] + m[2 + C + 1] + m[2 + C + 2] : m[2 + C];
    if (d2 < vT * vsum) {
      kHit = k;
      break;
    }
  }
  if (kHit >= 0) {
    float wsum = 0;
    for (k = 0; k < K; ++k) {
      wsum += MIX_W(mptr + k * R);
      if (wsum > T) {
        kForeground = k + 1;
        break;
      }
    }
  }
  return (uint8_t)(kHit < 0 || kHit >= kForeground ? 255 : 0);
}

/* MixtureOfGaussianV1BGS.cpp:29-71 */
static int mog1_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint32_t* flags) {
  const bgs_params* p = &e->p;
  const int K = p->mog1_nmixtures, C = e->ch, R = 2 + 2 * C;
  double lr = p->alpha;
  if (e->nframes == 0 || lr >= 1) {
    memset(e->mix, 0, e->n * K * R * sizeof(float));
    e->nframes = 0;
  }
  ++e->nframes;
  lr = (lr >= 0 && e->nframes > 1) ? lr : 1. / (double)(e->nframes < p->mog1_history ? e->nframes : p->mog1_history);
  const float alpha = (float)lr, T = (float)p->mog1_background_ratio, vT = (float)p->mog1_var_threshold;
  const double defaultNoiseSigma = 30 * 0.5;
  const float w0 = (float)0.05;
  const float sk0 = C == 3 ? (float)(w0 / (defaultNoiseSigma * 2 * sqrt(3.))) : (float)(w0 / (defaultNoiseSigma * 2));
  const float var0 = (float)(defaultNoiseSigma * defaultNoiseSigma * 4);
  const float minVar = (float)(p->mog1_noise_sigma * p->mog1_noise_sigma);
  uint8_t* mask = e->tmp8b;
#pragma omp parallel for num_threads(e->threads) schedule(static)
  for (int y = 0; y < e->rows; ++y)
    for (int x = 0; x < e->cols; ++x) {
      const size_t i = (size_t)y * e->cols + x;
      float pix[3];
      for (int c = 0; c < C; ++c) pix[c] = (float)e->cur[i * C + c];
      const uint8_t m = mog1_pixel(pix, C, e->mix + i * K * R, K, alpha, T, vT, w0, sk0, var0, minVar);
      mask[i] = p->enable_threshold ? thr_bin(m, p->threshold) : m; /* :55-56 */
    }
  write_mask(e, mask, fg, fg_step);
  *flags = BGS_FG_VALID; /* BackgroundSubtractorMOG has no getBackgroundImage: img_bgmodel ends up empty (:53, :68) */
  return BGS_OK;
}

/* ---------------------------------------------------------------- a9 GMG */

/* cv::BackgroundSubtractorGMG (OpenCV 2.4 modules/video/src/bgfg_gmg.cpp) as GMG::process drives it
 * (package_bgs/GMG.cpp:35-77: initializationFrames = 20, decisionThreshold = 0.7).  PARITY UNPINNED, and the LOWEST-confidence
 * recall in this file; assumptions that matter, written down so they can be re-pinned:
 *   G1 quantisation      : feature = OR_c ( (int)((v_c - 0.0) * levels / (255.0 - 0.0)) << 8c )           [double, truncation]
 *   G2 decay             : weights[i] *= 1.0f - learningRate  with a double learningRate -> (float)((double)w * (1.0 - lr))
 *   G3 insertFeature     : found -> weight += weights[i], move to front (memmove of the i entries before it);
 *                          list full -> drop the last entry, new one in front; else append, ++nfeatures, return true
 *   G4 training frames   : insertFeature(colour, 1.0f) and THE FEATURE COUNT PERSISTS (int& nfeatures); normalise on frame init-1
 *   G5 decision          : posterior = w*prior / (w*prior + (1-w)*(1-prior)) in double; foreground iff (1 - posterior) > threshold
 *   G6 after the pixel loop: cv::medianBlur(fgmask, smoothingRadius); ++frameNum
 *   G7 BackgroundSubtractorGMG has no getBackgroundImage -> img_bgmodel ends up empty (GMG.cpp:59, :74) */
static void gmg_normalize(float* w, int n) {
  float total = 0.0f;
  for (int i = 0; i < n; ++i) total += w[i];
  if (total != 0.0f)
    for (int i = 0; i < n; ++i) w[i] /= total;
}
static int gmg_insert(int color, float weight, int* colors, float* weights, int* nfeatures, int maxFeatures) {
  int idx = -1;
  for (int i = 0; i < *nfeatures; ++i)
    if (color == colors[i]) {
      weight += weights[i];
      idx = i;
      break;
    }
  if (idx >= 0) {
    memmove(colors + 1, colors, (size_t)idx * sizeof(int));
    memmove(weights + 1, weights, (size_t)idx * sizeof(float));
    colors[0] = color;
    weights[0] = weight;
  } else if (*nfeatures == maxFeatures) {
    memmove(colors + 1, colors, (size_t)(*nfeatures - 1) * sizeof(int));
    memmove(weights + 1, weights, (size_t)(*nfeatures - 1) * sizeof(float));
    colors[0] = color;
    weights[0] = weight;
  } else {
    colors[*nfeatures] = color;
    weights[*nfeatures] = weight;
    ++*nfeatures;
    return 1;
  }
  return 0;
}

static int gmg_process(orc_engine* e, uint8_t* fg, size_t fg_step, uint32_t* flags) {
  const bgs_params* p = &e->p;
  const int F = p->gmg_max_features, C = e->ch;
  if (F < 1 || F > 64) return BGS_ERR_UNSUPPORTED;
  if (!e->gmg_colors) {
    e->gmg_colors = (int32_t*)calloc(e->n * F, sizeof(int32_t));
    e->gmg_weights = (float*)calloc(e->n * F, sizeof(float));
    e->gmg_nfeat = (int32_t*)calloc(e->n, sizeof(int32_t));
  }
  const int64_t frameNum = e->nframes; /* frames processed before this one */
  const double lr = p->gmg_learning_rate, prior = p->gmg_background_prior, thr = p->gmg_decision_threshold;
  uint8_t* mask = e->tmp8a;
  for (size_t i = 0; i < e->n; ++i) {
    int* colors = e->gmg_colors + i * F;
    float* weights = e->gmg_weights + i * F;
    int nf = e->gmg_nfeat[i];
    unsigned feat = 0;
    for (int c = 0, shift = 0; c < C; ++c, shift += 8) feat |= (unsigned)(int)(((double)e->cur[i * C + c] - 0.0) * p->gmg_quantization_levels / (255.0 - 0.0)) << shift; /* G1 */
    const int color = (int)feat;
    int isfg = 0;
    if (frameNum >= p->gmg_init_frames) {
      double weight = 0.0;
      for (int k = 0; k < nf; ++k)
        if (color == colors[k]) {
          weight = weights[k];
          break;
        }
      const double posterior = (weight * prior) / (weight * prior + (1.0 - weight) * (1.0 - prior)); /* G5 */
      isfg = (1.0 - posterior) > thr;
      if (p->gmg_update_background_model) {
        for (int k = 0; k < nf; ++k) weights[k] = (float)((double)weights[k] * (1.0 - lr)); /* G2 */
        if (gmg_insert(color, (float)lr, colors, weights, &nf, F)) gmg_normalize(weights, nf);
      }
    } else if (p->gmg_update_background_model) {
      gmg_insert(color, 1.0f, colors, weights, &nf, F); /* G4 */
      if (frameNum == p->gmg_init_frames - 1) gmg_normalize(weights, nf);
    }
    e->gmg_nfeat[i] = nf;
    mask[i] = isfg ? 255 : 0;
  }
  const uint8_t* out = mask;
  if (p->gmg_smoothing_radius > 0) { /* G6 */
    orc_median_blur_u8(mask, e->tmp8b, e->rows, e->cols, p->gmg_smoothing_radius);
    out = e->tmp8b;
  }
  write_mask(e, out, fg, fg_step);
  *flags = BGS_FG_VALID; /* G7 */
  return BGS_OK;
}

/* ---------------------------------------------------------------- SigmaDelta (N4) */

/* SigmaDeltaBGS::process (package_bgs/bl/SigmaDeltaBGS.cpp:20-55) over sdLaMa091 (package_bgs/bl/sdLaMa091.cpp).
 * PINNED: tests compare this restatement with the reference's own sdLaMa091.cpp compiled into oracle/_ref.
 * Quirks kept on purpose:
 *  - AllocInit_8u_C3R initialises Vt through the C1R routine, i.e. only the first `cols` BYTES of every 3*cols-byte row get Vmin
 *    (sdLaMa091.cpp:190-201, 211-212); the rest is whatever malloc returned - zero pages for the mmap-sized buffers of real
 *    frames (>= 128 KB), which is what this restatement (and the GPU engine) uses;
 *  - Ot = absVal((int8_t)(Mt - I)): the difference wraps to int8 before the absolute value (:66-68, :559);
 *  - ++Vt / --Vt act on a uint8 (255 + 1 wraps to 0), then clamp with uint8-typed min/max (:576-583, :70-76). */
static uint32_t sd_process(orc_engine* e, uint8_t* fg, size_t fg_step) {
  const size_t nb = e->n * 3;
  const uint32_t N = (uint32_t)e->p.sd_amp_factor;
  const uint8_t vmin = (uint8_t)e->p.sd_min_var, vmax = (uint8_t)e->p.sd_max_var;
  if (!e->have1) { /* SigmaDeltaBGS.cpp:33-39: first frame allocates + initialises, no output */
    e->vt = (uint8_t*)malloc(nb);
    memcpy(e->bgimg, e->cur, nb);
    for (int y = 0; y < e->rows; ++y)
      for (int j = 0; j < 3 * e->cols; ++j) e->vt[(size_t)y * 3 * e->cols + j] = j < e->cols ? vmin : 0;
    e->have1 = 1;
    return 0;
  }
  uint8_t* mask = e->tmp8b;
  for (size_t px = 0; px < e->n; ++px) {
    int isfg = 0;
    for (int c = 0; c < 3; ++c) {
      const size_t i = 3 * px + c;
      uint8_t mt = e->bgimg[i];
      const uint8_t im = e->cur[i];
      if (mt < im) ++mt; else if (mt > im) --mt;                 /* :535-540 */
      const int8_t d8 = (int8_t)(uint8_t)(mt - im);              /* :559 absVal(int8_t) */
      const uint8_t ot = d8 < 0 ? (uint8_t)-d8 : (uint8_t)d8;
      const uint32_t amp = N * ot;                               /* :576 */
      uint8_t vt = e->vt[i];
      if (vt < amp) ++vt; else if (vt > amp) --vt;               /* :578-581, uint8 wrap */
      vt = vt < vmax ? vt : vmax;                                /* min(vt, Vmax) */
      vt = vt > vmin ? vt : vmin;                                /* max(., Vmin) */
      if (ot >= vt) isfg = 1;                                    /* :605 */
      e->bgimg[i] = mt;
      e->vt[i] = vt;
    }
    mask[px] = isfg ? 255 : 0;
  }
  write_mask(e, mask, fg, fg_step); /* SigmaDeltaBGS.cpp:44-52: first channel of the 3-channel map */
  return BGS_FG_VALID;
}

/* ---------------------------------------------------------------- dispatch */

int orc_process(orc_engine* e, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step, uint8_t* bg,
                size_t bg_step, uint32_t* out_flags) {
  uint32_t flags = 0;
  if (out_flags) *out_flags = 0;
  if (!e) return BGS_ERR_INVALID;
  if (!in || rows <= 0 || cols <= 0) return BGS_OK; /* if(img_input.empty()) return; */
  int rc = ensure_geometry(e, rows, cols, channels);
  if (rc) return rc;
  if (in_step < (size_t)cols * channels) return BGS_ERR_INVALID;
  for (int y = 0; y < rows; ++y) memcpy(e->cur + (size_t)y * cols * channels, in + (size_t)y * in_step, (size_t)cols * channels);
  switch (e->algo) {
    case BGS_FRAME_DIFF: flags = fd_process(e, fg, fg_step); break;
    case BGS_STATIC_FRAME_DIFF: flags = sfd_process(e, fg, fg_step, bg, bg_step); break;
    case BGS_WMM: flags = wmm_process(e, fg, fg_step, bg, bg_step); break;
    case BGS_WMV: flags = wmv_process(e, fg, fg_step); break;
    case BGS_ABL: flags = abl_process(e, fg, fg_step, bg, bg_step); break;
    case BGS_ASBL: flags = asbl_process(e, fg, fg_step, bg, bg_step); break;
    case BGS_MOG2: rc = mog2_process(e, fg, fg_step, bg, bg_step, &flags); break;
    case BGS_MOG1: rc = mog1_process(e, fg, fg_step, &flags); break;
    case BGS_GMG: rc = gmg_process(e, fg, fg_step, &flags); break;
    case BGS_SUBSENSE: { /* SuBSENSEBGS::process, package_bgs/pl/SuBSENSE.cpp:21-45 */
      if (!e->ss) { /* :27-36 first frame: construct + initialize(img, ROI = all 255), then fall through to operator() */
        rc = ss_create(&e->p, e->cur, rows, cols, channels, &e->ss);
        if (rc) return rc;
      }
      uint8_t* bgc = bg ? (uint8_t*)malloc(e->n * channels) : NULL;
      rc = ss_process(e->ss, e->cur, e->tmp8b, bgc);
      write_mask(e, e->tmp8b, fg, fg_step);
      if (bgc) {
        write_img(e, bgc, channels, bg, bg_step);
        free(bgc);
      }
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_SIGMA_DELTA:
      if (channels != 3) return BGS_ERR_UNSUPPORTED;
      flags = sd_process(e, fg, fg_step);
      break;
    case BGS_LOBSTER: { /* LOBSTERBGS::process, package_bgs/pl/LOBSTER.cpp:20-45 */
      if (!e->ss) {
        rc = lob_create(&e->p, e->cur, rows, cols, channels, &e->ss);
        if (rc) return rc;
      }
      uint8_t* bgc = bg ? (uint8_t*)malloc(e->n * channels) : NULL;
      rc = lob_process(e->ss, e->cur, e->tmp8b, bgc);
      write_mask(e, e->tmp8b, fg, fg_step);
      if (bgc) {
        write_img(e, bgc, channels, bg, bg_step);
        free(bgc);
      }
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_DP_ZIVKOVIC_AGMM:
    case BGS_DP_GRIMSON_GMM:
    case BGS_DP_WREN_GA:
    case BGS_DP_MEAN:
    case BGS_DP_ADAPTIVE_MEDIAN: /* dp_oracle.c; RgbImage accessors assume 3 channels */
      if (channels != 3) return BGS_ERR_UNSUPPORTED;
      if (!e->dp) {
        rc = dp_create(e->algo, &e->p, e->cur, rows, cols, &e->dp);
        if (rc) return rc;
      }
      rc = dp_process(e->dp, e->cur, e->nframes, e->tmp8b);
      write_mask(e, e->tmp8b, fg, fg_step);
      flags = BGS_FG_VALID;
      break;
    default: return BGS_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (e->algo != BGS_MOG2 && e->algo != BGS_MOG1) e->nframes++;
  if (out_flags) *out_flags = flags;
  return BGS_OK;
}

/* canonical SoA export (same plane names / order as bgs_get_state) */
int64_t orc_get_state(orc_engine* e, const char* plane, void* dst, size_t cap) {
  if (!e || !plane || !dst || !e->n) return BGS_ERR_STATE;
  const size_t n = e->n;
  const int C = e->ch;
#define NEED(bytes)                    \
  do {                                 \
    if (cap < (bytes)) return BGS_ERR_STATE; \
  } while (0)
  if (e->algo == BGS_MOG2) {
    const int K = e->p.mog2_nmixtures;
    float* f = (float*)dst;
    if (!strcmp(plane, "w") || !strcmp(plane, "var")) {
      const int off = !strcmp(plane, "var");
      NEED(n * K * 4);
      for (int k = 0; k < K; ++k)
        for (size_t i = 0; i < n; ++i) f[k * n + i] = e->gmm[(i * K + k) * 2 + off];
      return (int64_t)(n * K * 4);
    }
    if (!strcmp(plane, "mu")) {
      NEED(n * K * C * 4);
      for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
          for (size_t i = 0; i < n; ++i) f[((size_t)k * C + c) * n + i] = e->mean[(i * K + k) * C + c];
      return (int64_t)(n * K * C * 4);
    }
    if (!strcmp(plane, "nmodes")) {
      NEED(n);
      memcpy(dst, e->modes, n);
      return (int64_t)n;
    }
  }
  if (e->algo == BGS_MOG1) {
    const int K = e->p.mog1_nmixtures, R = 2 + 2 * C;
    float* f = (float*)dst;
    if (!strcmp(plane, "sortkey") || !strcmp(plane, "w")) {
      const int off = !strcmp(plane, "w");
      NEED(n * K * 4);
      for (int k = 0; k < K; ++k)
        for (size_t i = 0; i < n; ++i) f[k * n + i] = e->mix[(i * K + k) * R + off];
      return (int64_t)(n * K * 4);
    }
    if (!strcmp(plane, "mu") || !strcmp(plane, "var")) {
      const int off = 2 + (!strcmp(plane, "var") ? C : 0);
      NEED(n * K * C * 4);
      for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
          for (size_t i = 0; i < n; ++i) f[((size_t)k * C + c) * n + i] = e->mix[(i * K + k) * R + off + c];
      return (int64_t)(n * K * C * 4);
    }
  }
  if ((e->algo == BGS_SUBSENSE || e->algo == BGS_LOBSTER) && e->ss) return ss_get_state(e->ss, plane, dst, cap);
  if (e->dp) return dp_get_state(e->dp, plane, dst, cap);
  if (e->algo == BGS_GMG && e->gmg_colors) { /* canonical: colors int32 [F][n], weights f32 [F][n], nfeatures int32 [n] */
    const int F = e->p.gmg_max_features;
    if (!strcmp(plane, "nfeatures")) {
      NEED(n * 4);
      memcpy(dst, e->gmg_nfeat, n * 4);
      return (int64_t)(n * 4);
    }
    if (!strcmp(plane, "colors") || !strcmp(plane, "weights")) {
      NEED(n * F * 4);
      const int isw = !strcmp(plane, "weights");
      for (int f = 0; f < F; ++f)
        for (size_t i = 0; i < n; ++i) {
          const int live = f < e->gmg_nfeat[i]; /* entries past the count are dead storage: exported as 0 */
          if (isw)
            ((float*)dst)[(size_t)f * n + i] = live ? e->gmg_weights[i * F + f] : 0.f;
          else
            ((int32_t*)dst)[(size_t)f * n + i] = live ? e->gmg_colors[i * F + f] : 0;
        }
      return (int64_t)(n * F * 4);
    }
  }
  if (e->algo == BGS_SIGMA_DELTA && e->have1 && (!strcmp(plane, "mt") || !strcmp(plane, "vt"))) {
    NEED(n * 3);
    memcpy(dst, !strcmp(plane, "mt") ? e->bgimg : e->vt, n * 3);
    return (int64_t)(n * 3);
  }
  if (!strcmp(plane, "bg") && (e->algo == BGS_STATIC_FRAME_DIFF || e->algo == BGS_ABL || e->algo == BGS_ASBL)) {
    const size_t nb = n * (e->algo == BGS_ASBL ? 1 : C);
    NEED(nb);
    memcpy(dst, e->bgimg, nb);
    return (int64_t)nb;
  }
  if (!strcmp(plane, "prev1") && e->have1 && (e->algo == BGS_FRAME_DIFF || e->algo == BGS_WMM || e->algo == BGS_WMV)) {
    NEED(n * C);
    memcpy(dst, e->prev1, n * C);
    return (int64_t)(n * C);
  }
  if (!strcmp(plane, "prev2") && e->have2 && (e->algo == BGS_WMM || e->algo == BGS_WMV)) {
    NEED(n * C);
    memcpy(dst, e->prev2, n * C);
    return (int64_t)(n * C);
  }
  return BGS_ERR_STATE;
#undef NEED
}
