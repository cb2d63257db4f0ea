/*
 * subsense_oracle.c — CPU restatement of SuBSENSEBGS (package_bgs/pl/SuBSENSE.cpp:21-45 over
 * package_bgs/pl/BackgroundSubtractorSuBSENSE.cpp), 3-channel (:437-584) and 1-channel (:306-434) paths.  TEST INFRASTRUCTURE ONLY (see bgs_oracle.h).
 *
 * PARITY UNPINNED, and by construction only partly comparable with the reference:
 *   (1) the reference consumes libc rand() sequentially in raster order (no srand anywhere) — not reproducible by any
 *       parallel implementation.  Contract: every random draw is a counter-based hash of
 *       (frame index, pixel index, draw slot), ss_rand() below; the HIP kernels use the same function.
 *   (2) the reference's raster loop lets a pixel see model writes made earlier in the SAME frame by pixels above/left
 *       of it (neighbour diffusion, :538-551), and read its random neighbour's D_last / raw-segm means in whatever
 *       state the raster order left them.  Contract: two phases per frame.  Phase A classifies every pixel and updates
 *       its own maps from the model as it stood at the start of the frame (neighbour means are read from the previous
 *       frame's copy); phase B then applies all sample writes — self updates and neighbour diffusion — with raster
 *       order deciding conflicts (the write from the highest source pixel index wins, self before neighbour).
 *   What is identical to the reference for a given (model, frame): thresholds, LBSP intra/inter descriptors, the
 *   sample-consensus test and its early exit, all feedback formulas, the post-processing chain, the frame-level block.
 *   (3) OpenCV primitives (morphology, floodFill, medianBlur, addWeighted, INTER_AREA resize, accumulateWeighted) follow
 *       SURVEY.md App. A.  The frame-level down-sampling (cv::resize ... INTER_AREA to width/8 x height/8, :153, :656) takes
 *       OpenCV's integer-ratio path (resizeAreaFast_: sum * (1/64) in float, saturate_cast) when rows and cols are multiples of 8
 *       and the general one otherwise (resizeArea_<uchar, float> with computeResizeAreaTab's fractional cell weights: per source
 *       row buf += S * alpha over the cells of a destination column in table order, then sum = beta * buf for the first source
 *       row of a destination row and sum += beta * buf for the others, saturate_cast at the end) - recalled from OpenCV 2.4
 *       imgwarp.cpp like everything else that is OpenCV's; area_span() below is that table for one destination index.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bgs_oracle.h"
#include "subsense_oracle.h"

/* ------------------------------------------------------------------------------------------------ shared contract */

static inline uint8_t sat_u8_f(float v); /* cv::saturate_cast<uchar>(float), below */

/* computeResizeAreaTab (OpenCV 2.4 imgwarp.cpp) for ONE destination index d of an axis with ssize source and dsize destination
 * cells, scale = ssize / dsize > 1: a left partial cell (index l, weight al; l = -1: none), the whole cells [s1, s2) with weight af
 * each, a right partial cell (index r, weight ar; r = -1: none) - in that order, which is the order resizeArea_ accumulates in. */
typedef struct { int l, s1, s2, r; float al, af, ar; } area_span_t;
static area_span_t area_span(int ssize, int dsize, int d) {
  const double scale = 1.0 / ((double)dsize / ssize);  /* as cv::resize computes it: inv_scale = dsize / ssize, scale = 1 / inv_scale (differs from ssize / dsize in the last ulp for some sizes) */
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  if (sx2 > ssize - 1) sx2 = ssize - 1;
  if (sx1 > sx2) sx1 = sx2;
  area_span_t a;
  a.l = a.r = -1, a.al = a.ar = 0.f, a.s1 = sx1, a.s2 = sx2, a.af = (float)(1.0 / cell);
  if (sx1 - fsx1 > 1e-3) a.l = sx1 - 1, a.al = (float)((sx1 - fsx1) / cell);
  if (fsx2 - sx2 > 1e-3) {
    double w = fsx2 - sx2;
    if (w > 1.) w = 1.;
    if (w > cell) w = cell;
    a.r = sx2, a.ar = (float)(w / cell);
  }
  return a;
}
/* one source row's contribution to destination column x (channel c): resizeArea_'s buf[dx] */
static float area_row(const uint8_t* row, int C, int c, area_span_t ax) {
  float buf = 0.f;
  if (ax.l >= 0) buf += row[(size_t)ax.l * C + c] * ax.al;
  for (int sx = ax.s1; sx < ax.s2; ++sx) buf += row[(size_t)sx * C + c] * ax.af;
  if (ax.r >= 0) buf += row[(size_t)ax.r * C + c] * ax.ar;
  return buf;
}
/* cv::resize(img, dsw x dsh, INTER_AREA), general (non-integer ratio) path, one destination value */
static float area_value(const uint8_t* img, int rows, int cols, int C, int dsh, int dsw, int y, int x, int c) {
  const area_span_t ax = area_span(cols, dsw, x), ay = area_span(rows, dsh, y);
  float sum = 0.f;
  int first = 1;
  if (ay.l >= 0) sum = ay.al * area_row(img + (size_t)ay.l * cols * C, C, c, ax), first = 0;
  for (int sy = ay.s1; sy < ay.s2; ++sy) {
    const float b = ay.af * area_row(img + (size_t)sy * cols * C, C, c, ax);
    sum = first ? b : sum + b, first = 0;
  }
  if (ay.r >= 0) {
    const float b = ay.ar * area_row(img + (size_t)ay.r * cols * C, C, c, ax);
    sum = first ? b : sum + b;
  }
  return sum;
}
/* cv::resize(src, dcols x drows, INTER_AREA) for a down-scaling ratio that is not an integer on both axes (exported for the tests) */
void orc_resize_area_u8(const uint8_t* src, int srows, int scols, int ch, uint8_t* dst, int drows, int dcols) {
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x)
      for (int c = 0; c < ch; ++c) dst[((size_t)y * dcols + x) * ch + c] = (uint8_t)sat_u8_f(area_value(src, srows, scols, ch, drows, dcols, y, x, c));
}

uint32_t ss_rand(uint32_t frame, uint32_t pixel, uint32_t draw) {
  uint32_t x = frame * 0x9E3779B1u;
  x ^= pixel + 0x85EBCA6Bu + (x << 6) + (x >> 2);
  x ^= (draw + 1u) * 0xC2B2AE35u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x >> 1; /* 31 bits, like rand() with RAND_MAX = 2^31-1 */
}

/* RandUtils.h:13-25 gaussian 7x7 pattern, floor(fspecial('gaussian',7,2)*512) */
static const int SS_PATTERN[7][7] = {{2, 4, 6, 7, 6, 4, 2},     {4, 8, 12, 14, 12, 8, 4},  {6, 12, 21, 25, 21, 12, 6}, {7, 14, 25, 28, 25, 14, 7},
                                     {6, 12, 21, 25, 21, 12, 6}, {4, 8, 12, 14, 12, 8, 4}, {2, 4, 6, 7, 6, 4, 2}};
/* RandUtils.h:51-56 / :76-83 */
static const int SS_N3[8][2] = {{-1, 1}, {0, 1}, {1, 1}, {-1, 0}, {1, 0}, {-1, -1}, {0, -1}, {1, -1}};
static const int SS_N5[24][2] = {{-2, 2},  {-1, 2},  {0, 2},  {1, 2},  {2, 2},  {-2, 1},  {-1, 1},  {0, 1},  {1, 1},  {2, 1},  {-2, 0},  {-1, 0},
                                 {1, 0},   {2, 0},   {-2, -1}, {-1, -1}, {0, -1}, {1, -1}, {2, -1}, {-2, -2}, {-1, -2}, {0, -2}, {1, -2}, {2, -2}};

/* test instrumentation: when set (n bytes), phase A stores each pixel's sample-loop trip count there */
uint8_t* ss_debug_iters = 0;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* RandUtils.h:28-48 getRandSamplePosition with r = 1 + rnd % 512 */
static void ss_sample_pos(uint32_t rnd, int x0, int y0, int cols, int rows, int* xs, int* ys) {
  int r = 1 + (int)(rnd % 512u), x, y = 0;
  for (x = 0; x < 7; ++x) {
    for (y = 0; y < 7; ++y) {
      r -= SS_PATTERN[y][x];
      if (r <= 0) goto stop;
    }
  }
stop:
  x += x0 - 3;
  y += y0 - 3;
  *xs = clampi(x, 2, cols - 3);
  *ys = clampi(y, 2, rows - 3);
}

static inline int popc16(unsigned v) { return __builtin_popcount(v & 0xffffu); }
static inline uint8_t sat_u8_f(float v) {
  long i = lrint((double)v);
  return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}

/* LBSP_16bits_dbcross_3ch3t.i bit order */
static const int8_t LB_DX[16] = {-1, 1, 1, -1, 1, 0, -1, 0, -2, 2, 2, -2, 0, 0, 2, -2};
static const int8_t LB_DY[16] = {1, -1, 1, -1, 0, -1, 0, 1, -2, 2, -2, 2, 2, -2, 0, 0};

static inline unsigned lbsp1(const uint8_t* img, int cols, int C, int x, int y, int c, int ref, int t) {
  unsigned r = 0;
  for (int b = 0; b < 16; ++b) {
    const int v = img[((size_t)(y + LB_DY[b]) * cols + (x + LB_DX[b])) * C + c];
    r |= (unsigned)(abs(v - ref) > t) << (15 - b);
  }
  return r;
}

/* ------------------------------------------------------------------------------------------------ state */

struct ss_state {
  int rows, cols, C;
  size_t n;
  int nS, nReq, nMinColor, nDescOff, nMov; /* N samples, required, min colour dist thr, desc dist thr offset, samples for moving avgs */
  float relT;
  int lbspOff;
  uint8_t lut[256];
  /* model */
  uint8_t* color;  /* [nS][n][3] */
  uint16_t* desc;  /* [nS][n][3] */
  float *R, *V, *T, *Dlast[2], *DminLT, *DminST, *RawLT, *RawST[2], *FinLT, *FinST;
  int pp; /* which copy of Dlast / RawST is current */
  uint8_t *unstable, *blinks, *lastFG, *lastRaw, *lastRawBlink, *lastDilInv, *lastColor;
  uint16_t* lastDesc;
  /* frame-level */
  int dsw, dsh;
  float *dsLT, *dsST; /* [dsh][dsw][3] */
  int64_t frameIndex, framesSinceReset, cooldown;
  float lastNZ, capLo, capHi;
  int lrScaling, autoReset, use3x3, medK;
  /* scratch */
  uint8_t *raw, *t1, *t2, *t3;
  uint16_t* req; /* [n][2]: self request, neighbour request */
  /* LOBSTER variant (lob_* below) */
  int lobster, nColorThr, nDescThr;
  uint8_t* curColor;  /* scratch: the frame's colour / intra descriptor of pixels that posted an update request */
  uint16_t* curDesc;
};

#define REQ_VALID 0x8000u
#define REQ(slot, code) (uint16_t)(REQ_VALID | ((unsigned)(slot) << 5) | (unsigned)(code)) /* bits 5..14: sample slot (nBGSamples up to 1023), bits 0..4: target */

static float* fmap(size_t n, float v) {
  float* p = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = v;
  return p;
}

void ss_destroy(ss_state* s) {
  if (!s) return;
  void* ptrs[] = {s->color, s->desc, s->R, s->V, s->T, s->Dlast[0], s->Dlast[1], s->DminLT, s->DminST, s->RawLT, s->RawST[0], s->RawST[1], s->FinLT,
                  s->FinST, s->unstable, s->blinks, s->lastFG, s->lastRaw, s->lastRawBlink, s->lastDilInv, s->lastColor, s->lastDesc, s->dsLT, s->dsST,
                  s->raw, s->t1, s->t2, s->t3, s->req, s->curColor, s->curDesc};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
  free(s);
}

/* refreshModel (BackgroundSubtractorSuBSENSE.cpp:249-291, 3-channel branch) under the RNG contract */
static void ss_refresh(ss_state* s, float frac, int force) {
  const int nS = s->nS;
  const int nRefresh = frac < 1.0f ? (int)(frac * nS) : nS;
  const uint32_t fr = (uint32_t)s->frameIndex;
  const int start = frac < 1.0f ? (int)(ss_rand(fr, 0xFFFFFFFFu, 0) % (uint32_t)nS) : 0;
  for (int y = 2; y < s->rows - 2; ++y)
    for (int x = 2; x < s->cols - 2; ++x) {
      const size_t i = (size_t)y * s->cols + x;
      if (!(force || !s->lastFG[i])) continue;
      for (int m = start; m < start + nRefresh; ++m) {
        int xs, ys;
        ss_sample_pos(ss_rand(fr, (uint32_t)i, 16u + (uint32_t)(m - start)), x, y, s->cols, s->rows, &xs, &ys);
        const size_t j = (size_t)ys * s->cols + xs;
        if (force || !s->lastFG[j]) {
          const int k = m % nS;
          for (int c = 0; c < s->C; ++c) {
            s->color[((size_t)k * s->n + i) * s->C + c] = s->lastColor[j * s->C + c];
            s->desc[((size_t)k * s->n + i) * s->C + c] = s->lastDesc[j * s->C + c];
          }
        }
      }
    }
}

/* BackgroundSubtractorSuBSENSE::initialize (:82-247) with ROI = all 255 (SuBSENSE.cpp:36) */
int ss_create(const bgs_params* p, const uint8_t* img, int rows, int cols, int C, ss_state** out) {
  if (rows < 5 || cols < 5) return BGS_ERR_UNSUPPORTED; /* LBSP::validateROI leaves nothing */
  if (C != 1 && C != 3) return BGS_ERR_UNSUPPORTED;
  ss_state* s = (ss_state*)calloc(1, sizeof(*s));
  s->rows = rows, s->cols = cols, s->C = C, s->n = (size_t)rows * cols;
  s->nS = p->subsense_n_samples, s->nReq = p->subsense_n_required, s->nMinColor = p->subsense_min_color_dist_threshold;
  s->nDescOff = p->subsense_desc_dist_threshold_offset;
  s->nMov = p->subsense_samples_for_moving_avgs;
  s->relT = p->lbsp_rel_threshold, s->lbspOff = p->lbsp_threshold_offset; /* base-class ctor default 0 */
  const size_t n = s->n;
  const int total = rows * cols, qvga = 320 * 240;
  if (total >= qvga) { /* nOrigROIPxCount (= total) >= total/2 always holds for the all-255 ROI */
    s->lrScaling = 1, s->autoReset = 1;
    s->use3x3 = !(total > qvga * 2);
    int k = (int)floorf((float)total / qvga + 0.5f) + 9;
    if (k > 14) k = 14;
    s->medK = (k % 2) ? k : k - 1;
    s->capLo = 2.0f, s->capHi = 256.0f;
  } else {
    s->lrScaling = 0, s->autoReset = 0, s->use3x3 = 1, s->medK = 9;
    s->capLo = 4.0f, s->capHi = 512.0f;
  }
  s->T = fmap(n, s->capLo), s->R = fmap(n, 1.0f), s->V = fmap(n, 10.0f);
  s->Dlast[0] = fmap(n, 0), s->Dlast[1] = fmap(n, 0), s->DminLT = fmap(n, 0), s->DminST = fmap(n, 0);
  s->RawLT = fmap(n, 0), s->RawST[0] = fmap(n, 0), s->RawST[1] = fmap(n, 0), s->FinLT = fmap(n, 0), s->FinST = fmap(n, 0);
  s->dsw = cols / 8, s->dsh = rows / 8;
  s->dsLT = fmap((size_t)s->dsw * s->dsh * C + 1, 0), s->dsST = fmap((size_t)s->dsw * s->dsh * C + 1, 0);
  s->unstable = (uint8_t*)calloc(n, 1), s->blinks = (uint8_t*)calloc(n, 1), s->lastFG = (uint8_t*)calloc(n, 1), s->lastRaw = (uint8_t*)calloc(n, 1);
  s->lastRawBlink = (uint8_t*)calloc(n, 1), s->lastDilInv = (uint8_t*)calloc(n, 1);
  s->lastColor = (uint8_t*)calloc(n, C), s->lastDesc = (uint16_t*)calloc(n * C, 2);
  s->color = (uint8_t*)calloc((size_t)s->nS * n, C), s->desc = (uint16_t*)calloc((size_t)s->nS * n * C, 2);
  s->raw = (uint8_t*)calloc(n, 1), s->t1 = (uint8_t*)calloc(n, 1), s->t2 = (uint8_t*)calloc(n, 1), s->t3 = (uint8_t*)calloc(n, 1);
  s->req = (uint16_t*)calloc(n * 2, 2);
  orc_lbsp_lut(s->relT, s->lbspOff, C, s->lut); /* :209-210 (1ch: /3), :227-228 */
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) { /* :229-243 */
      const size_t i = (size_t)y * cols + x;
      for (int c = 0; c < C; ++c) {
        const int v = img[i * C + c];
        s->lastColor[i * C + c] = (uint8_t)v;
        s->lastDesc[i * C + c] = (uint16_t)lbsp1(img, cols, C, x, y, c, v, s->lut[v]);
      }
    }
  ss_refresh(s, 1.0f, 0); /* :246 */
  *out = s;
  return BGS_OK;
}

/* ------------------------------------------------------------------------------------------------ one frame */

static void ss_phase_a(ss_state* s, const uint8_t* img, size_t* nonzero_desc) {
  const int cols = s->cols, rows = s->rows, nS = s->nS, nReq = s->nReq, C = s->C;
  const size_t n = s->n;
  const uint32_t fr = (uint32_t)s->frameIndex;
  const int64_t fi = s->frameIndex;
  const size_t maxColor = 255 * (size_t)C, maxDesc = 16 * (size_t)C; /* s_nColorMaxDataRange_*, s_nDescMaxDataRange_* */
  const float fLT = 1.0f / (float)(fi < s->nMov ? fi : s->nMov), fST = 1.0f / (float)(fi < s->nMov / 4 ? fi : s->nMov / 4); /* :303-304 */
  const float* DlastOld = s->Dlast[s->pp];
  float* DlastNew = s->Dlast[s->pp ^ 1];
  const float* RawSTOld = s->RawST[s->pp];
  float* RawSTNew = s->RawST[s->pp ^ 1];
  memcpy(DlastNew, DlastOld, n * sizeof(float)); /* border pixels keep their values */
  memcpy(RawSTNew, RawSTOld, n * sizeof(float));
  memset(s->raw, 0, n);
  memset(s->req, 0, n * 2 * sizeof(uint16_t));
  size_t nz = 0;
  const int stabOff = s->nMinColor / 5; /* STAB_COLOR_DIST_OFFSET */
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) {
      const size_t i = (size_t)y * cols + x;
      const uint8_t* cur = img + i * C;
      float Rv = s->R[i], Vv = s->V[i], Tv = s->T[i];
      const int unst_old = s->unstable[i];
      /* thresholds :459-463 */
      const size_t colorThr = (size_t)((Rv * (float)s->nMinColor) - (float)((!unst_old) * stabOff)) / (C == 1 ? 2 : 1); /* :328 has a trailing /2 */
      const size_t descThr = ((size_t)1 << ((size_t)floorf(Rv + 0.5f))) + (size_t)s->nDescOff + (size_t)(unst_old * s->nDescOff);
      const size_t totColorThr = colorThr * 3, totDescThr = descThr * 3, scColorThr = totColorThr / 2;
      unsigned intra[3] = {0, 0, 0};
      for (int c = 0; c < C; ++c) intra[c] = lbsp1(img, cols, C, x, y, c, cur[c], s->lut[cur[c]]); /* :465-466 / :331 */
      const int unst = (Rv > 3.0f || (s->RawLT[i] - s->FinLT[i]) > 0.1f || (RawSTOld[i] - s->FinST[i]) > 0.1f) ? 1 : 0; /* :467 */
      s->unstable[i] = (uint8_t)unst;
      size_t minDesc = maxDesc, minSum = maxColor;
      int good = 0, idx = 0;
      while (good < nReq && idx < nS) { /* :469-497 (3ch) / :334-357 (1ch) */
        const uint8_t* bc = s->color + ((size_t)idx * n + i) * C;
        const uint16_t* bd = s->desc + ((size_t)idx * n + i) * C;
        if (C == 1) {
          const size_t cd = (size_t)abs((int)cur[0] - (int)bc[0]);
          if (cd <= colorThr) {
            const size_t intraD = (size_t)popc16(intra[0] ^ bd[0]);
            const unsigned inter = lbsp1(img, cols, 1, x, y, 0, bc[0], s->lut[bc[0]]);
            const size_t dd = (intraD + (size_t)popc16(inter ^ bd[0])) / 2;
            if (dd <= descThr) {
              size_t sd = (dd / 4) * (255 / 16) + cd;
              if (sd > 255) sd = 255;
              if (sd <= colorThr) {
                if (minDesc > dd) minDesc = dd;
                if (minSum > sd) minSum = sd;
                good++;
              }
            }
          }
        } else {
          size_t totDesc = 0, totSum = 0;
          int ok = 1;
          for (int c = 0; c < 3 && ok; ++c) {
            const size_t cd = (size_t)abs((int)cur[c] - (int)bc[c]);
            if (cd > scColorThr) {
              ok = 0;
              break;
            }
            const size_t intraD = (size_t)popc16(intra[c] ^ bd[c]);
            const unsigned inter = lbsp1(img, cols, 3, x, y, c, bc[c], s->lut[bc[c]]);
            const size_t interD = (size_t)popc16(inter ^ bd[c]);
            const size_t dd = (intraD + interD) / 2;
            size_t sd = (dd / 2) * (255 / 16) + cd;
            if (sd > 255) sd = 255;
            if (sd > scColorThr) {
              ok = 0;
              break;
            }
            totDesc += dd;
            totSum += sd;
          }
          if (ok && !(totDesc > totDescThr || totSum > totColorThr)) {
            if (minDesc > totDesc) minDesc = totDesc;
            if (minSum > totSum) minSum = totSum;
            good++;
          }
        }
        idx++;
      }
      if (ss_debug_iters) ss_debug_iters[i] = (uint8_t)idx;
      /* :498-499 */
      size_t l1 = 0, hd = 0;
      for (int c = 0; c < C; ++c) {
        l1 += (size_t)abs((int)s->lastColor[i * C + c] - (int)cur[c]);
        hd += (size_t)popc16(s->lastDesc[i * C + c] ^ intra[c]);
      }
      const float normLast = ((float)l1 / maxColor + (float)hd / maxDesc) / 2;
      const float dlast = DlastOld[i] * (1.0f - fST) + normLast * fST;
      DlastNew[i] = dlast;
      float dminLT = s->DminLT[i], dminST = s->DminST[i], rawLT = s->RawLT[i], rawST = RawSTOld[i];
      int isfg;
      uint16_t reqSelf = 0, reqNbr = 0;
      if (good < nReq) { /* foreground :500-515 */
        float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2 + (float)(nReq - good) / nReq;
        if (nm > 1.0f) nm = 1.0f;
        dminLT = dminLT * (1.0f - fLT) + nm * fLT;
        dminST = dminST * (1.0f - fST) + nm * fST;
        rawLT = rawLT * (1.0f - fLT) + fLT;
        rawST = rawST * (1.0f - fST) + fST;
        isfg = 1;
        if (s->cooldown && (ss_rand(fr, (uint32_t)i, 0) % 2u) == 0) reqSelf = REQ(ss_rand(fr, (uint32_t)i, 1) % (uint32_t)nS, 12);
      } else { /* background :516-552 */
        const float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2;
        dminLT = dminLT * (1.0f - fLT) + nm * fLT;
        dminST = dminST * (1.0f - fST) + nm * fST;
        rawLT = rawLT * (1.0f - fLT);
        rawST = rawST * (1.0f - fST);
        isfg = 0;
        const size_t lr = (size_t)ceilf(Tv);
        if ((ss_rand(fr, (uint32_t)i, 2) % lr) == 0) reqSelf = REQ(ss_rand(fr, (uint32_t)i, 3) % (uint32_t)nS, 12);
        const int use3 = s->use3x3 && !unst;
        int xn, yn;
        if (use3) {
          const int r = (int)(ss_rand(fr, (uint32_t)i, 4) % 8u);
          xn = x + SS_N3[r][0], yn = y + SS_N3[r][1];
        } else {
          const int r = (int)(ss_rand(fr, (uint32_t)i, 4) % 24u);
          xn = x + SS_N5[r][0], yn = y + SS_N5[r][1];
        }
        xn = clampi(xn, 2, cols - 3), yn = clampi(yn, 2, rows - 3);
        const size_t nrand = ss_rand(fr, (uint32_t)i, 5);
        const size_t j = (size_t)yn * cols + xn;
        const float nbrLast = DlastOld[j], nbrRaw = RawSTOld[j]; /* contract (2): previous frame's copy */
        if ((nrand % (use3 ? lr : (lr / 2 + 1))) == 0 || (nbrRaw > 0.995f && nbrLast < 0.010f && (nrand % ((size_t)s->capLo)) == 0))
          reqNbr = REQ(ss_rand(fr, (uint32_t)i, 6) % (uint32_t)nS, (yn - y + 2) * 5 + (xn - x + 2));
      }
      s->DminLT[i] = dminLT, s->DminST[i] = dminST, s->RawLT[i] = rawLT;
      RawSTNew[i] = rawST;
      s->raw[i] = isfg ? 255 : 0;
      s->req[i * 2] = reqSelf, s->req[i * 2 + 1] = reqNbr;
      /* feedback :553-576 */
      const float dmin_min = dminLT < dminST ? dminLT : dminST, dmin_max = dminLT > dminST ? dminLT : dminST;
      if (s->lastFG[i] || (dmin_min < 0.1f && isfg)) {
        if (Tv < s->capHi) Tv += 0.5f / (dmin_max * Vv);
      } else if (Tv > s->capLo)
        Tv -= 0.25f * Vv / dmin_max;
      if (Tv < s->capLo)
        Tv = s->capLo;
      else if (Tv > s->capHi)
        Tv = s->capHi;
      if (dmin_max > 0.1f && s->blinks[i])
        Vv += 1.0f;
      else if (Vv > 0.1f) {
        Vv -= s->lastFG[i] ? 0.1f / 4 : unst ? 0.1f / 2 : 0.1f;
        if (Vv < 0.1f) Vv = 0.1f;
      }
      const float pw = 1.0f + dmin_min * 2;
      if ((double)Rv < (double)pw * (double)pw) /* std::pow(float, int) promotes to double (C++11); the double square is exact */
        Rv += 0.01f * (Vv - 0.1f);
      else {
        Rv -= 0.01f / Vv;
        if (Rv < 1.0f) Rv = 1.0f;
      }
      s->R[i] = Rv, s->V[i] = Vv, s->T[i] = Tv;
      if (C == 3 ? (popc16(intra[0]) + popc16(intra[1]) + popc16(intra[2]) >= 4) : (popc16(intra[0]) >= 2)) ++nz; /* :577-578 / :430-431 */
      for (int c = 0; c < C; ++c) {                                                                            /* :579-582 */
        s->lastDesc[i * C + c] = (uint16_t)intra[c];
        s->lastColor[i * C + c] = cur[c];
      }
    }
  s->pp ^= 1;
  *nonzero_desc = nz;
}

/* Phase B: every target gathers the requests aimed at it from its 5x5 neighbourhood in raster order of the SOURCES */
static void ss_phase_b(ss_state* s) {
  const int cols = s->cols, rows = s->rows;
  const size_t n = s->n;
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) {
      const size_t j = (size_t)y * cols + x;
      for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
          const int ys = y + dy, xs = x + dx;
          if (ys < 2 || ys >= rows - 2 || xs < 2 || xs >= cols - 2) continue;
          const size_t i = (size_t)ys * cols + xs;
          for (int q = 0; q < 2; ++q) { /* self request first, then the neighbour request */
            const unsigned r = s->req[i * 2 + q];
            if (!(r & REQ_VALID)) continue;
            const int code = (int)(r & 0x1f), slot = (int)((r >> 5) & 0x3ff);
            const int ty = ys + code / 5 - 2, tx = xs + code % 5 - 2;
            if (ty != y || tx != x) continue;
            const uint8_t* srcC = s->lobster ? s->curColor : s->lastColor; /* = current frame colour of the source */
            const uint16_t* srcD = s->lobster ? s->curDesc : s->lastDesc;  /* = its current intra descriptor */
            for (int c = 0; c < s->C; ++c) {
              s->color[((size_t)slot * n + j) * s->C + c] = srcC[i * s->C + c];
              s->desc[((size_t)slot * n + j) * s->C + c] = srcD[i * s->C + c];
            }
          }
        }
    }
}

/* 3x3 majority-free general median via the oracle primitive */
static void ss_post(ss_state* s, uint8_t* out) {
  const int rows = s->rows, cols = s->cols;
  const size_t n = s->n;
  uint8_t *pre = s->t1, *flood = s->t2, *tmp = s->t3;
  for (size_t i = 0; i < n; ++i) { /* :624-627 */
    const uint8_t blink = s->raw[i] ^ s->lastRaw[i];
    s->blinks[i] = blink | s->lastRawBlink[i];
    s->lastRawBlink[i] = blink;
    s->lastRaw[i] = s->raw[i];
  }
  orc_dilate3x3(s->raw, tmp, rows, cols, 1); /* MORPH_CLOSE :628 */
  orc_erode3x3(tmp, pre, rows, cols, 1);
  memcpy(flood, pre, n);
  orc_floodfill_from_origin(flood, rows, cols, 255); /* :630 */
  for (size_t i = 0; i < n; ++i) flood[i] = (uint8_t)~flood[i];
  orc_erode3x3(pre, tmp, rows, cols, 3); /* :632 */
  for (size_t i = 0; i < n; ++i) tmp[i] = s->raw[i] | flood[i] | tmp[i];
  orc_median_blur_u8(tmp, s->lastFG, rows, cols, s->medK); /* :635 */
  orc_dilate3x3(s->lastFG, tmp, rows, cols, 3);
  for (size_t i = 0; i < n; ++i) { /* :637-639 */
    s->blinks[i] &= s->lastDilInv[i];
    s->lastDilInv[i] = (uint8_t)~tmp[i];
    s->blinks[i] &= s->lastDilInv[i];
  }
  memcpy(out, s->lastFG, n);
}

int ss_process(ss_state* s, const uint8_t* img, uint8_t* fg /* [n] */, uint8_t* bg /* [n][3] or NULL */) {
  const size_t n = s->n;
  ++s->frameIndex;
  const int64_t fi = s->frameIndex;
  const float fLT = 1.0f / (float)(fi < s->nMov ? fi : s->nMov), fST = 1.0f / (float)(fi < s->nMov / 4 ? fi : s->nMov / 4);
  size_t nz = 0;
  ss_phase_a(s, img, &nz);
  ss_phase_b(s);
  ss_post(s, fg);
  /* :641-642 addWeighted(f32, 1-f, u8, (1/255)*f, 0, dst, CV_32F) */
  const double aLT = (double)(1.0f - fLT), bLT = (1.0 / 255) * (double)fLT, aST = (double)(1.0f - fST), bST = (1.0 / 255) * (double)fST;
  for (size_t i = 0; i < n; ++i) {
    s->FinLT[i] = (float)((double)s->FinLT[i] * aLT + (double)(float)s->lastFG[i] * bLT + 0.0);
    s->FinST[i] = (float)((double)s->FinST[i] * aST + (double)(float)s->lastFG[i] * bST + 0.0);
  }
  /* :643-655 LBSP threshold LUT auto-adjustment */
  const size_t relevant = (size_t)(s->rows - 4) * (s->cols - 4);
  const float ratio = (float)nz / relevant;
  if (ratio < 0.1f && s->lastNZ < 0.1f) {
    for (int t = 0; t < 256; ++t)
      if (s->lut[t] > sat_u8_f((float)((double)s->lbspOff + ceil((double)((float)t * s->relT / 4))))) --s->lut[t];
  } else if (ratio > 0.5f && s->lastNZ > 0.5f) {
    for (int t = 0; t < 256; ++t)
      if (s->lut[t] < sat_u8_f((float)s->lbspOff + 255 * s->relT)) ++s->lut[t];
  }
  s->lastNZ = ratio;
  if (s->lrScaling) { /* :656-699 */
    size_t totDiff = 0;
    const int C = s->C;
    for (int y = 0; y < s->dsh; ++y)
      for (int x = 0; x < s->dsw; ++x) {
        float d[3] = {0, 0, 0};
        for (int c = 0; c < C; ++c) {
          float v;
          if (s->rows % 8 == 0 && s->cols % 8 == 0) { /* INTER_AREA, integer ratio on both axes: resizeAreaFast_ */
            int sum = 0;
            for (int yy = 0; yy < 8; ++yy)
              for (int xx = 0; xx < 8; ++xx) sum += img[((size_t)(y * 8 + yy) * s->cols + (x * 8 + xx)) * C + c];
            v = (float)sat_u8_f((float)sum * (1.f / 64));
          } else {
            v = (float)sat_u8_f(area_value(img, s->rows, s->cols, C, s->dsh, s->dsw, y, x, c)); /* resizeArea_ */
          }
          float* lt = s->dsLT + ((size_t)y * s->dsw + x) * C + c;
          float* st = s->dsST + ((size_t)y * s->dsw + x) * C + c;
          *lt = v * fLT + *lt * (1 - fLT); /* accumulateWeighted: src*a + dst*(1-a) */
          *st = v * fST + *st * (1 - fST);
          d[c] = fabsf(*st - *lt);
        }
        if (C == 1)
          totDiff += (size_t)d[0] / 2; /* :664 */
        else {
          size_t m = (size_t)d[0];
          if ((size_t)d[1] > m) m = (size_t)d[1];
          if ((size_t)d[2] > m) m = (size_t)d[2];
          totDiff += m;
        }
      }
    const float diffRatio = (float)totDiff / (s->dsh * s->dsw);
    const int thr = s->nMinColor / 2; /* FRAMELEVEL_MIN_COLOR_DIFF_THRESHOLD */
    if (s->autoReset) {
      if (s->framesSinceReset > 1000)
        s->autoReset = 0;
      else if (diffRatio >= thr && s->cooldown == 0) {
        s->framesSinceReset = 0;
        ss_refresh(s, 0.1f, 0);
        s->cooldown = s->nMov / 4;
        for (size_t i = 0; i < n; ++i) s->T[i] = 1.0f;
      } else
        ++s->framesSinceReset;
    } else if (diffRatio >= thr * 2) {
      s->framesSinceReset = 0;
      s->autoReset = 1;
    }
    if (diffRatio >= thr / 2) {
      int lo = 2 >> (int)(diffRatio / 2), hi = 256 >> (int)(diffRatio / 2);
      s->capLo = (float)(lo > 1 ? lo : 1);
      s->capHi = (float)(hi > 1 ? hi : 1);
    } else {
      s->capLo = 2.0f, s->capHi = 256.0f;
    }
    if (s->cooldown > 0) --s->cooldown;
  }
  if (bg) { /* getBackgroundImage :702-718 */
    for (size_t i = 0; i < n; ++i)
      for (int c = 0; c < s->C; ++c) {
        float acc = 0;
        for (int k = 0; k < s->nS; ++k) acc += ((float)s->color[((size_t)k * n + i) * s->C + c]) / s->nS;
        bg[i * s->C + c] = sat_u8_f(acc);
      }
  }
  return BGS_OK;
}

int64_t ss_get_state(ss_state* s, const char* plane, void* dst, size_t cap) {
  const size_t n = s->n;
  struct {
    const char* name;
    const void* p;
    size_t bytes;
  } tab[] = {{"R", s->R, n * 4},          {"V", s->V, n * 4},           {"T", s->T, n * 4},           {"Dlast", s->Dlast[s->pp], n * 4},
             {"DminLT", s->DminLT, n * 4}, {"DminST", s->DminST, n * 4}, {"RawLT", s->RawLT, n * 4},   {"RawST", s->RawST[s->pp], n * 4},
             {"FinLT", s->FinLT, n * 4},   {"FinST", s->FinST, n * 4},   {"unstable", s->unstable, n}, {"blinks", s->blinks, n},
             {"lastfg", s->lastFG, n},     {"lastraw", s->lastRaw, n},   {"lastcolor", s->lastColor, n * s->C}, {"lastdesc", s->lastDesc, n * 2 * s->C},
             {"color", s->color, (size_t)s->nS * n * s->C}, {"desc", s->desc, (size_t)s->nS * n * 2 * s->C}, {"lut", s->lut, 256}};
  for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); ++k)
    if (!strcmp(plane, tab[k].name)) {
      if (!tab[k].p || cap < tab[k].bytes) return BGS_ERR_STATE;
      memcpy(dst, tab[k].p, tab[k].bytes);
      return (int64_t)tab[k].bytes;
    }
  if (!strcmp(plane, "scalars")) { /* frameIndex, framesSinceReset, cooldown, capLo, capHi, autoReset, lastNZ */
    if (cap < 7 * sizeof(double)) return BGS_ERR_STATE;
    double* d = (double*)dst;
    d[0] = (double)s->frameIndex, d[1] = (double)s->framesSinceReset, d[2] = (double)s->cooldown, d[3] = s->capLo, d[4] = s->capHi, d[5] = s->autoReset,
    d[6] = s->lastNZ;
    return 7 * sizeof(double);
  }
  return BGS_ERR_STATE;
}


/* ================================================================================================== LOBSTER
 * LOBSTERBGS::process (package_bgs/pl/LOBSTER.cpp:20-45) over BackgroundSubtractorLOBSTER (package_bgs/pl/
 * BackgroundSubtractorLOBSTER.cpp): initialize :29-121, refreshModel :123-170, operator() :172-284, getBackgroundImage
 * :286-303.  Same family as SuBSENSE (LBSP descriptors + colour samples, consensus with early exit, stochastic self /
 * 3x3-neighbour replacement) without the feedback loops, so the same two-phase / counter-RNG contract applies (header
 * of this file): draw slots per pixel and frame are 0 "rand()%nLearningRate" (self), 1 sample slot (self), 2
 * "rand()%nLearningRate" (neighbour), 3 neighbour position, 4 sample slot (neighbour); refreshModel uses ss_refresh.
 * LOBSTERBGS calls operator() with the default learning rate BGSLOBSTER_DEFAULT_LEARNING_RATE = 16. */
#define LOB_LEARNING_RATE 16u

int lob_create(const bgs_params* p, const uint8_t* img, int rows, int cols, int C, ss_state** out) {
  if (rows < 5 || cols < 5) return BGS_ERR_UNSUPPORTED; /* LBSP::validateROI leaves nothing: CV_Assert(nROIPxCount>0) :52 */
  if (C != 1 && C != 3) return BGS_ERR_UNSUPPORTED;
  if (p->subsense_n_required > p->subsense_n_samples) return BGS_ERR_UNSUPPORTED; /* CV_Assert :19 */
  ss_state* s = (ss_state*)calloc(1, sizeof(*s));
  s->lobster = 1;
  s->rows = rows, s->cols = cols, s->C = C, s->n = (size_t)rows * cols;
  s->nS = p->subsense_n_samples, s->nReq = p->subsense_n_required;
  s->nColorThr = p->subsense_min_color_dist_threshold, s->nDescThr = p->subsense_desc_dist_threshold_offset;
  s->relT = p->lbsp_rel_threshold, s->lbspOff = p->lbsp_threshold_offset;
  s->medK = 9; /* DEFAULT_MEDIAN_BLUR_KERNEL_SIZE, BackgroundSubtractorLBSP.cpp:17 */
  const size_t n = s->n;
  s->lastFG = (uint8_t*)calloc(n, 1);
  s->lastColor = (uint8_t*)calloc(n, C), s->lastDesc = (uint16_t*)calloc(n * C, 2);
  s->curColor = (uint8_t*)calloc(n, C), s->curDesc = (uint16_t*)calloc(n * C, 2);
  s->color = (uint8_t*)calloc((size_t)s->nS * n, C), s->desc = (uint16_t*)calloc((size_t)s->nS * n * C, 2);
  s->raw = (uint8_t*)calloc(n, 1);
  s->req = (uint16_t*)calloc(n * 2, 2);
  for (int t = 0; t < 256; ++t) { /* :85-86 (1ch: the sum / 2) and :103-104 */
    float v = (float)(size_t)t * s->relT + (float)(size_t)s->lbspOff;
    if (C == 1) v = v / 2;
    s->lut[t] = sat_u8_f(v);
  }
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) { /* :87-98 / :105-119 */
      const size_t i = (size_t)y * cols + x;
      for (int c = 0; c < C; ++c) {
        const int v = img[i * C + c];
        s->lastColor[i * C + c] = (uint8_t)v;
        s->lastDesc[i * C + c] = (uint16_t)lbsp1(img, cols, C, x, y, c, v, s->lut[v]);
      }
    }
  ss_refresh(s, 1.0f, 0); /* :120 refreshModel(1.0f): frameIndex 0, all of lastFG is 0 */
  *out = s;
  return BGS_OK;
}

int lob_process(ss_state* s, const uint8_t* img, uint8_t* fg, uint8_t* bg) {
  const size_t n = s->n;
  const int rows = s->rows, cols = s->cols, C = s->C, nS = s->nS;
  ++s->frameIndex;
  const uint32_t fr = (uint32_t)s->frameIndex;
  memset(s->raw, 0, n); /* oCurrFGMask = 0 :180 */
  memset(s->req, 0, n * 4);
  /* phase A: classification against the model as it stood at the start of the frame + update requests */
  const size_t descThr3 = (size_t)s->nDescThr * 3, colorThr3 = (size_t)s->nColorThr * 3; /* :225-228 */
  const size_t scDesc = descThr3 / 2, scColor = colorThr3 / 2;
  for (int y = 2; y < rows - 2; ++y)
    for (int x = 2; x < cols - 2; ++x) {
      const size_t i = (size_t)y * cols + x;
      const uint8_t* cur = img + i * C;
      size_t good = 0, idx = 0;
      while (good < (size_t)s->nReq && idx < (size_t)nS) {
        const uint8_t* bc = s->color + (idx * n + i) * C;
        const uint16_t* bd = s->desc + (idx * n + i) * C;
        if (C == 1) { /* :192-205 */
          const size_t cd = (size_t)abs((int)cur[0] - (int)bc[0]);
          if (cd <= (size_t)s->nColorThr / 2) {
            const unsigned in = lbsp1(img, cols, 1, x, y, 0, bc[0], s->lut[bc[0]]);
            if ((size_t)popc16(in ^ bd[0]) <= (size_t)s->nDescThr) good++;
          }
        } else { /* :241-258 */
          size_t totC = 0, totD = 0;
          int ok = 1;
          for (int c = 0; c < 3 && ok; ++c) {
            const size_t cd = (size_t)abs((int)cur[c] - (int)bc[c]);
            if (cd > scColor) {
              ok = 0;
              break;
            }
            const unsigned in = lbsp1(img, cols, 3, x, y, c, bc[c], s->lut[bc[c]]);
            const size_t dd = (size_t)popc16(in ^ bd[c]);
            if (dd > scDesc) {
              ok = 0;
              break;
            }
            totC += cd, totD += dd;
          }
          if (ok && totD <= descThr3 && totC <= colorThr3) good++;
        }
        idx++;
      }
      if (good < (size_t)s->nReq) {
        s->raw[i] = 255; /* :207 / :260 */
      } else {
        const uint32_t pi = (uint32_t)i;
        uint16_t reqSelf = 0, reqNbr = 0;
        if ((ss_rand(fr, pi, 0) % LOB_LEARNING_RATE) == 0) reqSelf = REQ(ss_rand(fr, pi, 1) % (uint32_t)nS, 12); /* :209-214 / :262-269 */
        if ((ss_rand(fr, pi, 2) % LOB_LEARNING_RATE) == 0) { /* :215-222 / :270-279, getRandNeighborPosition_3x3 RandUtils.h:59-71 */
          const int r = (int)(ss_rand(fr, pi, 3) % 8u);
          const int xn = clampi(x + SS_N3[r][0], 2, cols - 3), yn = clampi(y + SS_N3[r][1], 2, rows - 3);
          reqNbr = REQ(ss_rand(fr, pi, 4) % (uint32_t)nS, (yn - y + 2) * 5 + (xn - x + 2));
        }
        s->req[i * 2] = reqSelf, s->req[i * 2 + 1] = reqNbr;
        if (reqSelf || reqNbr)
          for (int c = 0; c < C; ++c) { /* what the update writes: the current colour and its intra descriptor */
            s->curColor[i * C + c] = cur[c];
            s->curDesc[i * C + c] = (uint16_t)lbsp1(img, cols, C, x, y, c, cur[c], s->lut[cur[c]]);
          }
      }
    }
  ss_phase_b(s);
  orc_median_blur_u8(s->raw, s->lastFG, rows, cols, s->medK); /* :281 */
  memcpy(fg, s->lastFG, n);                                    /* :282 */
  if (bg) { /* getBackgroundImage :286-303 */
    for (size_t i = 0; i < n; ++i)
      for (int c = 0; c < C; ++c) {
        float acc = 0;
        for (int k = 0; k < nS; ++k) acc += ((float)s->color[((size_t)k * n + i) * C + c]) / nS;
        bg[i * C + c] = sat_u8_f(acc);
      }
  }
  return BGS_OK;
}
