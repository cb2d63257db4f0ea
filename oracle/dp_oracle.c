/* dp_oracle.c - TEST INFRASTRUCTURE ONLY (the checker the HIP kernels are compared with; never linked into the product).
 *
 * CPU restatement of the in-tree `package_bgs/dp/` background models, statement by statement from the
 * reference's own sources (they are self-contained float/byte arithmetic; they cannot be compiled here only because
 * dp/Image.h includes <opencv2/opencv.hpp>, which is absent - so this is a restatement from source, not from recall):
 *
 *   BGS_DP_ZIVKOVIC_AGMM   DPZivkovicAGMMBGS::process  dp/DPZivkovicAGMMBGS.cpp:29-80  over ZivkovicAGMM::SubtractPixel  dp/ZivkovicAGMM.cpp:103-364
 *   BGS_DP_GRIMSON_GMM     DPGrimsonGMMBGS::process    dp/DPGrimsonGMMBGS.cpp:29-82    over GrimsonGMM::SubtractPixel    dp/GrimsonGMM.cpp:119-295
 *   BGS_DP_WREN_GA         DPWrenGABGS::process        dp/DPWrenGABGS.cpp:29-81        over WrenGA::{SubtractPixel,Update} dp/WrenGA.cpp:79-148
 *   BGS_DP_MEAN            DPMeanBGS::process          dp/DPMeanBGS.cpp:29-82          over MeanBGS::{SubtractPixel,Update} dp/MeanBGS.cpp:52-108
 *   BGS_DP_ADAPTIVE_MEDIAN DPAdaptiveMedianBGS::process dp/DPAdaptiveMedianBGS.cpp:29-81 over AdaptiveMedianBGS dp/AdaptiveMedianBGS.cpp:58-118
 *
 * Wrapper behaviour common to all five (e.g. DPZivkovicAGMMBGS.cpp:41-76): the first frame initialises the model
 * (InitModel) and is then processed like any other; every frame runs Subtract, clears the low-threshold mask and calls
 * Update with it (so the update mask is BACKGROUND everywhere: the conditional update always fires); img_output is the
 * HIGH-threshold mask (high = 2 * low, "used by post-processing"), img_bgmodel is never written.
 * Pixels are the frame's bytes in memory order (pixel(0) is the first byte, i.e. B of a BGR frame; the sources call it R).
 * Arithmetic is float where the source's is float and double where it is double (gcc x86-64: no x87, no FMA contraction).
 */
#include "dp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DP_BACKGROUND 0   /* dp/Bgs.h:40 */
#define DP_FOREGROUND 255 /* dp/Bgs.h:41 */

struct dp_state {
  bgs_algo algo;
  int rows, cols, K;
  size_t n;
  float low, high, alpha;
  int learning_frames, sampling_rate;
  float* modes;    /* Zivkovic: [n][K][5] = sigma, muR, muG, muB, weight (struct GMM, ZivkovicAGMM.h:96-103)
                      Grimson:  [n][K][6] = variance, muR, muG, muB, weight, significants (GrimsonGMM.h) */
  uint8_t* nmodes; /* [n] */
  float* gauss;    /* Wren: [n][4] = mu[3], var[0]  (var[1..2] exist in the source but are never read after init) */
  float* mean;     /* Mean: [n][3] */
  uint8_t* median; /* AdaptiveMedian: [n][3] */
};

void dp_destroy(dp_state* s) {
  if (!s) return;
  free(s->modes), free(s->nmodes), free(s->gauss), free(s->mean), free(s->median);
  free(s);
}

int dp_create(bgs_algo algo, const bgs_params* p, const uint8_t* img, int rows, int cols, dp_state** out) {
  dp_state* s = (dp_state*)calloc(1, sizeof(*s));
  if (!s) return BGS_ERR_NOMEM;
  s->algo = algo, s->rows = rows, s->cols = cols, s->n = (size_t)rows * cols;
  s->low = p->dp_threshold;  /* params.LowThreshold() = threshold */
  s->high = 2 * s->low;      /* params.HighThreshold() = 2*params.LowThreshold() */
  s->alpha = p->dp_alpha;
  s->K = p->dp_gaussians;
  s->learning_frames = p->learning_frames, s->sampling_rate = p->dp_sampling_rate;
  const size_t n = s->n;
  switch (algo) {
    case BGS_DP_ZIVKOVIC_AGMM:
    case BGS_DP_GRIMSON_GMM: {
      if (s->K < 1 || s->K > 8) return dp_destroy(s), BGS_ERR_UNSUPPORTED;
      const int F = algo == BGS_DP_ZIVKOVIC_AGMM ? 5 : 6;
      s->modes = (float*)calloc(n * s->K * F, sizeof(float)); /* InitModel: everything 0 (ZivkovicAGMM.cpp:79-94, GrimsonGMM.cpp:97-109) */
      s->nmodes = (uint8_t*)calloc(n, 1);
      break;
    }
    case BGS_DP_WREN_GA: /* WrenGA::InitModel, WrenGA.cpp:63-77: mu = first frame, var = m_variance = 36 */
      s->gauss = (float*)malloc(n * 4 * sizeof(float));
      for (size_t i = 0; i < n; ++i) {
        for (int ch = 0; ch < 3; ++ch) s->gauss[i * 4 + ch] = img[i * 3 + ch];
        s->gauss[i * 4 + 3] = 36.0f;
      }
      break;
    case BGS_DP_MEAN: /* MeanBGS::InitModel, MeanBGS.cpp:38-50 */
      s->mean = (float*)malloc(n * 3 * sizeof(float));
      for (size_t i = 0; i < n * 3; ++i) s->mean[i] = (float)img[i];
      break;
    case BGS_DP_ADAPTIVE_MEDIAN: /* AdaptiveMedianBGS::InitModel, AdaptiveMedianBGS.cpp:46-56 */
      if (s->sampling_rate == 0) return dp_destroy(s), BGS_ERR_UNSUPPORTED; /* frame_num % 0 */
      s->median = (uint8_t*)malloc(n * 3);
      memcpy(s->median, img, n * 3);
      break;
    default: return dp_destroy(s), BGS_ERR_UNSUPPORTED;
  }
  *out = s;
  return BGS_OK;
}

/* ZivkovicAGMM::SubtractPixel, dp/ZivkovicAGMM.cpp:103-364.  g = the pixel's MaxModes GMM structs. */
static uint8_t zivkovic_pixel(const dp_state* s, float* g, const uint8_t* pixel, uint8_t* pModesUsed) {
  enum { SIGMA = 0, MUR = 1, MUG = 2, MUB = 3, WEIGHT = 4, F = 5 };
  const float m_bg_threshold = 0.75f, m_variance = 36.0f, m_complexity_prior = 0.05f; /* :66-69 */
  const float Alpha = s->alpha;
  int bFitsPDF = 0, bBackgroundHigh = 0;
  float fOneMinAlpha = 1 - Alpha;              /* :113 */
  float prune = -Alpha * m_complexity_prior;   /* :115 */
  int nModes = *pModesUsed;
  float totalWeight = 0.0f;
  int backgroundGaussians = 0; /* :121-134 */
  double sum = 0.0;
  for (int i = 0; i < nModes; ++i) {
    if (sum < m_bg_threshold) {
      backgroundGaussians++;
      sum += g[i * F + WEIGHT];
    } else
      break;
  }
  for (int iModes = 0; iModes < nModes; iModes++) { /* :137-255; nModes shrinks inside the loop, as in the source */
    float* m = g + iModes * F;
    float weight = m[WEIGHT];
    if (!bFitsPDF) {
      float var = m[SIGMA], muR = m[MUR], muG = m[MUG], muB = m[MUB];
      float dR = muR - pixel[0], dG = muG - pixel[1], dB = muB - pixel[2];
      float dist = (dR * dR + dG * dG + dB * dB);
      if (dist < s->high * var && iModes < backgroundGaussians) bBackgroundHigh = 1; /* :160-161 */
      if (dist < s->low * var) { /* :164 */
        bFitsPDF = 1;
        float k = Alpha / weight; /* :175 */
        weight = fOneMinAlpha * weight + prune;
        weight += Alpha;
        m[WEIGHT] = weight;
        m[MUR] = muR - k * (dR);
        m[MUG] = muG - k * (dG);
        m[MUB] = muB - k * (dB);
        float sigmanew = var + k * (dist - var); /* :189 */
        m[SIGMA] = sigmanew < 4 ? 4 : sigmanew > 5 * m_variance ? 5 * m_variance : sigmanew; /* :192 */
        for (int iLocal = iModes; iLocal > 0; iLocal--) { /* :219-234 */
          float* a = g + iLocal * F;
          float* b = g + (iLocal - 1) * F;
          if (a[WEIGHT] > b[WEIGHT]) {
            float t[5];
            memcpy(t, a, sizeof(t)), memcpy(a, b, sizeof(t)), memcpy(b, t, sizeof(t));
          } else
            break;
        }
      } else { /* :236-247 */
        weight = fOneMinAlpha * weight + prune;
        if (weight < -prune) {
          weight = 0.0;
          nModes--;
        }
        m[WEIGHT] = weight;
      }
    } else { /* :252-263 */
      weight = fOneMinAlpha * weight + prune;
      if (weight < -prune) {
        weight = 0.0;
        nModes--;
      }
      m[WEIGHT] = weight;
    }
    totalWeight += weight;
  }
  for (int iLocal = 0; iLocal < nModes; iLocal++) g[iLocal * F + WEIGHT] = g[iLocal * F + WEIGHT] / totalWeight; /* :259-263 */
  if (!bFitsPDF) { /* :266-346 */
    if (nModes == s->K) {
      /* replace the weakest */
    } else
      nModes++;
    float* m = g + (nModes - 1) * F;
    if (nModes == 1)
      m[WEIGHT] = 1;
    else
      m[WEIGHT] = Alpha;
    float sum2 = 0.0;
    for (int iLocal = 0; iLocal < nModes; iLocal++) sum2 += g[iLocal * F + WEIGHT];
    float invSum = 1.0f / sum2;
    for (int iLocal = 0; iLocal < nModes; iLocal++) g[iLocal * F + WEIGHT] *= invSum;
    m[MUR] = pixel[0], m[MUG] = pixel[1], m[MUB] = pixel[2];
    m[SIGMA] = m_variance;
    for (int iLocal = nModes - 1; iLocal > 0; iLocal--) { /* :331-345 */
      float* a = g + iLocal * F;
      float* b = g + (iLocal - 1) * F;
      if (a[WEIGHT] > b[WEIGHT]) {
        float t[5];
        memcpy(t, a, sizeof(t)), memcpy(a, b, sizeof(t)), memcpy(b, t, sizeof(t));
      } else
        break;
    }
  }
  *pModesUsed = (uint8_t)nModes;
  return bBackgroundHigh ? DP_BACKGROUND : DP_FOREGROUND; /* :358-365 */
}

/* GrimsonGMM::SubtractPixel, dp/GrimsonGMM.cpp:119-295.  qsort(compareGMM) orders by `significants`, largest first;
 * glibc's qsort is a merge sort for arrays this small, i.e. stable: restated as a stable insertion sort.
 * sqrt(float) resolves to the float overload (the file is C++ and <cmath> is in scope through OpenCV's headers). */
static void grimson_sort(float* g, int numModes) {
  enum { SIG = 5, F = 6 };
  for (int i = 1; i < numModes; ++i) {
    float t[6];
    memcpy(t, g + i * F, sizeof(t));
    int j = i - 1;
    while (j >= 0 && g[j * F + SIG] < t[SIG]) { /* compareGMM(a, b) > 0 <=> a.significants < b.significants */
      memcpy(g + (j + 1) * F, g + j * F, sizeof(t));
      j--;
    }
    memcpy(g + (j + 1) * F, t, sizeof(t));
  }
}

static uint8_t grimson_pixel(const dp_state* s, float* g, const uint8_t* pixel, uint8_t* pNumModes) {
  enum { VAR = 0, MUR = 1, MUG = 2, MUB = 3, WEIGHT = 4, SIG = 5, F = 6 };
  const float m_bg_threshold = 0.75f, m_variance = 36.0f; /* GrimsonGMM.cpp:77-78 */
  const float Alpha = s->alpha;
  int numModes = *pNumModes;
  int bFitsPDF = 0, bBackgroundHigh = 0;
  float fOneMinAlpha = 1 - Alpha;
  float totalWeight = 0.0f;
  int backgroundGaussians = 0;
  double sum = 0.0;
  for (int i = 0; i < numModes; ++i) {
    if (sum < m_bg_threshold) {
      backgroundGaussians++;
      sum += g[i * F + WEIGHT];
    } else
      break;
  }
  for (int iModes = 0; iModes < numModes; iModes++) {
    float* m = g + iModes * F;
    float weight = m[WEIGHT];
    if (!bFitsPDF) {
      float var = m[VAR], muR = m[MUR], muG = m[MUG], muB = m[MUB];
      float dR = muR - pixel[0], dG = muG - pixel[1], dB = muB - pixel[2];
      float dist = (dR * dR + dG * dG + dB * dB);
      if (dist < s->high * var && iModes < backgroundGaussians) bBackgroundHigh = 1;
      if (dist < s->low * var) {
        bFitsPDF = 1;
        float k = Alpha / weight;
        weight = fOneMinAlpha * weight + Alpha;
        m[WEIGHT] = weight;
        m[MUR] = muR - k * (dR);
        m[MUG] = muG - k * (dG);
        m[MUB] = muB - k * (dB);
        float sigmanew = var + k * (dist - var);
        m[VAR] = sigmanew < 4 ? 4 : sigmanew > 5 * m_variance ? 5 * m_variance : sigmanew;
        m[SIG] = m[WEIGHT] / sqrtf(m[VAR]);
      } else {
        weight = fOneMinAlpha * weight;
        if (weight < 0.0) {
          weight = 0.0;
          numModes--;
        }
        m[WEIGHT] = weight;
        m[SIG] = m[WEIGHT] / sqrtf(m[VAR]);
      }
    } else {
      weight = fOneMinAlpha * weight;
      if (weight < 0.0) {
        weight = 0.0;
        numModes--;
      }
      m[WEIGHT] = weight;
      m[SIG] = m[WEIGHT] / sqrtf(m[VAR]);
    }
    totalWeight += weight;
  }
  double invTotalWeight = 1.0 / totalWeight;
  for (int iLocal = 0; iLocal < numModes; iLocal++) {
    g[iLocal * F + WEIGHT] *= (float)invTotalWeight;
    g[iLocal * F + SIG] = g[iLocal * F + WEIGHT] / sqrtf(g[iLocal * F + VAR]);
  }
  grimson_sort(g, numModes);
  if (!bFitsPDF) {
    if (numModes < s->K) numModes++;
    float* m = g + (numModes - 1) * F;
    m[MUR] = pixel[0], m[MUG] = pixel[1], m[MUB] = pixel[2];
    m[VAR] = m_variance;
    m[SIG] = 0;
    if (numModes == 1)
      m[WEIGHT] = 1;
    else
      m[WEIGHT] = Alpha;
    float sum2 = 0.0;
    for (int iLocal = 0; iLocal < numModes; iLocal++) sum2 += g[iLocal * F + WEIGHT];
    double invSum = 1.0 / sum2;
    for (int iLocal = 0; iLocal < numModes; iLocal++) {
      g[iLocal * F + WEIGHT] *= (float)invSum;
      g[iLocal * F + SIG] = g[iLocal * F + WEIGHT] / sqrtf(g[iLocal * F + VAR]);
    }
  }
  grimson_sort(g, numModes);
  *pNumModes = (uint8_t)numModes;
  return bBackgroundHigh ? DP_BACKGROUND : DP_FOREGROUND;
}

int dp_process(dp_state* s, const uint8_t* img, int64_t frame_num, uint8_t* fg) {
  const size_t n = s->n;
  switch (s->algo) {
    case BGS_DP_ZIVKOVIC_AGMM: /* ZivkovicAGMM::Subtract :376-407; Update is empty (:96-99) */
      for (size_t i = 0; i < n; ++i) fg[i] = zivkovic_pixel(s, s->modes + i * s->K * 5, img + i * 3, s->nmodes + i);
      break;
    case BGS_DP_GRIMSON_GMM:
      for (size_t i = 0; i < n; ++i) fg[i] = grimson_pixel(s, s->modes + i * s->K * 6, img + i * 3, s->nmodes + i);
      break;
    case BGS_DP_WREN_GA:
      for (size_t i = 0; i < n; ++i) {
        float* gm = s->gauss + i * 4;
        const uint8_t* px = img + i * 3;
        float dist = 0; /* SubtractPixel, WrenGA.cpp:113-134 */
        for (int ch = 0; ch < 3; ++ch) {
          float delta = gm[ch] - px[ch];
          dist += delta * delta;
        }
        fg[i] = dist > s->high * gm[3] ? DP_FOREGROUND : DP_BACKGROUND;
        /* Update with an all-BACKGROUND mask, WrenGA.cpp:79-111 */
        float dR = gm[0] - px[0], dG = gm[1] - px[1], dB = gm[2] - px[2];
        float d2 = (dR * dR + dG * dG + dB * dB);
        gm[0] -= s->alpha * (dR);
        gm[1] -= s->alpha * (dG);
        gm[2] -= s->alpha * (dB);
        float sigmanew = gm[3] + s->alpha * (d2 - gm[3]);
        gm[3] = sigmanew < 4 ? 4 : sigmanew > 5 * 36.0f ? 5 * 36.0f : sigmanew;
      }
      break;
    case BGS_DP_MEAN:
      for (size_t i = 0; i < n; ++i) {
        float* mean = s->mean + i * 3;
        const uint8_t* px = img + i * 3;
        float dist = 0; /* SubtractPixel, MeanBGS.cpp:77-98 */
        for (int ch = 0; ch < 3; ++ch) dist += (px[ch] - mean[ch]) * (px[ch] - mean[ch]);
        fg[i] = dist > s->high ? DP_FOREGROUND : DP_BACKGROUND;
        for (int ch = 0; ch < 3; ++ch) mean[ch] = s->alpha * mean[ch] + (1.0f - s->alpha) * px[ch]; /* Update :52-75 */
      }
      break;
    case BGS_DP_ADAPTIVE_MEDIAN: {
      const int update = (frame_num % s->sampling_rate) == 1; /* AdaptiveMedianBGS.cpp:60 */
      for (size_t i = 0; i < n; ++i) {
        uint8_t* med = s->median + i * 3;
        const uint8_t* px = img + i * 3;
        int diffR = abs(px[0] - med[0]), diffG = abs(px[1] - med[1]), diffB = abs(px[2] - med[2]); /* :92-108 */
        fg[i] = (diffR <= s->high && diffG <= s->high && diffB <= s->high) ? DP_BACKGROUND : DP_FOREGROUND;
        if (update)
          for (int ch = 0; ch < 3; ++ch) {
            if (px[ch] > med[ch])
              med[ch]++;
            else if (px[ch] < med[ch])
              med[ch]--;
          }
      }
      break;
    }
    default: return BGS_ERR_UNSUPPORTED;
  }
  return BGS_OK;
}

/* canonical SoA export, same names and order as bgs_get_state */
int64_t dp_get_state(dp_state* s, const char* plane, void* dst, size_t cap) {
  const size_t n = s->n;
  if (s->modes && !strcmp(plane, "modes")) {
    const int F = s->algo == BGS_DP_ZIVKOVIC_AGMM ? 5 : 6, P = s->K * F;
    if (cap < n * P * 4) return BGS_ERR_STATE;
    for (int q = 0; q < P; ++q)
      for (size_t i = 0; i < n; ++i) ((float*)dst)[(size_t)q * n + i] = s->modes[i * P + q];
    return (int64_t)(n * P * 4);
  }
  if (s->nmodes && !strcmp(plane, "nmodes")) {
    if (cap < n) return BGS_ERR_STATE;
    memcpy(dst, s->nmodes, n);
    return (int64_t)n;
  }
  if (s->gauss && !strcmp(plane, "gauss")) {
    if (cap < n * 16) return BGS_ERR_STATE;
    for (int q = 0; q < 4; ++q)
      for (size_t i = 0; i < n; ++i) ((float*)dst)[(size_t)q * n + i] = s->gauss[i * 4 + q];
    return (int64_t)(n * 16);
  }
  if (s->mean && !strcmp(plane, "mean")) {
    if (cap < n * 12) return BGS_ERR_STATE;
    for (int q = 0; q < 3; ++q)
      for (size_t i = 0; i < n; ++i) ((float*)dst)[(size_t)q * n + i] = s->mean[i * 3 + q];
    return (int64_t)(n * 12);
  }
  if (s->median && !strcmp(plane, "median")) {
    if (cap < n * 3) return BGS_ERR_STATE;
    memcpy(dst, s->median, n * 3);
    return (int64_t)(n * 3);
  }
  return BGS_ERR_STATE;
}
