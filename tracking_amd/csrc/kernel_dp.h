// kernel_dp.h — SURVEY.md N4: the in-tree package_bgs/dp/ background models, one fused kernel each.
//
//   dp_gmm_kernel<K, false>  ZivkovicAGMM::SubtractPixel  package_bgs/dp/ZivkovicAGMM.cpp:103-364  (DPZivkovicAGMMBGS.cpp:29-80)
//   dp_gmm_kernel<K, true>   GrimsonGMM::SubtractPixel    package_bgs/dp/GrimsonGMM.cpp:119-295    (DPGrimsonGMMBGS.cpp:29-82)
//   dp_wren_kernel           WrenGA::SubtractPixel + Update  package_bgs/dp/WrenGA.cpp:79-134       (DPWrenGABGS.cpp:29-81)
//   dp_mean_kernel           MeanBGS::SubtractPixel + Update package_bgs/dp/MeanBGS.cpp:52-98       (DPMeanBGS.cpp:29-82)
//   dp_median_kernel         AdaptiveMedianBGS::SubtractPixel + Update  dp/AdaptiveMedianBGS.cpp:58-108 (DPAdaptiveMedianBGS.cpp:29-81)
//
// The wrappers return the HIGH-threshold mask (2 x threshold), clear the update mask before Update (so the update
// always fires) and never write a background image.  Pixel bytes are taken in memory order (pixel(0) = first byte).
// State: tiled AoSoA over the engine's global pixel index g = stream*n + i (the layout that took MOG2 from 5.3 to 6.1 TB/s,
// DESIGN.md §6.2): tile = 256 pixels x all planes, plane q of pixel g at state[((g >> 8)*planes + q)*256 + (g & 255)], so a
// workgroup's whole working set is one contiguous block and every access is a coalesced 1 KiB row; workgroups are
// renumbered so that each XCD walks one contiguous eighth of the model (xcd_block).
// The sources' control flow is kept statement by statement (mode count that shrinks inside the loop, adjacent-swap
// sorts, stale entries past the count left untouched) with all indices static after unrolling.
#pragma once
#include "bgs_device.h"

namespace bgs {

struct DpArgs {
  const uint8_t* frame;  // [count][n][3]
  float* state;          // [tiles][planes][256]   (GMMs, WrenGA, Mean)
  uint8_t* bstate;       // nmodes [S][n]  |  median [S][n][3]
  uint8_t* fg;           // [count][n] or null
  uint64_t* fg_bits;     // [count][n/64] or null
  size_t n;              // pixels per stream
  size_t npix;           // pixels of this launch (count * n)
  int first;             // first stream of this launch
  float low, high, alpha;
  int update;            // AdaptiveMedian: frame_num % samplingRate == 1
  int xcd_swizzle;
  // clip launches of the GMM kernels (bgs_process_clip_device): `frames` consecutive frames per launch, the model in registers in
  // between; frame / fg / fg_bits point at the first one, the strides lead to the next (bits_stride in 64-bit words).  0 = 1 frame.
  int frames;
  size_t frame_stride, fg_stride, bits_stride;
};

constexpr int kDpTile = 256;
// plane 0 of global pixel g; plane q is q*kDpTile floats further
__device__ __forceinline__ float* dp_plane0(const DpArgs& a, size_t gp, int planes) {
  const size_t g = (size_t)a.first * a.n + gp;
  return a.state + (g / kDpTile) * (size_t)planes * kDpTile + (g % kDpTile);
}

__device__ __forceinline__ void dp_store_mask(const DpArgs& a, size_t gp, bool active, int m, int t = 0) {
  if (active && a.fg) a.fg[(size_t)t * a.fg_stride + gp] = (uint8_t)m;
  if (a.fg_bits) {  // npix % 64 == 0 is checked on the host: a wave is either all active or all idle
    const unsigned long long w = __ballot(active && m != 0);
    if ((threadIdx.x & (kWave - 1)) == 0 && active) a.fg_bits[(size_t)t * a.bits_stride + (gp >> 6)] = w;
  }
}

template <int K, bool GRIMSON>
__global__ __launch_bounds__(kBlock) void dp_gmm_kernel(const DpArgs a) {
  constexpr int F = GRIMSON ? 6 : 5;
  // field order inside a mode = the source's struct: Zivkovic {sigma, muR, muG, muB, weight}, Grimson {variance, muR, muG, muB, weight, significants}
  constexpr int VAR = 0, MU = 1, WEIGHT = 4, SIG = 5;
  const size_t gp = xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x;
  const bool active = gp < a.npix;
  const int T = a.frames > 0 ? a.frames : 1;  // frames of this launch (clip launches: the model stays in registers in between)
  float* st = nullptr;
  uint8_t* pn = nullptr;
  int nModes = 0, nLoaded = 0, nUsed = 0;
  float g[K][F];
  uint32_t g0[K][F];
  uint32_t pw = 0;
  if (active) {
    st = dp_plane0(a, gp, K * F);
    pn = a.bstate + (size_t)a.first * a.n + gp;
    pw = (uint32_t)a.frame[gp * 3] | ((uint32_t)a.frame[gp * 3 + 1] << 8) | ((uint32_t)a.frame[gp * 3 + 2] << 16);
    // Data-dependent traffic (exact): the sources never read a mode at an index >= the pixel's count (they only create one
    // there, writing every field), so only the used modes are loaded; at the end a field is stored if its bits changed, or
    // unconditionally when the slot was not loaded and is now in use.  One lane = one pixel: no neighbour shares the slot.
    nModes = *pn;
    nLoaded = nModes;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        g[k][f] = k < nLoaded ? st[(k * F + f) * kDpTile] : 0.f;
        g0[k][f] = __float_as_uint(g[k][f]);
      }
  }
  for (int t = 0; t < T; ++t) {
  int mask = 0;
  if (active) {
    const float px[3] = {(float)(pw & 0xffu), (float)((pw >> 8) & 0xffu), (float)(pw >> 16)};
    if (t + 1 < T) {  // the next frame's pixel, requested before this frame's arithmetic
      const uint8_t* nf = a.frame + (size_t)(t + 1) * a.frame_stride + gp * 3;
      pw = (uint32_t)nf[0] | ((uint32_t)nf[1] << 8) | ((uint32_t)nf[2] << 16);
    }
    const float m_bg_threshold = 0.75f, m_variance = 36.0f, m_complexity_prior = 0.05f;
    const float Alpha = a.alpha;
    bool bFitsPDF = false, bBackgroundHigh = false;
    const float fOneMinAlpha = 1 - Alpha;
    const float prune = -Alpha * m_complexity_prior;  // Zivkovic only
    float totalWeight = 0.0f;
    int backgroundGaussians = 0;
    {
      double sum = 0.0;
      bool stop = false;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < nModes && !stop) {
          if (sum < (double)m_bg_threshold) {
            backgroundGaussians++;
            sum = __dadd_rn(sum, (double)g[k][WEIGHT]);
          } else
            stop = true;
        }
    }
    auto swap_modes = [&](float(&x)[F], float(&y)[F]) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float t = x[f];
        x[f] = y[f], y[f] = t;
      }
    };
#pragma unroll
    for (int im = 0; im < K; ++im) {
      if (im < nModes) {  // nModes shrinks inside the loop, as in the source
        float weight = g[im][WEIGHT];
        if (!bFitsPDF) {
          const float var = g[im][VAR], muR = g[im][MU], muG = g[im][MU + 1], muB = g[im][MU + 2];
          const float dR = muR - px[0], dG = muG - px[1], dB = muB - px[2];
          const float dist = (dR * dR + dG * dG + dB * dB);
          if (dist < a.high * var && im < backgroundGaussians) bBackgroundHigh = true;
          if (dist < a.low * var) {
            bFitsPDF = true;
            const float k = div_rn(Alpha, weight);
            if constexpr (GRIMSON) {
              weight = fOneMinAlpha * weight + Alpha;
            } else {
              weight = fOneMinAlpha * weight + prune;
              weight += Alpha;
            }
            g[im][WEIGHT] = weight;
            g[im][MU] = muR - k * (dR);
            g[im][MU + 1] = muG - k * (dG);
            g[im][MU + 2] = muB - k * (dB);
            const float sigmanew = var + k * (dist - var);
            g[im][VAR] = sigmanew < 4 ? 4 : sigmanew > 5 * m_variance ? 5 * m_variance : sigmanew;
            if constexpr (GRIMSON) {
              g[im][SIG] = div_rn(g[im][WEIGHT], sqrt_rn(g[im][VAR]));
            } else {
              bool stop = false;  // ZivkovicAGMM.cpp:219-234: move the grown mode up while it outweighs its predecessor
#pragma unroll
              for (int il = im; il > 0; --il)
                if (!stop) {
                  if (g[il][WEIGHT] > g[il - 1][WEIGHT])
                    swap_modes(g[il], g[il - 1]);
                  else
                    stop = true;
                }
            }
          } else {
            if constexpr (GRIMSON) {
              weight = fOneMinAlpha * weight;
              if (weight < 0.0f) weight = 0.0f, nModes--;
              g[im][WEIGHT] = weight;
              g[im][SIG] = div_rn(weight, sqrt_rn(g[im][VAR]));
            } else {
              weight = fOneMinAlpha * weight + prune;
              if (weight < -prune) weight = 0.0f, nModes--;
              g[im][WEIGHT] = weight;
            }
          }
        } else {
          if constexpr (GRIMSON) {
            weight = fOneMinAlpha * weight;
            if (weight < 0.0f) weight = 0.0f, nModes--;
            g[im][WEIGHT] = weight;
            g[im][SIG] = div_rn(weight, sqrt_rn(g[im][VAR]));
          } else {
            weight = fOneMinAlpha * weight + prune;
            if (weight < -prune) weight = 0.0f, nModes--;
            g[im][WEIGHT] = weight;
          }
        }
        totalWeight += weight;
      }
    }
    // stable insertion by adjacent swaps, largest `significants` first (qsort + compareGMM, GrimsonGMM.cpp:45-56)
    auto grimson_sort = [&](int count) {
#pragma unroll
      for (int i2 = 1; i2 < K; ++i2)
        if (i2 < count) {
          bool stop = false;
#pragma unroll
          for (int j = i2; j > 0; --j)
            if (!stop) {
              if (g[j - 1][F - 1] < g[j][F - 1])
                swap_modes(g[j - 1], g[j]);
              else
                stop = true;
            }
        }
    };
    if constexpr (GRIMSON) {
      const double invTotalWeight = 1.0 / (double)totalWeight;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < nModes) {
          g[k][WEIGHT] *= (float)invTotalWeight;
          g[k][SIG] = div_rn(g[k][WEIGHT], sqrt_rn(g[k][VAR]));
        }
      grimson_sort(nModes);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < nModes) g[k][WEIGHT] = div_rn(g[k][WEIGHT], totalWeight);
    }
    if (!bFitsPDF) {
      if (nModes < K) nModes++;  // else: replace the weakest (the last one)
      const int last = nModes - 1;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k == last) {
          if constexpr (GRIMSON) {
            g[k][MU] = px[0], g[k][MU + 1] = px[1], g[k][MU + 2] = px[2];
            g[k][VAR] = m_variance;
            g[k][SIG] = 0;
          }
          g[k][WEIGHT] = nModes == 1 ? 1.0f : Alpha;
        }
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < nModes) sum += g[k][WEIGHT];
      if constexpr (GRIMSON) {
        const double invSum = 1.0 / (double)sum;
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (k < nModes) {
            g[k][WEIGHT] *= (float)invSum;
            g[k][SIG] = div_rn(g[k][WEIGHT], sqrt_rn(g[k][VAR]));
          }
      } else {
        const float invSum = div_rn(1.0f, sum);
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (k < nModes) g[k][WEIGHT] *= invSum;
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (k == last) g[k][MU] = px[0], g[k][MU + 1] = px[1], g[k][MU + 2] = px[2], g[k][VAR] = m_variance;
        bool stop = false;  // ZivkovicAGMM.cpp:331-345
#pragma unroll
        for (int il = K - 1; il > 0; --il)
          if (il <= last && !stop) {
            if (g[il][WEIGHT] > g[il - 1][WEIGHT])
              swap_modes(g[il], g[il - 1]);
            else
              stop = true;
          }
      }
    }
    if constexpr (GRIMSON) grimson_sort(nModes);  // the second qsort runs whether or not a mode was added (:281)
    mask = bBackgroundHigh ? 0 : 255;
    nUsed = max(nUsed, nModes);
  }
  dp_store_mask(a, gp, active, mask, t);
  }
  if (active) {
    // a slot that was not loaded is written if it was in use at the end of ANY frame of the launch: the source wrote it then, and a
    // later frame that prunes the mode again leaves those values behind the count
    if (nModes != nLoaded) *pn = (uint8_t)nModes;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int f = 0; f < F; ++f)
        if (k < nLoaded ? __float_as_uint(g[k][f]) != g0[k][f] : k < nUsed) st[(k * F + f) * kDpTile] = g[k][f];
  }
}

__global__ __launch_bounds__(kBlock) void dp_wren_kernel(const DpArgs a) {
  const size_t gp = xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x;
  const bool active = gp < a.npix;
  int mask = 0;
  if (active) {
    float* st = dp_plane0(a, gp, 4);
    float mu[3] = {st[0], st[kDpTile], st[2 * kDpTile]}, var = st[3 * kDpTile];
    const float px[3] = {(float)a.frame[gp * 3], (float)a.frame[gp * 3 + 1], (float)a.frame[gp * 3 + 2]};
    float dist = 0;  // SubtractPixel, WrenGA.cpp:113-134
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float delta = mu[ch] - px[ch];
      dist += delta * delta;
    }
    mask = dist > a.high * var ? 255 : 0;
    const float dR = mu[0] - px[0], dG = mu[1] - px[1], dB = mu[2] - px[2];  // Update, :79-111
    const float d2 = (dR * dR + dG * dG + dB * dB);
    mu[0] -= a.alpha * (dR), mu[1] -= a.alpha * (dG), mu[2] -= a.alpha * (dB);
    const float sigmanew = var + a.alpha * (d2 - var);
    var = sigmanew < 4 ? 4 : sigmanew > 5 * 36.0f ? 5 * 36.0f : sigmanew;
    st[0] = mu[0], st[kDpTile] = mu[1], st[2 * kDpTile] = mu[2], st[3 * kDpTile] = var;
  }
  dp_store_mask(a, gp, active, mask);
}

__global__ __launch_bounds__(kBlock) void dp_mean_kernel(const DpArgs a) {
  const size_t gp = xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x;
  const bool active = gp < a.npix;
  int mask = 0;
  if (active) {
    float* st = dp_plane0(a, gp, 3);
    float dist = 0;  // SubtractPixel, MeanBGS.cpp:77-98
    float mean[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      mean[ch] = st[ch * kDpTile];
      const float px = (float)a.frame[gp * 3 + ch];
      dist += (px - mean[ch]) * (px - mean[ch]);
      st[ch * kDpTile] = a.alpha * mean[ch] + (1.0f - a.alpha) * px;  // Update, :52-75
    }
    mask = dist > a.high ? 255 : 0;
  }
  dp_store_mask(a, gp, active, mask);
}

// G = 4: a lane owns 4 consecutive pixels (3 dwords of frame, 3 dwords of median, 1 dword of mask); G = 1 for sizes /
// pointers that are not 4-aligned
template <int G>
__global__ __launch_bounds__(kBlock) void dp_median_kernel(const DpArgs a) {
  const size_t gp = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = gp < a.npix;
  uint32_t mask4 = 0;
  if (active) {
    uint8_t* med = a.bstate + ((size_t)a.first * a.n + gp) * 3;
    uint8_t fb[G * 3], mb[G * 3];
    if constexpr (G == 4) {
      const uint32_t* fw = reinterpret_cast<const uint32_t*>(a.frame + gp * 3);
      const uint32_t* mw = reinterpret_cast<const uint32_t*>(med);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const uint32_t f = fw[k], m = mw[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[4 * k + j] = (uint8_t)(f >> (8 * j)), mb[4 * k + j] = (uint8_t)(m >> (8 * j));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) fb[j] = a.frame[gp * 3 + j], mb[j] = med[j];
    }
#pragma unroll
    for (int o = 0; o < G; ++o) {
      bool bgd = true;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int px = fb[o * 3 + ch], m = mb[o * 3 + ch];
        bgd = bgd && (float)abs(px - m) <= a.high;  // AdaptiveMedianBGS.cpp:92-108
        mb[o * 3 + ch] = (uint8_t)(px > m ? m + 1 : px < m ? m - 1 : m);  // :58-84 (stored only on update frames)
      }
      mask4 |= (bgd ? 0u : 255u) << (8 * o);
    }
    if (a.update) {
      if constexpr (G == 4) {
        uint32_t* mw = reinterpret_cast<uint32_t*>(med);
#pragma unroll
        for (int k = 0; k < 3; ++k) mw[k] = (uint32_t)mb[4 * k] | ((uint32_t)mb[4 * k + 1] << 8) | ((uint32_t)mb[4 * k + 2] << 16) | ((uint32_t)mb[4 * k + 3] << 24);
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) med[j] = mb[j];
      }
    }
    if (a.fg) {
      if constexpr (G == 4)
        *reinterpret_cast<uint32_t*>(a.fg + gp) = mask4;
      else
        a.fg[gp] = (uint8_t)mask4;
    }
  }
  if (a.fg_bits) {
    if constexpr (G == 4) {
      const uint32_t nib = (mask4 & 1u) | ((mask4 >> 7) & 2u) | ((mask4 >> 14) & 4u) | ((mask4 >> 21) & 8u);
      store_packed_mask<4>(a.fg_bits, gp, nib, active);
    } else {
      const unsigned long long w = __ballot(active && mask4 != 0);
      if ((threadIdx.x & (kWave - 1)) == 0 && active) a.fg_bits[gp >> 6] = w;
    }
  }
}

// Zivkovic / Grimson InitModel (dp/ZivkovicAGMM.cpp:77-97, GrimsonGMM.cpp:93-117): every mode field and the mode count of
// the launch's pixels = 0.  Runs on the launch stream at a stream's first frame, like mog2_clear_kernel / mog1_clear_kernel.
__global__ __launch_bounds__(kBlock) void dp_gmm_clear_kernel(const DpArgs a, int planes) {
  const size_t gp = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (gp >= a.npix) return;
  float* st = dp_plane0(a, gp, planes);
  for (int q = 0; q < planes; ++q) st[q * kDpTile] = 0.f;
  a.bstate[(size_t)a.first * a.n + gp] = 0;
}

// WrenGA / Mean: model = the first frame (InitModel); one lane per pixel
__global__ __launch_bounds__(kBlock) void dp_init_kernel(const DpArgs a, int planes, float var0) {
  const size_t gp = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (gp >= a.npix) return;
  float* st = dp_plane0(a, gp, planes);
  for (int ch = 0; ch < 3; ++ch) st[ch * kDpTile] = (float)a.frame[gp * 3 + ch];
  if (planes == 4) st[3 * kDpTile] = var0;
}

}  // namespace bgs
