// kernel_mog1.h — K5: MixtureOfGaussianV1BGS (cv::BackgroundSubtractorMOG, KaewTraKulPong-Bowden) update + classify.
//
// Replaces  MixtureOfGaussianV1BGS::process  package_bgs/MixtureOfGaussianV1BGS.cpp:51-56
// Algorithm: OpenCV 2.4 bgfg_gaussmix.cpp process8uC3 / process8uC1: scan the modes in sortKey
// order until an empty one (w < FLT_EPSILON) or a match (d2 < vT * sum(var)); update the match and bubble it up by
// sortKey = w_old / sqrt(sum var); otherwise replace the weakest; renormalise; foreground iff the hit lies past the
// background prefix (cumulative weight > T).
//
// Layout: tiled AoSoA like MOG2 (kernel_mog2.h): tile = 256 pixels x NP planes, NP = K*(2+2C) floats per pixel
// (40 for BGR, 20 for gray); plane index of mode k: sortKey = k*R, weight = k*R+1, mean[c] = k*R+2+c, var[c] = k*R+2+C+c.
// One lane owns PX = 2 consecutive pixels (8-byte accesses; 40 floats x 2 px already fill 80 VGPRs).
// Algorithmic traffic (BGR): r 3 + 160, w 160 + 1 = 324 B/pixel/frame — HBM-bound, no MFMA, no LDS.
#pragma once
#include <cfloat>

#include "bgs_device.h"

namespace bgs {

constexpr int kMog1K = 5;
constexpr int kMog1Tile = 256;

struct Mog1Args {
  const uint8_t* frame;  // [P][C]
  uint8_t* fg;           // [P] or null
  uint64_t* fg_bits;     // [P/64] or null
  float* state;          // tiles of 256 px x NP planes
  size_t state_off, npix;
  float alpha, T, vT, w0, sk0, var0, minVar;
  int thr, enable_thr, packed, xcd_swizzle;
};

template <int C>
struct Mog1Px {
  float sk[kMog1K], w[kMog1K], mu[kMog1K][C], var[kMog1K][C];
};

template <int C>
__device__ __forceinline__ void mog1_swap(Mog1Px<C>& s, int i, int j) {
  float t;
  t = s.sk[i], s.sk[i] = s.sk[j], s.sk[j] = t;
  t = s.w[i], s.w[i] = s.w[j], s.w[j] = t;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    t = s.mu[i][c], s.mu[i][c] = s.mu[j][c], s.mu[j][c] = t;
    t = s.var[i][c], s.var[i][c] = s.var[j][c], s.var[j][c] = t;
  }
}

template <int C>
__device__ __forceinline__ float mog1_varsum(const Mog1Px<C>& s, int k) {
  if constexpr (C == 3)
    return s.var[k][0] + s.var[k][1] + s.var[k][2];
  else
    return s.var[k][0];
}

// one pixel, same statement order as process8uC3 so every float rounds identically; returns 0 / 255
template <int C>
__device__ __forceinline__ int mog1_pixel(Mog1Px<C>& s, const float (&pix)[C], const Mog1Args& a, const float alpha) {
  constexpr int K = kMog1K;
  int kHit = -1, kForeground = -1;
  if (alpha > 0) {
    float wsum = 0;
    bool done = false;
    int k_end = K;  // value of the scan index when the reference's first loop exits
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!done) {
        const float w = s.w[k];
        wsum += w;
        if (w < FLT_EPSILON) {
          done = true, k_end = k;
        } else {
          float diff[C], d2 = 0;
#pragma unroll
          for (int c = 0; c < C; ++c) diff[c] = pix[c] - s.mu[k][c], d2 += diff[c] * diff[c];
          if (d2 < a.vT * mog1_varsum<C>(s, k)) {
            wsum -= w;
            const float dw = alpha * (1.f - w);
            s.w[k] = w + dw;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              s.mu[k][c] = s.mu[k][c] + alpha * diff[c];
              const float v = s.var[k][c] + alpha * (diff[c] * diff[c] - s.var[k][c]);
              s.var[k][c] = v > a.minVar ? v : a.minVar;
            }
            s.sk[k] = div_rn(w, sqrt_rn(mog1_varsum<C>(s, k)));  // sic: the OLD weight
            bool moving = true;
            int pos = k;
#pragma unroll
            for (int k1 = k - 1; k1 >= 0; --k1) {
              moving = moving && !(s.sk[k1] >= s.sk[k1 + 1]);
              if (moving) mog1_swap<C>(s, k1, k1 + 1), pos = k1;
            }
            kHit = pos;
            done = true, k_end = k;
          }
        }
      }
    }
    if (kHit < 0) {  // no match: replace the first empty mode, else the last one
      const int kk = k_end < K - 1 ? k_end : K - 1;
      kHit = kk;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k == kk) {
          wsum += a.w0 - s.w[k];
          s.w[k] = a.w0;
#pragma unroll
          for (int c = 0; c < C; ++c) s.mu[k][c] = pix[c], s.var[k][c] = a.var0;
          s.sk[k] = a.sk0;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k >= k_end) wsum += s.w[k];
    }
    const float wscale = div_rn(1.f, wsum);
    wsum = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      s.w[k] *= wscale;
      wsum += s.w[k];
      s.sk[k] *= wscale;
      if (wsum > a.T && kForeground < 0) kForeground = k + 1;
    }
    return kHit >= kForeground ? 255 : 0;
  }
  // learning rate 0: classify only
  bool done = false;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (!done) {
      if (s.w[k] < FLT_EPSILON) {
        done = true;
      } else {
        float d2 = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float d = pix[c] - s.mu[k][c];
          d2 += d * d;
        }
        if (d2 < a.vT * mog1_varsum<C>(s, k)) kHit = k, done = true;
      }
    }
  }
  if (kHit >= 0) {
    float wsum = 0;
    bool stop = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!stop) {
        wsum += s.w[k];
        if (wsum > a.T) kForeground = k + 1, stop = true;
      }
    }
  }
  return (kHit < 0 || kHit >= kForeground) ? 255 : 0;
}

template <int C>
__host__ __device__ constexpr int mog1_planes() { return kMog1K * (2 + 2 * C); }

template <int C>
__device__ __forceinline__ size_t mog1_plane_off(int p, size_t sp) {
  return (sp >> 8) * (size_t)(mog1_planes<C>() * kMog1Tile) + (size_t)p * kMog1Tile + (sp & 255);
}

// grid: ceil(npix / PX / kBlock); npix % PX == 0 and state_off % PX == 0 (host picks PX = 1 otherwise)
template <int C, int PX>
__global__ __launch_bounds__(kBlock) void mog1_update_kernel(const Mog1Args a) {
  constexpr int K = kMog1K, R = 2 + 2 * C, NP = K * R;
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {  // each XCD streams one contiguous eighth of the launch (see kernel_mog2.h)
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = (blk * kBlock + threadIdx.x) * PX;
  const bool active = p0 < a.npix;
  uint32_t bits = 0;
  if (active) {
    const size_t sp = a.state_off + p0;
    // Data-dependent traffic (exact).  The reference never reads the mean / variance of a mode whose weight is below
    // FLT_EPSILON (its scan stops there, bgfg_gaussmix.cpp) - it only ever creates a mode in such a slot, writing every
    // field.  So the weight and sort-key planes of all K modes are loaded first, and the 2C mean / variance planes of mode k
    // only if one of this lane's pixels has a live mode k.  A slot the lane did not load is written per pixel (scalar
    // stores) by the pixel that created a mode there, so with PX = 2 the lane's other pixel keeps its stale, unread entries.
    float st[NP][PX];
    unsigned need = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      load_f<PX>(a.state + mog1_plane_off<C>(k * R, sp), st[k * R]);
      load_f<PX>(a.state + mog1_plane_off<C>(k * R + 1, sp), st[k * R + 1]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      bool live = false;
#pragma unroll
      for (int j = 0; j < PX; ++j) live = live || st[k * R + 1][j] >= FLT_EPSILON;
      need |= (unsigned)live << k;
#pragma unroll
      for (int f = 2; f < R; ++f) {
        if (live) {
          load_f<PX>(a.state + mog1_plane_off<C>(k * R + f, sp), st[k * R + f]);
        } else {
#pragma unroll
          for (int j = 0; j < PX; ++j) st[k * R + f][j] = 0.f;
        }
      }
    }
    uint8_t px[PX * C];
#pragma unroll
    for (int i = 0; i < PX * C; ++i) px[i] = a.frame[p0 * C + i];
    uint32_t mword = 0;
    uint64_t dirty = 0;  // NP <= 40 planes
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      Mog1Px<C> s;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        s.sk[k] = st[k * R][j], s.w[k] = st[k * R + 1][j];
#pragma unroll
        for (int c = 0; c < C; ++c) s.mu[k][c] = st[k * R + 2 + c][j], s.var[k][c] = st[k * R + 2 + C + c][j];
      }
      float pix[C];
#pragma unroll
      for (int c = 0; c < C; ++c) pix[c] = (float)px[j * C + c];
      const int m = thr_bin(mog1_pixel<C>(s, pix, a, a.alpha), a.thr, a.enable_thr);
      mword |= (uint32_t)m << (8 * j);
      bits |= (uint32_t)(m != 0) << j;
      // write-back is data-dependent (exact): a plane is stored only if one of this lane's pixels changed its bits.  On a
      // quiet scene one mode matches and only the weights / sort keys of the others move; their means and variances stay.
      auto put = [&](int q, float v) {
        dirty |= (uint64_t)(__float_as_uint(st[q][j]) != __float_as_uint(v)) << q;
        st[q][j] = v;
      };
#pragma unroll
      for (int k = 0; k < K; ++k) {
        put(k * R, s.sk[k]), put(k * R + 1, s.w[k]);
#pragma unroll
        for (int c = 0; c < C; ++c) put(k * R + 2 + c, s.mu[k][c]), put(k * R + 2 + C + c, s.var[k][c]);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int f = 0; f < R; ++f) {
        const int q = k * R + f;
        if (f < 2 || ((need >> k) & 1u)) {
          if ((dirty >> q) & 1ull) store_f<PX>(a.state + mog1_plane_off<C>(q, sp), st[q]);
        } else {
          // mean / variance of a slot this lane did not load: the pixel that now owns a mode there writes its own element,
          // whatever the value (a created field may equal the zero the register was filled with)
#pragma unroll
          for (int j = 0; j < PX; ++j)
            if (st[k * R + 1][j] >= FLT_EPSILON) a.state[mog1_plane_off<C>(q, sp) + j] = st[q][j];
        }
      }
    }
    if (a.fg) {
#pragma unroll
      for (int j = 0; j < PX; ++j) a.fg[p0 + j] = (uint8_t)(mword >> (8 * j));
    }
  }
  if (a.packed) store_packed_mask<PX>(a.fg_bits, p0, bits, active);
}

// ---- clip launches (bgs_process_clip_device): T consecutive frames of every stream in one launch, the model of a pixel loaded
// once, updated T times in registers with the statements of the single-frame kernel, written back once - see kernel_mog2.h.
// One pixel per lane; the data-dependent loads / stores are the single-frame kernel's: a plane is stored if any frame of the clip
// changed its bits, a mean / variance slot that was not loaded is written by the pixel that owns a mode there at the end.
constexpr int kMog1ClipMax = 8;
struct Mog1ClipArgs {
  Mog1Args m;                            // frame / fg / fg_bits point at the first frame of the launch
  size_t frame_stride, fg_stride, bits_stride;  // bytes (bits_stride: 64-bit words) from one frame to the next
  float alpha[kMog1ClipMax];             // learning rate of each frame (1/n while the history fills)
};

template <int C, int T>
__global__ __launch_bounds__(kBlock) void mog1_clip_kernel(const Mog1ClipArgs c) {
  constexpr int K = kMog1K, R = 2 + 2 * C, NP = K * R;
  const Mog1Args& a = c.m;
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = blk * kBlock + threadIdx.x;
  if (p0 >= a.npix) return;  // wave-uniform whenever masks are bit-packed (npix % 64 == 0)
  const size_t sp = a.state_off + p0;
  uint32_t pixw[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint8_t* f = a.frame + (size_t)t * c.frame_stride + p0 * C;
    if constexpr (C == 3)
      pixw[t] = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16);
    else
      pixw[t] = f[0];
  }
  float st[NP];
  unsigned need = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) st[k * R] = a.state[mog1_plane_off<C>(k * R, sp)], st[k * R + 1] = a.state[mog1_plane_off<C>(k * R + 1, sp)];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool live = st[k * R + 1] >= FLT_EPSILON;
    need |= (unsigned)live << k;
#pragma unroll
    for (int f = 2; f < R; ++f) st[k * R + f] = live ? a.state[mog1_plane_off<C>(k * R + f, sp)] : 0.f;
  }
  Mog1Px<C> s;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s.sk[k] = st[k * R], s.w[k] = st[k * R + 1];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) s.mu[k][cc] = st[k * R + 2 + cc], s.var[k][cc] = st[k * R + 2 + C + cc];
  }
  unsigned everLive = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float pix[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) pix[cc] = (float)((pixw[t] >> (8 * cc)) & 0xffu);
    const int m = thr_bin(mog1_pixel<C>(s, pix, a, c.alpha[t]), a.thr, a.enable_thr);
    if (a.fg) a.fg[(size_t)t * c.fg_stride + p0] = (uint8_t)m;
    if (a.packed) store_packed_mask<1>(a.fg_bits + (size_t)t * c.bits_stride, p0, (uint32_t)(m != 0), true);
#pragma unroll
    for (int k = 0; k < K; ++k) everLive |= (unsigned)(s.w[k] >= FLT_EPSILON) << k;
  }
  auto put = [&](int q, float v) {  // store a plane only if the clip changed its bits
    if (__float_as_uint(st[q]) != __float_as_uint(v)) a.state[mog1_plane_off<C>(q, sp)] = v;
  };
#pragma unroll
  for (int k = 0; k < K; ++k) {
    put(k * R, s.sk[k]), put(k * R + 1, s.w[k]);
    if ((need >> k) & 1u) {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) put(k * R + 2 + cc, s.mu[k][cc]), put(k * R + 2 + C + cc, s.var[k][cc]);
    } else if ((everLive >> k) & 1u) {  // not loaded, in use after some frame of the clip (the reference wrote it then): every field
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        a.state[mog1_plane_off<C>(k * R + 2 + cc, sp)] = s.mu[k][cc];
        a.state[mog1_plane_off<C>(k * R + 2 + C + cc, sp)] = s.var[k][cc];
      }
    }
  }
}

template <int C>
__global__ __launch_bounds__(kBlock) void mog1_clear_kernel(const Mog1Args a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npix) return;
  const size_t sp = a.state_off + p;
#pragma unroll
  for (int q = 0; q < mog1_planes<C>(); ++q) a.state[mog1_plane_off<C>(q, sp)] = 0.f;
}

}  // namespace bgs
