// kernel_mog1.h — K5: MixtureOfGaussianV1BGS (cv::BackgroundSubtractorMOG, KaewTraKulPong-Bowden) update + classify.
//
// Replaces  MixtureOfGaussianV1BGS::process  package_bgs/MixtureOfGaussianV1BGS.cpp:51-56
// Algorithm: OpenCV 2.4 bgfg_gaussmix.cpp process8uC3 / process8uC1: scan the modes in sortKey
// order until an empty one (w < FLT_EPSILON) or a match (d2 < vT * sum(var)); update the match and bubble it up by
// sortKey = w_old / sqrt(sum var); otherwise replace the weakest; renormalise; foreground iff the hit lies past the
// background prefix (cumulative weight > T).
//
// Layout (round 3, the form MOG2 took in kernel_mog2.h): tile = 256 pixels; per tile
//   sortKey[K][256], weight[K][256]   planes in RANK order (the reference's sorted order; they change every frame anyway)
//   record[K slots][256][2C floats]   {mean[C], var[C]} of the mode that lives in that SLOT - a mode never moves once created
//   meta[256] uint16                  rank -> slot, 3 bits per rank, slot + 1 (0: this rank never held a mode)
// = 162 B/pixel for BGR, 82 for gray.  Rounds 1-2 kept means and variances as 2C more planes per RANK: the bubble of a matched mode
// dragged whole records through the sort, and - what cost more - a frame issued up to 30 predicated 4-byte loads and 40
// predicated stores per pixel: the kernel was bound by instruction issue at 47 % (S_surv) / 72 % (S_sat) of the achievable
// HBM rate for the bytes it moved (profiles/r03_mog1_pmc.txt).  Now a mode's record is two 12-byte loads, the one record a frame
// changes (the matched or created mode) two 12-byte stores.  Assigned slots are a prefix of the ranks (a mode is created at the
// first rank whose weight is below FLT_EPSILON: a dead mode's slot is reused, else the next free slot is taken).
// One pixel per lane.  Traffic (BGR, data-dependent): r 3 + 2 + 40 + 24 x (ranks loaded), w 40 (changed planes only) + 24 + 2 + 1.
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
  float sk[kMog1K], w[kMog1K], mu[kMog1K][C], var[kMog1K][C];  // by rank
  int sl[kMog1K];                                               // slot + 1 of the mode at each rank, 0 = none yet
};

template <int C>
__device__ __forceinline__ void mog1_swap(Mog1Px<C>& s, int i, int j) {
  float t;
  t = s.sk[i], s.sk[i] = s.sk[j], s.sk[j] = t;
  t = s.w[i], s.w[i] = s.w[j], s.w[j] = t;
  const int u = s.sl[i];
  s.sl[i] = s.sl[j], s.sl[j] = u;
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

// one pixel, same statement order as process8uC3 so every float rounds identically; returns 0 / 255.
// hit = the rank (after the re-sort) of the one mode whose mean / variance this frame wrote (matched or created), -1 if none.
template <int C>
__device__ __forceinline__ int mog1_pixel(Mog1Px<C>& s, const float (&pix)[C], const Mog1Args& a, const float alpha, int& hit) {
  constexpr int K = kMog1K;
  int kHit = -1, kForeground = -1;
  hit = -1;
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
      int na = 0;  // slots in use: the ranks that ever held a mode are a prefix
#pragma unroll
      for (int k = 0; k < K; ++k) na += s.sl[k] != 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k == kk) {
          wsum += a.w0 - s.w[k];
          s.w[k] = a.w0;
#pragma unroll
          for (int c = 0; c < C; ++c) s.mu[k][c] = pix[c], s.var[k][c] = a.var0;
          s.sk[k] = a.sk0;
          if (s.sl[k] == 0) s.sl[k] = na + 1;  // (then k == na: every rank below holds a mode)
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k >= k_end) wsum += s.w[k];
    }
    hit = kHit;
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

// ---- tile layout (floats)
template <int C>
__host__ __device__ constexpr int mog1_tile_floats() { return kMog1K * kMog1Tile * 2 + kMog1K * kMog1Tile * 2 * C + kMog1Tile / 2; }
template <int C>
struct Mog1Ptr {
  float* skw;      // + rank * 256: sortKey; + (K + rank) * 256: weight   (already at the pixel)
  float* rec;      // + slot * 256 * 2C: the pixel's record in that slot
  uint16_t* meta;  // the pixel's rank -> slot word
};
template <int C>
__device__ __forceinline__ Mog1Ptr<C> mog1_ptr(float* state, size_t sp) {
  float* tile = state + (sp >> 8) * (size_t)mog1_tile_floats<C>();
  const int l = (int)(sp & 255);
  Mog1Ptr<C> p;
  p.skw = tile + l;
  p.rec = tile + 2 * kMog1K * kMog1Tile + l * 2 * C;
  p.meta = reinterpret_cast<uint16_t*>(tile + 2 * kMog1K * kMog1Tile + kMog1K * kMog1Tile * 2 * C) + l;
  return p;
}

// a record = {mean[C], var[C]}: two 12-byte accesses for BGR, one 8-byte access for gray
template <int C>
__device__ __forceinline__ void mog1_rec_load(const float* p, float (&mu)[C], float (&var)[C]) {
  if constexpr (C == 3) {
    typedef float f3 __attribute__((ext_vector_type(3)));
    typedef f3 __attribute__((aligned(4))) f3u;
    const f3 m = *reinterpret_cast<const f3u*>(p), v = *reinterpret_cast<const f3u*>(p + 3);
    mu[0] = m.x, mu[1] = m.y, mu[2] = m.z, var[0] = v.x, var[1] = v.y, var[2] = v.z;
  } else {
    const float2 r = *reinterpret_cast<const float2*>(p);
    mu[0] = r.x, var[0] = r.y;
  }
}
template <int C>
__device__ __forceinline__ void mog1_rec_store(float* p, const float (&mu)[C], const float (&var)[C]) {
  if constexpr (C == 3) {
    typedef float f3 __attribute__((ext_vector_type(3)));
    typedef f3 __attribute__((aligned(4))) f3u;
    f3 m = {mu[0], mu[1], mu[2]}, v = {var[0], var[1], var[2]};
    *reinterpret_cast<f3u*>(p) = m, *reinterpret_cast<f3u*>(p + 3) = v;
  } else {
    *reinterpret_cast<float2*>(p) = make_float2(mu[0], var[0]);
  }
}

// Load a pixel's model in rank order: the records of ranks 0 .. M-1 (the caller knows that no lane of the wave holds more than M
// modes; the ranks that ever held a mode are a prefix).  Every load is unconditional - a rank without a mode reads slot 0, whose
// values are never used (its weight is 0: the reference's scan stops there) - so that none of them is waited for before the last
// is issued.
template <int C, int M>
__device__ __forceinline__ void mog1_load(const Mog1Ptr<C>& q, unsigned meta, Mog1Px<C>& s) {
#pragma unroll
  for (int k = 0; k < kMog1K; ++k) {
    s.sl[k] = (int)((meta >> (3 * k)) & 7u);
    if (k < M) {
      mog1_rec_load<C>(q.rec + (size_t)(s.sl[k] ? s.sl[k] - 1 : 0) * (kMog1Tile * 2 * C), s.mu[k], s.var[k]);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) s.mu[k][c] = 0.f, s.var[k][c] = 0.f;
    }
  }
}

__device__ __forceinline__ unsigned mog1_meta_pack(const int (&sl)[kMog1K]) {
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < kMog1K; ++k) m |= (unsigned)sl[k] << (3 * k);
  return m;
}

// grid: ceil(npix / kBlock), one pixel per lane
template <int C>
__global__ __launch_bounds__(kBlock) void mog1_update_kernel(const Mog1Args a) {
  constexpr int K = kMog1K;
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {  // each XCD streams one contiguous eighth of the launch (see kernel_mog2.h)
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = blk * kBlock + threadIdx.x;
  const bool active = p0 < a.npix;
  const size_t pc = active ? p0 : a.npix - 1;  // lanes past the end work on the last pixel and store nothing
  const Mog1Ptr<C> q = mog1_ptr<C>(a.state, a.state_off + pc);
  const unsigned meta = *q.meta;
  float sk0[K], w0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) sk0[k] = q.skw[k * kMog1Tile], w0[k] = q.skw[(K + k) * kMog1Tile];
  uint8_t px[C];
#pragma unroll
  for (int c = 0; c < C; ++c) px[c] = a.frame[pc * C + c];
  Mog1Px<C> s;
  // one wave-uniform choice: the most modes any lane of the wave holds (at least two are loaded).  Round 3 chose between two and
  // all five; with fresh sensor noise in every frame (round 4's benchmark frames) a quiet scene keeps spawning short-lived third
  // modes, and most waves have a lane with three - which then cost every lane of the wave five records.
  if (!__any((meta >> 6) != 0))
    mog1_load<C, 2>(q, meta, s);
  else if (!__any((meta >> 9) != 0))
    mog1_load<C, 3>(q, meta, s);
  else if (!__any((meta >> 12) != 0))
    mog1_load<C, 4>(q, meta, s);
  else
    mog1_load<C, 5>(q, meta, s);
#pragma unroll
  for (int k = 0; k < K; ++k) s.sk[k] = sk0[k], s.w[k] = w0[k];
  float pix[C];
#pragma unroll
  for (int c = 0; c < C; ++c) pix[c] = (float)px[c];
  int hit;
  const int m = thr_bin(mog1_pixel<C>(s, pix, a, a.alpha, hit), a.thr, a.enable_thr);
  if (active) {
    // write-back is data-dependent (exact): a sortKey / weight plane only if its bits changed (a quiet pixel has one or two modes,
    // the planes of the others stay 0), the one record the frame wrote, the rank -> slot word if the order changed
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (__float_as_uint(s.sk[k]) != __float_as_uint(sk0[k])) q.skw[k * kMog1Tile] = s.sk[k];
      if (__float_as_uint(s.w[k]) != __float_as_uint(w0[k])) q.skw[(K + k) * kMog1Tile] = s.w[k];
    }
    if (hit >= 0) {
      float mu[C], var[C];
      int slot = 0;
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k == hit) {
          slot = s.sl[k] - 1;
#pragma unroll
          for (int c = 0; c < C; ++c) mu[c] = s.mu[k][c], var[c] = s.var[k][c];
        }
      mog1_rec_store<C>(q.rec + (size_t)slot * (kMog1Tile * 2 * C), mu, var);
    }
    const unsigned meta_new = mog1_meta_pack(s.sl);
    if (meta_new != meta) *q.meta = (uint16_t)meta_new;
    if (a.fg) a.fg[p0] = (uint8_t)m;
  }
  if (a.packed) store_packed_mask<1>(a.fg_bits, p0, (uint32_t)(m != 0), active);
}

// ---- clip launches (bgs_process_clip_device): T consecutive frames of every stream in one launch, the model of a pixel loaded
// once, updated T times in registers with the statements of the single-frame kernel, written back once - see kernel_mog2.h.
// Every record that some frame of the clip wrote goes back to its slot.
constexpr int kMog1ClipMax = 8;
struct Mog1ClipArgs {
  Mog1Args m;                            // frame / fg / fg_bits point at the first frame of the launch
  size_t frame_stride, fg_stride, bits_stride;  // bytes (bits_stride: 64-bit words) from one frame to the next
  float alpha[kMog1ClipMax];             // learning rate of each frame (1/n while the history fills)
};

template <int C, int T>
__global__ __launch_bounds__(kBlock) void mog1_clip_kernel(const Mog1ClipArgs c) {
  constexpr int K = kMog1K;
  const Mog1Args& a = c.m;
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = blk * kBlock + threadIdx.x;
  if (p0 >= a.npix) return;  // wave-uniform whenever masks are bit-packed (npix % 64 == 0)
  const Mog1Ptr<C> q = mog1_ptr<C>(a.state, a.state_off + p0);
  uint32_t pixw[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint8_t* f = a.frame + (size_t)t * c.frame_stride + p0 * C;
    if constexpr (C == 3)
      pixw[t] = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16);
    else
      pixw[t] = f[0];
  }
  const unsigned meta = *q.meta;
  float sk0[K], w0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) sk0[k] = q.skw[k * kMog1Tile], w0[k] = q.skw[(K + k) * kMog1Tile];
  Mog1Px<C> s;
  mog1_load<C, kMog1K>(q, meta, s);
#pragma unroll
  for (int k = 0; k < K; ++k) s.sk[k] = sk0[k], s.w[k] = w0[k];
  unsigned wrote = 0;  // bit (slot + 1): some frame of the clip wrote that slot's record
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float pix[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) pix[cc] = (float)((pixw[t] >> (8 * cc)) & 0xffu);
    int hit;
    const int m = thr_bin(mog1_pixel<C>(s, pix, a, c.alpha[t], hit), a.thr, a.enable_thr);
    if (a.fg) a.fg[(size_t)t * c.fg_stride + p0] = (uint8_t)m;
    if (a.packed) store_packed_mask<1>(a.fg_bits + (size_t)t * c.bits_stride, p0, (uint32_t)(m != 0), true);
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k == hit) wrote |= 1u << s.sl[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (__float_as_uint(s.sk[k]) != __float_as_uint(sk0[k])) q.skw[k * kMog1Tile] = s.sk[k];
    if (__float_as_uint(s.w[k]) != __float_as_uint(w0[k])) q.skw[(K + k) * kMog1Tile] = s.w[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (s.sl[k] && ((wrote >> s.sl[k]) & 1u)) mog1_rec_store<C>(q.rec + (size_t)(s.sl[k] - 1) * (kMog1Tile * 2 * C), s.mu[k], s.var[k]);
  const unsigned meta_new = mog1_meta_pack(s.sl);
  if (meta_new != meta) *q.meta = (uint16_t)meta_new;
}

// needToInitialize: bgmodel = zeros (a zero rank -> slot word: no mode anywhere)
template <int C>
__global__ __launch_bounds__(kBlock) void mog1_clear_kernel(const Mog1Args a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npix) return;
  const Mog1Ptr<C> q = mog1_ptr<C>(a.state, a.state_off + p);
#pragma unroll
  for (int k = 0; k < 2 * kMog1K; ++k) q.skw[k * kMog1Tile] = 0.f;
#pragma unroll
  for (int k = 0; k < kMog1K; ++k)
#pragma unroll
    for (int f = 0; f < 2 * C; ++f) q.rec[(size_t)k * (kMog1Tile * 2 * C) + f] = 0.f;
  *q.meta = 0;
}

}  // namespace bgs
