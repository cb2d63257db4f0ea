// kernel_fanout.h — one frame, several classes: what FrameProcessor::process does with its pre-processed frame
// (FrameProcessor.cpp:169-340: the same img_prep goes to every enabled IBGS, one after the other), for the byte-stream classes
//   FrameDifferenceBGS, StaticFrameDifferenceBGS, WeightedMovingMeanBGS, WeightedMovingVarianceBGS, AdaptiveBackgroundLearning, SigmaDeltaBGS
// as ONE kernel over ONE read of the frame and ONE shared history ring (frames t-1, t-2), writing every class's own mask /
// background / state.  Each class's arithmetic is the *_body function of kernel_pointwise.h that its stand-alone kernel runs, so the
// outputs are those of the separate engines bit for bit; what is saved is traffic: BASELINE configs[2] (WeightedMovingVariance +
// AdaptiveBackgroundLearning on the same 3840x2160 frames) moves 3 x 3 (frames) + 3 + 3 (ABL state in / out) + 1 + 1 (masks) = 17
// B/pixel here instead of 10 + 10, all five history / state classes together 26 instead of 47.
// Mapping: like the stand-alone kernels, one lane owns G consecutive pixels; with AdaptiveBackgroundLearning's table in play the
// launch takes abl_kernel's form (1024-lane persistent workgroups, the 64 KB table in LDS once per workgroup).
#pragma once
#include "kernel_pointwise.h"

namespace bgs {

enum { kFanFD = 1, kFanSFD = 2, kFanWMM = 4, kFanWMV = 8, kFanABL = 16, kFanSD = 32 };

struct FanOut {
  uint8_t* fg;     // [npix] or null
  uint8_t* bg;     // [npix][C] or null (classes that deliver a background image)
  uint64_t* bits;  // [npix/64] or null
  int thr, enable_thr, enable_weight;
};

struct FanArgs {
  const uint8_t *cur, *p1, *p2;  // this frame, the one before, the one before that (shared history)
  size_t npix;
  unsigned mask;                 // classes that produce output THIS frame (a class still warming up is simply absent)
  FanOut fd, sfd, wmm, wmv, abl, sd;
  const uint8_t* sfd_bg;         // StaticFrameDifference: the frozen first frame
  uint8_t* abl_state;            // AdaptiveBackgroundLearning: uint8 background, updated in place
  int abl_update;
  uint8_t *sd_mt, *sd_vt;        // SigmaDelta: Mt, Vt, updated in place
  uint32_t sd_N;
  int sd_vmin, sd_vmax;
};

template <int G, int C>
__device__ __forceinline__ void fan_tile(const FanArgs& a, const uint8_t* T, size_t p0, bool active) {
  PxGroup<G, C> x, y, z, d;
#pragma unroll
  for (int i = 0; i < PxGroup<G, C>::NB / 4; ++i) x.b.w[i] = y.b.w[i] = z.b.w[i] = d.b.w[i] = 0;
  if (active) {
    x.load(a.cur + p0 * C);
    if (a.mask & (kFanFD | kFanWMM | kFanWMV)) y.load(a.p1 + p0 * C);
    if (a.mask & (kFanWMM | kFanWMV)) z.load(a.p2 + p0 * C);
  }
  // every branch below is wave-uniform (launch arguments); a class's block is its stand-alone kernel's body
  if (a.mask & kFanFD) {
    if (active) absdiff_body<G, C>(x, y, d);
    gray_thr_store_to<G, C>(d, a.fd.thr, a.fd.enable_thr, a.fd.fg, a.fd.bits, p0, active);
  }
  if (a.mask & kFanSFD) {
    if (active) {
      PxGroup<G, C> b;
      b.load(a.sfd_bg + p0 * C);
      absdiff_body<G, C>(x, b, d);
    }
    gray_thr_store_to<G, C>(d, a.sfd.thr, a.sfd.enable_thr, a.sfd.fg, a.sfd.bits, p0, active);
  }
  if (a.mask & kFanWMM) {
    if (active) {
      PxGroup<G, C> bgq;
      wmm_body<G, C>(x, y, z, a.wmm.enable_weight, d, bgq);
      if (a.wmm.bg) bgq.store(a.wmm.bg + p0 * C);
    }
    gray_thr_store_to<G, C>(d, a.wmm.thr, a.wmm.enable_thr, a.wmm.fg, a.wmm.bits, p0, active);
  }
  if (a.mask & kFanWMV) {
    if (active) wmv_body<G, C>(x, y, z, a.wmv.enable_weight, a.wmv.enable_thr, a.wmv.thr, d);
    gray_thr_store_to<G, C>(d, a.wmv.thr, a.wmv.enable_thr, a.wmv.fg, a.wmv.bits, p0, active);
  }
  if (a.mask & kFanABL) {
    if (active) {
      PxGroup<G, C> bgq;
      bgq.load(a.abl_state + p0 * C);
      if (a.abl_update) {
        abl_body<G, C, true>(x, bgq, T, d);
        bgq.store(a.abl_state + p0 * C);
      } else {
        abl_body<G, C, false>(x, bgq, T, d);
      }
      if (a.abl.bg) bgq.store(a.abl.bg + p0 * C);
    }
    gray_thr_store_to<G, C>(d, a.abl.thr, a.abl.enable_thr, a.abl.fg, a.abl.bits, p0, active);
  }
  if constexpr (C == 3) {
    if (a.mask & kFanSD) {
      PxGroup<G, 1> m;
#pragma unroll
      for (int i = 0; i < PxGroup<G, 1>::NB / 4; ++i) m.b.w[i] = 0;
      uint32_t bits = 0;
      if (active) {
        PxGroup<G, 3> mt, vt;
        mt.load(a.sd_mt + p0 * 3);
        vt.load(a.sd_vt + p0 * 3);
        sd_body<G>(x, mt, vt, a.sd_N, a.sd_vmin, a.sd_vmax, m, bits);
        mt.store(a.sd_mt + p0 * 3);
        vt.store(a.sd_vt + p0 * 3);
        if (a.sd.fg) m.store(a.sd.fg + p0);
      }
      if (a.sd.bits) {
        if constexpr (64 % G == 0) store_packed_mask<G>(a.sd.bits, p0, bits, active);
      }
    }
  }
}

// LUT = the launch updates AdaptiveBackgroundLearning's background: 1024-lane persistent workgroups with the table in LDS
template <int G, int C, bool LUT>
__global__ __launch_bounds__(LUT ? kAblBlock : kBlock) void fan_kernel(const FanArgs a, const uint8_t* __restrict__ lut) {
  if constexpr (LUT) {
    __shared__ uint8_t T[256 * kAblLutStride];
    abl_load_lut(T, lut, kAblBlock);
    __syncthreads();
    const size_t per_tile = (size_t)kAblBlock * G, ntiles = (a.npix + per_tile - 1) / per_tile;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const size_t p0 = (tile * kAblBlock + threadIdx.x) * G;
      fan_tile<G, C>(a, T, p0, p0 < a.npix);
    }
  } else {
    const size_t p0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * G;
    fan_tile<G, C>(a, nullptr, p0, p0 < a.npix);
  }
}

}  // namespace bgs
