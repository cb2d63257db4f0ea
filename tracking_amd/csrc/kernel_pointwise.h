// kernel_pointwise.h — K1/K2/K3: the byte-in / byte-out IBGS classes, each ONE fused kernel
// (the reference runs 3-15 full-frame OpenCV passes with float temporaries for the same result).
//
//   K1 framediff_kernel  FrameDifferenceBGS::process        package_bgs/FrameDifferenceBGS.cpp:45-51
//                        StaticFrameDifferenceBGS::process  package_bgs/StaticFrameDifferenceBGS.cpp:42-48
//   K2 wmm_kernel        WeightedMovingMeanBGS::process     package_bgs/WeightedMovingMeanBGS.cpp:52-84
//      wmv_kernel        WeightedMovingVarianceBGS::process package_bgs/WeightedMovingVarianceBGS.cpp:53-106, 126-137
//   K3 abl_kernel        AdaptiveBackgroundLearning::process package_bgs/AdaptiveBackgroundLearning.cpp:43-71
//
// Mapping: one lane owns G consecutive pixels (G*C bytes, whole dwords: G = 16 -> C x 16-byte loads, G = 4 -> 12/4 bytes,
// G = 1 -> byte loads for ragged sizes), mask store G bytes.  HBM-bound byte streams, no reuse, no LDS.
// The OpenCV primitive semantics (P1..P8) are listed in DESIGN.md §5.
#pragma once
#include "bgs_device.h"

namespace bgs {

struct FrameArgs {
  const uint8_t* cur;   // [npix][C]
  const uint8_t* p1;    // previous frame / static background / ABL state (read)
  const uint8_t* p2;    // frame before that
  uint8_t* state_out;   // ABL: updated background state (may alias p1)
  uint8_t* fg;          // [npix] or null
  uint8_t* bg;          // [npix][C] or null
  uint64_t* fg_bits;    // [npix/64] or null
  size_t npix;
  int thr, enable_thr, enable_weight, update, xcd_swizzle;
  double alpha, beta;   // ABL: alpha, 1-alpha
};

template <int G, int C>
struct PxGroup {
  static constexpr int NB = (G * C + 3) / 4 * 4;
  Bytes<NB> b;
  __device__ __forceinline__ void load(const uint8_t* p) {
    if constexpr ((G * C) % 4 == 0) {
      b = load_bytes<NB>(p);
    } else {
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) b.w[i] = 0;
#pragma unroll
      for (int i = 0; i < G * C; ++i) b.set(i, p[i]);
    }
  }
  __device__ __forceinline__ void store(uint8_t* p) const {
    if constexpr ((G * C) % 4 == 0) {
      store_bytes<NB>(p, b);
    } else {
#pragma unroll
      for (int i = 0; i < G * C; ++i) p[i] = (uint8_t)b.get(i);
    }
  }
};

// gray (3ch) + threshold of a per-channel u8 difference image, G pixels: the mask bytes to `fg`, the bits to `fg_bits`
template <int G, int C>
__device__ __forceinline__ void gray_thr_store_to(const PxGroup<G, C>& d, int thr, int enable_thr, uint8_t* fg, uint64_t* fg_bits, size_t p0, bool active) {
  PxGroup<G, 1> m;
#pragma unroll
  for (int i = 0; i < PxGroup<G, 1>::NB / 4; ++i) m.b.w[i] = 0;
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    int g;
    if constexpr (C == 3 && G % 4 == 0) {  // four BGR pixels = three dwords: each pixel's bytes brought to bits 0..23 (gray_bgr_dword)
      const uint32_t w0 = d.b.w[j / 4 * 3], w1 = d.b.w[j / 4 * 3 + 1], w2 = d.b.w[j / 4 * 3 + 2];
      const uint32_t px = j % 4 == 0 ? w0 : j % 4 == 1 ? __builtin_amdgcn_alignbyte(w1, w0, 3) : j % 4 == 2 ? __builtin_amdgcn_alignbyte(w2, w1, 2) : w2 >> 8;
      g = (int)gray_bgr_dword(px);
    } else {
      g = (C == 3) ? gray_bgr(d.b.get(3 * j), d.b.get(3 * j + 1), d.b.get(3 * j + 2)) : d.b.get(j);
    }
    const int v = thr_bin(g, thr, enable_thr);
    m.b.set(j, v);
    bits |= (uint32_t)(v != 0) << j;
  }
  if (active && fg) m.store(fg + p0);
  if (fg_bits) {
    if constexpr (64 % G == 0) store_packed_mask<G>(fg_bits, p0, active ? bits : 0u, active);
  }
}
template <int G, int C>
__device__ __forceinline__ void gray_thr_store(const PxGroup<G, C>& d, const FrameArgs& a, size_t p0, bool active, bool packed) {
  gray_thr_store_to<G, C>(d, a.thr, a.enable_thr, a.fg, packed ? a.fg_bits : nullptr, p0, active);
}

// cv::absdiff 8U (FrameDifferenceBGS.cpp:45, StaticFrameDifferenceBGS.cpp:42)
template <int G, int C>
__device__ __forceinline__ void absdiff_body(const PxGroup<G, C>& x, const PxGroup<G, C>& y, PxGroup<G, C>& d) {
#pragma unroll
  for (int i = 0; i < G * C; ++i) d.b.set(i, abs(x.b.get(i) - y.b.get(i)));
}

template <int G, int C>
__global__ __launch_bounds__(kBlock) void framediff_kernel(const FrameArgs a) {
  const size_t p0 = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = p0 < a.npix;
  PxGroup<G, C> d;
  if (active) {
    PxGroup<G, C> x, y;
    x.load(a.cur + p0 * C);
    y.load(a.p1 + p0 * C);
    absdiff_body<G, C>(x, y, d);
  }
  gray_thr_store<G, C>(d, a, p0, active, a.fg_bits != nullptr);
}

// cv::cvtColor(BGR2GRAY) / copy of a 1-channel frame into a.fg (AdaptiveSelectiveBackgroundLearning.cpp:37-40, :47-48)
template <int G, int C>
__global__ __launch_bounds__(kBlock) void gray_kernel(const FrameArgs a) {
  const size_t p0 = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = p0 < a.npix;
  PxGroup<G, C> d;
  if (active) d.load(a.cur + p0 * C);
  gray_thr_store<G, C>(d, a, p0, active, false);
}

// the f32 image the reference builds with convertTo(CV_32F, 1./255.)
__device__ __forceinline__ float to_unit(int v) { return (float)v * (float)(1. / 255.); }

// weighted mean of three frames as the reference's MatExpr evaluates it:
//   weighted  : t = addWeighted(I,w0,P1,w1) ; mean = P2*(float)w2 + t   (scaleAdd, float)
template <bool WMM_UNWEIGHTED>
__device__ __forceinline__ float mean3(float i0, float i1, float i2, double w0, double w1, double w2) {
  if constexpr (WMM_UNWEIGHTED) {
    const float t = i0 + i1;  // WeightedMovingMeanBGS.cpp:66: (A + B + C) / 3.0
    return add_weighted(t, 1. / 3.0, i2, 1. / 3.0);
  } else {
    const float t = add_weighted(i0, w0, i1, w1);
    return i2 * (float)w2 + t;
  }
}

// WeightedMovingMeanBGS.cpp:52-84: the background bytes `bgq` and the per-channel |I - bg| bytes `d` of G pixels
template <int G, int C>
__device__ __forceinline__ void wmm_body(const PxGroup<G, C>& x, const PxGroup<G, C>& y, const PxGroup<G, C>& z, int enable_weight, PxGroup<G, C>& d, PxGroup<G, C>& bgq) {
#pragma unroll
  for (int i = 0; i < PxGroup<G, C>::NB / 4; ++i) bgq.b.w[i] = 0;
#pragma unroll
  for (int i = 0; i < G * C; ++i) {
    const int xi = x.b.get(i);
    const float i0 = to_unit(xi), i1 = to_unit(y.b.get(i)), i2 = to_unit(z.b.get(i));
    const float bgf = enable_weight ? mean3<false>(i0, i1, i2, 0.5, 0.3, 0.2) : mean3<true>(i0, i1, i2, 0, 0, 0);
    const int b8 = sat_u8(bgf * 255.f);
    bgq.b.set(i, b8);
    d.b.set(i, abs(xi - b8));
  }
}

// (Round 3 tried wmv_kernel's integer shortcut here: byte = round(M / 10), M = 5 b0 + 3 b1 + 2 b2, exact except when M ends in 5 - an
// exact tie that the float pipeline's own rounding errors decide.  That is one byte in ten on a live scene, not one in fifty: the
// list traffic in LDS cost more than the float pipeline it replaced, 0.210 against 0.171 ms on 8 x 4K.  All 2^24 triples matched.)
template <int G, int C>
__global__ __launch_bounds__(kBlock) void wmm_kernel(const FrameArgs a) {
  const size_t p0 = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = p0 < a.npix;
  PxGroup<G, C> d;
  if (active) {
    PxGroup<G, C> x, y, z, bgq;
    x.load(a.cur + p0 * C);
    y.load(a.p1 + p0 * C);
    z.load(a.p2 + p0 * C);
    wmm_body<G, C>(x, y, z, a.enable_weight, d, bgq);
    if (a.bg) bgq.store(a.bg + p0 * C);
  }
  gray_thr_store<G, C>(d, a, p0, active, a.fg_bits != nullptr);
}

// WeightedMovingVarianceBGS.cpp:126-137: weight * (|x - mean|)^2, every step rounded to float
__device__ __forceinline__ float wvar(float x, float mean, double w) {
  const float dd = fabsf(x - mean);
  const float p = dd * dd;
  return p * (float)w;
}

// Exact shortcut for quiet pixels.  With the default weights (0.5, 0.3, 0.2; they sum to 1) the byte the reference produces for a
// channel is round(255 * weighted std-dev of the three frame values), and a weighted std-dev of values spanning a range r is at
// most r/2 (two-point distribution with the closest achievable split 0.5 | 0.5); the float pipeline's error is ~1e-5.  So if
// every channel of a pixel has range(b0,b1,b2) < 2*thr, every channel byte is <= thr, hence gray <= thr and the thresholded
// mask is 0 - no float work needed.  Only pixels with real temporal change take the full path (which stays bit-exact).
template <int G, int C>
__device__ __forceinline__ void wmv_body(const PxGroup<G, C>& x, const PxGroup<G, C>& y, const PxGroup<G, C>& z, int enable_weight, int enable_thr, int thr, PxGroup<G, C>& d) {
  const double w0 = enable_weight ? 0.5 : 0.3, w1 = 0.3, w2 = enable_weight ? 0.2 : 0.3;  // :68-70 (unweighted = 0.3 x3, sic)
  const bool shortcut = enable_weight && enable_thr && thr >= 1;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    bool quiet = shortcut;
    if (shortcut) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int b0 = x.b.get(j * C + c), b1 = y.b.get(j * C + c), b2 = z.b.get(j * C + c);
        quiet = quiet && (max(b0, max(b1, b2)) - min(b0, min(b1, b2)) < 2 * thr);
      }
    }
    if (quiet) {
#pragma unroll
      for (int c = 0; c < C; ++c) d.b.set(j * C + c, 0);  // any value <= thr gives the same mask
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int i = j * C + c;
        const float i0 = to_unit(x.b.get(i)), i1 = to_unit(y.b.get(i)), i2 = to_unit(z.b.get(i));
        const float m = mean3<false>(i0, i1, i2, w0, w1, w2);
        const float v = (wvar(i0, m, w0) + wvar(i1, m, w1)) + wvar(i2, m, w2);  // :83
        const float sd = sqrt_rn(v);                                            // :95
        d.b.set(i, sat_u8(sd * 255.f));                                         // :99
      }
    }
  }
}

// ---- round 3: the moving pixels' bytes without the float pipeline, where that is provably the same result.
// With the default weights the byte is round(255 * sd), 255 * sd = sqrt(N) / 10 with the INTEGER
//   N = 10 * (5 b0^2 + 3 b1^2 + 2 b2^2) - (5 b0 + 3 b1 + 2 b2)^2      (100 * 255^2 * weighted variance, 0 .. 6 502 500),
// and the reference's float pipeline (to_unit, addWeighted in double, three wvar terms, sqrt, * 255: ~55 instructions per byte) differs
// from the real number by well under 1e-4 in that byte (in units of N: < 6 at the top of the range, < 1 near 0).  So
// k = rint(sqrt(N) / 10) IS the reference's byte unless N lies within W = 6 + N / 65536 of a rounding boundary (10 k +- 5)^2, and
// is never more than 1 away from it.
//   * threshold on (the default): a pixel's mask is gray(bytes) > thr, gray is a convex combination, so gray(k's) further than 2
//     from thr settles the mask; only the pixels inside that band need their exact bytes;
//   * threshold off (the mask is the gray value itself): the bytes near a rounding boundary need the float pipeline - 1-2 % of a
//     moving scene.
// The few bytes that need it are appended to a per-workgroup list in LDS and computed with the float pipeline by as many lanes as
// there are entries, one byte each (a lane that computed its own rare bytes in place would hold up the other 63), and patched into
// the owners' registers through a staging copy in LDS.  A workgroup whose list overflows (adversarial input) takes the float
// pipeline for everything, as before.  Tests: every one of the 2^24 byte triples against the CPU restatement's float pipeline with
// the threshold off (tests/test_gpu_parity.py: test_wmv_every_byte_triple...), full-range random frames at seven thresholds with
// it on (test_wmv_mask_band_around_threshold), besides the clips.
constexpr int kWmvListCap = 512, kWmvStageStride = 13;  // entries per workgroup; dwords per lane in the staging copy (odd: no bank conflicts)

__device__ __forceinline__ int wmv_exact_byte(int b0, int b1, int b2, double w0, double w1, double w2) {
  const float i0 = to_unit(b0), i1 = to_unit(b1), i2 = to_unit(b2);
  const float m = mean3<false>(i0, i1, i2, w0, w1, w2);
  const float v = (wvar(i0, m, w0) + wvar(i1, m, w1)) + wvar(i2, m, w2);  // :83
  return sat_u8(sqrt_rn(v) * 255.f);                                      // :95, :99
}

// default weights only.  Returns k; `near` (if asked for) = too close to a rounding boundary to be taken without the float pipeline.
template <bool NEAR>
__device__ __forceinline__ int wmv_fast_byte(int b0, int b1, int b2, bool& near) {
  const int d01 = b0 - b1, d02 = b0 - b2, d12 = b1 - b2;
  const int n = 15 * d01 * d01 + 10 * d02 * d02 + 6 * d12 * d12;  // = 10 * (5 b0^2 + 3 b1^2 + 2 b2^2) - (5 b0 + 3 b1 + 2 b2)^2
  const int k = __float2int_rn(__builtin_amdgcn_sqrtf((float)n) * 0.1f);  // (v_sqrt_f32: 1 ulp, far inside the window below)
  if constexpr (NEAR) {
    const int lo = 10 * k - 5, hi = 10 * k + 5, w = 6 + (n >> 16);
    near = (k > 0 && n <= lo * lo + w) || n >= hi * hi - w;
  }
  return min(k, 255);
}

template <int G, int C>
__global__ __launch_bounds__(kBlock) void wmv_kernel(const FrameArgs a) {
  constexpr int NW = PxGroup<G, C>::NB / 4;
  static_assert(NW <= kWmvStageStride, "staging row too short");
  __shared__ uint32_t stage[kBlock * kWmvStageStride];
  __shared__ uint32_t list[kWmvListCap];  // b0 | b1 << 8 | b2 << 16 | byte index << 24, and who owns the byte:
  __shared__ uint8_t owner[kWmvListCap];
  __shared__ unsigned count;
  const size_t p0 = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = p0 < a.npix;
  PxGroup<G, C> d, x, y, z;
#pragma unroll
  for (int i = 0; i < NW; ++i) d.b.w[i] = x.b.w[i] = y.b.w[i] = z.b.w[i] = 0;
  if (active) {
    x.load(a.cur + p0 * C);
    y.load(a.p1 + p0 * C);
    z.load(a.p2 + p0 * C);
  }
  const bool fast = a.enable_weight != 0 && G * C <= 64;  // (the byte index travels in 6 bits)  [launch-uniform]
  if (!fast) {
    if (active) wmv_body<G, C>(x, y, z, a.enable_weight, a.enable_thr, a.thr, d);
  } else {
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    auto push = [&](int i, int b0, int b1, int b2) {
      const unsigned pos = atomicAdd(&count, 1u);
      if (pos < (unsigned)kWmvListCap) list[pos] = (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)i << 24), owner[pos] = (uint8_t)threadIdx.x;
    };
    const bool shortcut = a.enable_thr && a.thr >= 1;
#pragma unroll
    for (int j = 0; j < G; ++j) {
      bool quiet = shortcut;
      if (shortcut) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int b0 = x.b.get(j * C + c), b1 = y.b.get(j * C + c), b2 = z.b.get(j * C + c);
          quiet = quiet && (max(b0, max(b1, b2)) - min(b0, min(b1, b2)) < 2 * a.thr);
        }
      }
      if (quiet) continue;  // (a quiet pixel keeps its zero bytes: any value <= thr gives the same mask - wmv_body)
      int kk[C];
      bool nr[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int i = j * C + c;
        nr[c] = false;
        kk[c] = a.enable_thr ? wmv_fast_byte<false>(x.b.get(i), y.b.get(i), z.b.get(i), nr[c]) : wmv_fast_byte<true>(x.b.get(i), y.b.get(i), z.b.get(i), nr[c]);
        d.b.set(i, kk[c]);
      }
      if (a.enable_thr) {  // every byte is within 1 of k, gray's weights sum to 1: the mask is settled unless gray(k) is within 2 of thr
        int g;
        if constexpr (C == 3)
          g = gray_bgr(kk[0], kk[1], kk[2]);
        else
          g = kk[0];
        const bool band = g >= a.thr - 2 && g <= a.thr + 2;
#pragma unroll
        for (int c = 0; c < C; ++c) nr[c] = band;
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
        if (nr[c]) push(j * C + c, x.b.get(j * C + c), y.b.get(j * C + c), z.b.get(j * C + c));
    }
    __syncthreads();
    const unsigned n = count;  // (workgroup-uniform)
    if (n > (unsigned)kWmvListCap) {
      wmv_body<G, C>(x, y, z, a.enable_weight, a.enable_thr, a.thr, d);
    } else if (n > 0) {
#pragma unroll
      for (int i = 0; i < NW; ++i) stage[threadIdx.x * kWmvStageStride + i] = d.b.w[i];
      __syncthreads();
      uint8_t* sb = reinterpret_cast<uint8_t*>(stage);
      for (unsigned e = threadIdx.x; e < n; e += kBlock) {
        const uint32_t ent = list[e];
        sb[(owner[e] * kWmvStageStride) * 4 + (ent >> 24)] = (uint8_t)wmv_exact_byte((int)(ent & 0xffu), (int)((ent >> 8) & 0xffu), (int)((ent >> 16) & 0xffu), 0.5, 0.3, 0.2);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NW; ++i) d.b.w[i] = stage[threadIdx.x * kWmvStageStride + i];
    }
  }
  gray_thr_store<G, C>(d, a, p0, active, a.fg_bits != nullptr);
}

// AdaptiveBackgroundLearning.  The new background byte, sat_u8(addWeighted(i/255, alpha, b/255, 1-alpha) * 255) (:54-58), is a
// pure function of the two bytes for a fixed alpha, and its f64 arithmetic (~25 VALU instructions per byte) made the kernel
// issue-bound at 0.52-0.58 of the HBM peak (profiles/r01_bench_configs.txt; a 10 B/pixel stream leaves ~20 lane-instructions
// per byte before VALU issue, not HBM, sets the time).  So the 256 x 256 answers are tabulated ONCE per alpha by abl_lut_kernel
// - with exactly that arithmetic - and the frame kernel looks them up in LDS:
//   * table in LDS with a row stride of 260 bytes (65 dwords: rows of neighbouring background values start on different banks);
//   * 1024-lane persistent workgroups (2 per CU at 65 KB of LDS each) that walk the launch tile by tile, so the 64 KB table load
//     is paid once per workgroup, not once per 12 KB of frame;
//   * per byte: one ds_read_u8 + ~8 integer instructions (the mask side |i - b| -> gray -> threshold was integer already).
constexpr int kAblLutStride = 260;
constexpr int kAblBlock = 1024;
// 256 rows x 260 bytes = 66 560 B of static LDS: fine on gfx950 (160 KB per CU, the only target of this library - csrc/Makefile), above the
// 64 KiB a workgroup may have on earlier CDNA parts, where this kernel would need a 256-byte row stride
static_assert(256 * kAblLutStride <= 160 * 1024 / 2, "two resident workgroups per CU must fit the LDS of gfx950");

// lut[b * 256 + x] = the background byte AdaptiveBackgroundLearning.cpp:54-58 produces from input byte x and background byte b
__global__ __launch_bounds__(kBlock) void abl_lut_kernel(uint8_t* lut, double alpha, double beta) {
  const int x = threadIdx.x, b = blockIdx.x;
  lut[b * 256 + x] = (uint8_t)sat_u8(add_weighted(to_unit(x), alpha, to_unit(b), beta) * 255.f);
}

// the 64 KB table into LDS (row stride 260), all lanes of the workgroup; the caller synchronises
__device__ __forceinline__ void abl_load_lut(uint8_t* T, const uint8_t* __restrict__ lut, int nthreads) {
  const uint4* src = reinterpret_cast<const uint4*>(lut);
  for (int i = threadIdx.x; i < 256 * 16; i += nthreads) {  // 16 x 16-byte pieces per row; 260 = 65 dwords keeps dword alignment
    const uint4 v = src[i];
    uint32_t* dst = reinterpret_cast<uint32_t*>(T + (i >> 4) * kAblLutStride + (i & 15) * 16);
    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
  }
}
// AdaptiveBackgroundLearning.cpp:43-65 for G pixels: d = |I - B| per channel; with UPDATE the background bytes become the table's
template <int G, int C, bool UPDATE>
__device__ __forceinline__ void abl_body(const PxGroup<G, C>& x, PxGroup<G, C>& bgq, const uint8_t* T, PxGroup<G, C>& d) {
#pragma unroll
  for (int i = 0; i < G * C; ++i) {
    const int xi = x.b.get(i), bi = bgq.b.get(i);
    if constexpr (UPDATE) bgq.b.set(i, T[bi * kAblLutStride + xi]);  // :54-58
    // :50, :64-65: saturate(|i/255 - b/255| * 255) in float == |i - b| for all 65 536 byte pairs (checked exhaustively,
    // CPU test test_unit_absdiff_is_integer_absdiff), so the float round trip is skipped
    d.b.set(i, abs(xi - bi));
  }
}

template <int G, int C, bool UPDATE>
__global__ __launch_bounds__(kAblBlock) void abl_kernel(const FrameArgs a, const uint8_t* __restrict__ lut) {
  __shared__ uint8_t T[UPDATE ? 256 * kAblLutStride : 16];
  if constexpr (UPDATE) {
    abl_load_lut(T, lut, kAblBlock);
    __syncthreads();
  }
  const size_t per_tile = (size_t)kAblBlock * G;
  const size_t ntiles = (a.npix + per_tile - 1) / per_tile;
  const bool packed = a.fg_bits != nullptr;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t p0 = (tile * kAblBlock + threadIdx.x) * G;
    const bool active = p0 < a.npix;
    PxGroup<G, C> d;
    if (active) {
      PxGroup<G, C> x, bgq;
      x.load(a.cur + p0 * C);
      bgq.load(a.p1 + p0 * C);
      abl_body<G, C, UPDATE>(x, bgq, T, d);
      if constexpr (UPDATE) bgq.store(a.state_out + p0 * C);
      if (a.bg) bgq.store(a.bg + p0 * C);
    }
    gray_thr_store<G, C>(d, a, p0, active, packed);
  }
}

// K-N4 SigmaDeltaBGS::process over sdLaMa091 (package_bgs/bl/SigmaDeltaBGS.cpp:41-52, sdLaMa091.cpp:529-633): pure byte arithmetic,
// one fused pass instead of the reference's four.  p1/state_out = Mt (in place), p2w = Vt (in place).  16 B/pixel.
struct SigmaDeltaArgs {
  const uint8_t* cur;
  uint8_t* mt;
  uint8_t* vt;
  uint8_t* fg;
  uint64_t* fg_bits;
  size_t npix;
  uint32_t N;
  int vmin, vmax, xcd_swizzle;
};

// Two bytes per instruction: the byte pairs (0, 2) and (1, 3) of a dword as 16-bit lanes (v_pk_* on CDNA).  One step of the
// estimator on both lanes; returns 1 in a lane whose channel does NOT vote foreground.  N <= 256 so that N * ot fits 16 bits.
typedef short sd_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short sd_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sd_pair(uint32_t im2, uint32_t& m2, uint32_t& v2, uint32_t n2, uint32_t vmin2, uint32_t vmax2) {
  auto S = [](uint32_t w) { return __builtin_bit_cast(sd_s2, w); };
  auto U = [](uint32_t w) { return __builtin_bit_cast(sd_u2, w); };
  auto W = [](auto v) { return __builtin_bit_cast(uint32_t, v); };
  const sd_s2 one = {1, 1}, mone = {-1, -1};
  // sdLaMa091.cpp:535-540  Mt += sign(It - Mt)
  sd_s2 m = S(m2) + __builtin_elementwise_min(__builtin_elementwise_max(S(im2) - S(m2), mone), one);
  // :559 absVal((int8_t)(Mt - It)): u = (Mt - It) mod 256, |int8(u)| mod 256 = min(u, 256 - u)
  const sd_u2 u = U(W(m - S(im2)) & 0x00ff00ffu), c256 = {256, 256};
  const sd_u2 ot = __builtin_elementwise_min(u, (sd_u2)(c256 - u));
  // :576-581  Vt += sign(N * Ot - Vt) on a uint8 (255 + 1 wraps to 0); N * Ot only matters up to 256
  const sd_u2 amp = __builtin_elementwise_min((sd_u2)(ot * U(n2)), c256);
  sd_s2 v = S(v2) + __builtin_elementwise_min(__builtin_elementwise_max(S(W(amp)) - S(v2), mone), one);
  sd_u2 vu = U(W(v) & 0x00ff00ffu);
  vu = __builtin_elementwise_max(__builtin_elementwise_min(vu, U(vmax2)), U(vmin2));  // :583
  m2 = W(m), v2 = W(vu);
  // :605  foreground vote Ot >= Vt; the complement: bit 15 of (Ot - Vt) as 0 / 1 per lane
  return (W(S(W(ot)) - S(W(vu))) >> 15) & 0x00010001u;
}

// one sdLaMa091 step for G pixels (BGR): Mt and Vt updated in place, the mask bytes in `m`, one bit per pixel in `bits`
template <int G>
__device__ __forceinline__ void sd_body(const PxGroup<G, 3>& x, PxGroup<G, 3>& mt, PxGroup<G, 3>& vt, uint32_t N, int vmin, int vmax, PxGroup<G, 1>& m, uint32_t& bits) {
  if (G % 4 == 0 && N <= 256u) {
    // 4 pixels = 12 bytes = 3 dwords at a time, two bytes per packed 16-bit instruction (38 instead of 73 instructions per pixel)
    const uint32_t n2 = N * 0x10001u, vmin2 = (uint32_t)vmin * 0x10001u, vmax2 = (uint32_t)vmax * 0x10001u;
#pragma unroll
    for (int g = 0; g < G / 4; ++g) {
      uint32_t nf[3];  // per byte: 1 = this channel does not vote foreground
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int i = 3 * g + d;
        const uint32_t xw = x.b.w[i], mw = mt.b.w[i], vw = vt.b.w[i];
        uint32_t mlo = mw & 0x00ff00ffu, mhi = (mw >> 8) & 0x00ff00ffu, vlo = vw & 0x00ff00ffu, vhi = (vw >> 8) & 0x00ff00ffu;
        const uint32_t flo = sd_pair(xw & 0x00ff00ffu, mlo, vlo, n2, vmin2, vmax2);
        const uint32_t fhi = sd_pair((xw >> 8) & 0x00ff00ffu, mhi, vhi, n2, vmin2, vmax2);
        mt.b.w[i] = mlo | (mhi << 8), vt.b.w[i] = vlo | (vhi << 8);
        nf[d] = flo | (fhi << 8);
      }
      // pixel j of the group owns bytes 3j .. 3j+2 of the 12: background iff all three of its channels say so
      const uint32_t s0 = __builtin_amdgcn_sad_u8(nf[0] & 0x00ffffffu, 0u, 0u);
      const uint32_t s1 = __builtin_amdgcn_sad_u8(nf[0] & 0xff000000u, 0u, __builtin_amdgcn_sad_u8(nf[1] & 0x0000ffffu, 0u, 0u));
      const uint32_t s2 = __builtin_amdgcn_sad_u8(nf[1] & 0xffff0000u, 0u, __builtin_amdgcn_sad_u8(nf[2] & 0x000000ffu, 0u, 0u));
      const uint32_t s3 = __builtin_amdgcn_sad_u8(nf[2] & 0xffffff00u, 0u, 0u);
      const uint32_t f0 = s0 != 3u, f1 = s1 != 3u, f2 = s2 != 3u, f3 = s3 != 3u;
      m.b.w[g] = (f0 * 0xffu) | (f1 * 0xff00u) | (f2 * 0xff0000u) | (f3 * 0xff000000u);
      bits |= (f0 | (f1 << 1) | (f2 << 2) | (f3 << 3)) << (4 * g);
    }
  } else {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      bool isfg = false;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = 3 * j + c;
        int mv = mt.b.get(i);
        const int im = x.b.get(i);
        mv += (mv < im) - (mv > im);                                   // sdLaMa091.cpp:535-540
        const int d8 = (int)(int8_t)(uint8_t)(mv - im);                // :559 absVal(int8_t): the difference wraps to int8 first
        const uint32_t ot = (uint32_t)(d8 < 0 ? -d8 : d8) & 0xffu;
        const uint32_t amp = N * ot;                                 // :576
        uint32_t v = (uint32_t)vt.b.get(i);
        v = (v + (v < amp) - (v > amp)) & 0xffu;                       // :578-581 on a uint8: 255+1 wraps to 0
        v = min(v, (uint32_t)vmax);                                  // :583 max(min(Vt, Vmax), Vmin) with uint8 operands
        v = max(v, (uint32_t)vmin);
        isfg = isfg || ot >= v;                                        // :605
        mt.b.set(i, mv);
        vt.b.set(i, (int)v);
      }
      m.b.set(j, isfg ? 255 : 0);
      bits |= (uint32_t)isfg << j;
    }
  }
}

template <int G>
__global__ __launch_bounds__(kBlock) void sigmadelta_kernel(const SigmaDeltaArgs a) {
  const size_t p0 = (xcd_block(a.xcd_swizzle) * kBlock + threadIdx.x) * G;
  const bool active = p0 < a.npix;
  PxGroup<G, 1> m;
#pragma unroll
  for (int i = 0; i < PxGroup<G, 1>::NB / 4; ++i) m.b.w[i] = 0;
  uint32_t bits = 0;
  if (active) {
    PxGroup<G, 3> x, mt, vt;
    x.load(a.cur + p0 * 3);
    mt.load(a.mt + p0 * 3);
    vt.load(a.vt + p0 * 3);
    sd_body<G>(x, mt, vt, a.N, a.vmin, a.vmax, m, bits);
    mt.store(a.mt + p0 * 3);
    vt.store(a.vt + p0 * 3);
    if (a.fg) m.store(a.fg + p0);
  }
  if (a.fg_bits) {
    if constexpr (64 % G == 0) store_packed_mask<G>(a.fg_bits, p0, bits, active);
  }
}

// Vt initialisation exactly as sdLaMa091AllocInit_8u_C3R leaves it (see the quirk note in DESIGN.md §5): byte j of every
// 3*cols-byte row is Vmin for j < cols, 0 beyond.
__global__ __launch_bounds__(kBlock) void sigmadelta_init_vt_kernel(uint8_t* vt, size_t nbytes, int cols, int vmin) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nbytes) return;
  vt[i] = (int)(i % ((size_t)3 * cols)) < cols ? (uint8_t)vmin : (uint8_t)0;
}

}  // namespace bgs
