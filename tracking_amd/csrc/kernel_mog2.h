// kernel_mog2.h — K4: MixtureOfGaussianV2BGS (cv::BackgroundSubtractorMOG2 update + classify + wrapper threshold,
// optionally the getBackgroundImage pass) as ONE pointwise CDNA4 kernel.
//
// Replaces  MixtureOfGaussianV2BGS::process  package_bgs/MixtureOfGaussianV2BGS.cpp:56-62
//           (mog(img, fg, alpha); mog.getBackgroundImage(bg); cv::threshold(fg, 15))
// Algorithm: Zivkovic adaptive GMM as implemented by OpenCV 2.4 bgfg_gaussmix2.cpp MOG2Invoker.
//
// Model layout, round 3: RANKED WEIGHTS + FIXED SLOTS.
//   The reference keeps each pixel's K = 5 modes physically sorted by weight: the matched mode bubbles towards the front and
//   drags every mode it passes - {weight, variance, mean[3]} = 20 bytes each - through the swap.  On a busy scene that makes all
//   100 bytes of a pixel's model dirty every frame although only ONE mode's variance and mean were recomputed.  Here
//     * the WEIGHTS stay in rank order (plane r = weight of the r-th heaviest mode; every live weight changes every frame anyway);
//     * a mode's {variance, mean[3]} is a 16-byte RECORD that lives in a fixed SLOT from the frame the mode is created until it
//       is replaced (slots are handed out in creation order: a pixel with n modes owns slots 0..n-1);
//     * a 16-bit META word per pixel maps rank -> slot: bits 3r..3r+2 = slot + 1 of rank r, 0 = rank unused (so a zeroed model is
//       an empty model, BackgroundSubtractorMOG2::initialize).  modesUsed = number of non-zero fields.
//   A frame then reads 3 (frame) + 2 (meta) + 20 (weights) + 80 (records) and writes 20 (weights) + 16 (the one record that was
//   updated or created) + 2 (meta, when the order changed) + 1 (mask) = 144 B/pixel instead of 206.
//   The arithmetic is the reference's statement for statement: the records are gathered into rank order in registers, the same
//   predicated bubble runs there (the slot ids travel with it), and only what changed goes back.
// Tiles: pixels are grouped in tiles of kMog2Tile (256); a tile is 5 weight planes (T floats each), 5 record planes (T float4
//   each) and T meta words = 102 T contiguous bytes.  One workgroup owns one tile, one lane one pixel: every access is a
//   coalesced wave instruction (dwordx4 for the records), and with the XCD-aware block order each XCD streams one contiguous
//   eighth of the model.
// Stores are SECTOR-COMPLETE (args.complete): HBM moves 32-byte sectors, so a lane also writes back an unchanged value of its own
//   when another lane of the same sector (2 lanes of a record plane, 8 of a weight plane, 16 of the meta row) has something to
//   write there - no partially written sector ever reaches the memory controller.
// No LDS: the op is pointwise and HBM-bound.
#pragma once
#include "bgs_device.h"

namespace bgs {

constexpr int kMog2K = 5;
#ifndef BGS_MOG2_TILE
#define BGS_MOG2_TILE 256
#endif
constexpr int kMog2Tile = BGS_MOG2_TILE;                                          // pixels per tile
constexpr size_t kMog2TileBytes = (size_t)kMog2Tile * (4 * kMog2K + 16 * kMog2K + 2);  // 102 B per pixel: 26 112 B
constexpr size_t kMog2RecOff = (size_t)kMog2Tile * 4 * kMog2K;                    // byte offset of record plane 0 inside a tile
constexpr size_t kMog2MetaOff = kMog2RecOff + (size_t)kMog2Tile * 16 * kMog2K;   // byte offset of the meta row
static_assert(kMog2Tile % 64 == 0, "a wave never straddles two tiles");

struct Mog2Args {
  const uint8_t* frame;  // [P][3] interleaved BGR
  uint8_t* fg;           // [P] or null
  uint8_t* bgimg;        // [P][3] or null
  uint64_t* fg_bits;     // [P/64] or null
  uint8_t* state;        // model tiles (updated in place)
  size_t state_off;      // first pixel of this launch inside the model
  size_t npix;           // pixels in this launch
  float alphaT, alpha1, prune;
  float Tb, TB, Tg, varInit, varMin, varMax, tau;
  int thr, enable_thr, shadow_val;
  int shadow, want_bg, packed;  // wave-uniform feature switches
  unsigned* stat;               // null, or 2 counters: sampled waves, sampled waves whose largest mode count is below K-1 (auto mode)
  unsigned stat_mask;           // workgroups with (blockIdx.x & stat_mask) == 0 are sampled (~256 per launch)
  int sparse;                   // 0 dense: every weight, record and meta word loaded and written back (placement probe, A/B);
                                // 1 everything loaded, only what changed written; >= 2 a lane also loads only the modes its pixel has
  int complete;                 // sector-complete stores (see above)
  int xcd_swizzle;              // workgroups that share an XCD walk one contiguous eighth of the launch
};

struct Mog2Ptr {
  float* w;        // weight of rank r at w[r * kMog2Tile]
  float4* rec;     // record of slot s at rec[s * kMog2Tile]: {variance, mean0, mean1, mean2}
  uint16_t* meta;
};
__device__ __forceinline__ Mog2Ptr mog2_ptr(uint8_t* state, size_t sp) {
  uint8_t* tb = state + (sp / kMog2Tile) * kMog2TileBytes;
  const size_t in = sp % kMog2Tile;
  Mog2Ptr p;
  p.w = reinterpret_cast<float*>(tb) + in;
  p.rec = reinterpret_cast<float4*>(tb + kMog2RecOff) + in;
  p.meta = reinterpret_cast<uint16_t*>(tb + kMog2MetaOff) + in;
  return p;
}
// modesUsed of a meta word: its non-zero 3-bit fields (always a prefix)
__device__ __host__ __forceinline__ int mog2_meta_count(unsigned meta) {
  unsigned t = (meta | (meta >> 1) | (meta >> 2)) & 0x1249u;
  t = (t & 1u) + ((t >> 3) & 1u) + ((t >> 6) & 1u) + ((t >> 9) & 1u) + ((t >> 12) & 1u);
  return (int)t;
}

// One pixel's model in rank order, as MOG2Invoker sees it; sl[r] = slot + 1 of the mode at rank r (0: rank unused)
struct Mog2Px {
  float w[kMog2K], var[kMog2K], m0[kMog2K], m1[kMog2K], m2[kMog2K];
  int sl[kMog2K];
};

__device__ __forceinline__ void mog2_swap(Mog2Px& s, int i, int j) {
  float t;
  t = s.w[i], s.w[i] = s.w[j], s.w[j] = t;
  t = s.var[i], s.var[i] = s.var[j], s.var[j] = t;
  t = s.m0[i], s.m0[i] = s.m0[j], s.m0[j] = t;
  t = s.m1[i], s.m1[i] = s.m1[j], s.m1[j] = t;
  t = s.m2[i], s.m2[i] = s.m2[j], s.m2[j] = t;
  const int u = s.sl[i];
  s.sl[i] = s.sl[j], s.sl[j] = u;
}

// detectShadowGMM of bgfg_gaussmix2.cpp, predicated form of its early returns
__device__ __forceinline__ bool mog2_shadow(const Mog2Px& s, int nmodes, float x0, float x1, float x2, const Mog2Args& a) {
  bool done = false, result = false;
  float tWeight = 0.f;
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode) {
    if (mode < nmodes && !done) {
      float num = 0.0f, den = 0.0f;
      num += x0 * s.m0[mode];
      den += s.m0[mode] * s.m0[mode];
      num += x1 * s.m1[mode];
      den += s.m1[mode] * s.m1[mode];
      num += x2 * s.m2[mode];
      den += s.m2[mode] * s.m2[mode];
      if (den == 0) {
        done = true;
      } else {
        if (num <= den && num >= a.tau * den) {
          const float q = div_rn(num, den);
          float d2a = 0.0f, dD;
          dD = q * s.m0[mode] - x0, d2a += dD * dD;
          dD = q * s.m1[mode] - x1, d2a += dD * dD;
          dD = q * s.m2[mode] - x2, d2a += dD * dD;
          if (d2a < a.Tb * s.var[mode] * q * q) result = true, done = true;
        }
        if (!done) {
          tWeight += s.w[mode];
          if (tWeight > a.TB) done = true;
        }
      }
    }
  }
  return result;
}

// One pixel of MOG2Invoker::operator() — same statement order as the reference so every float rounds identically.
// Returns the raw mask value (0 background, shadow_val, 255 foreground) before the wrapper's threshold.
// alphaT / alpha1 / prune are the learning-rate terms of THIS frame (they differ between the frames of a clip launch).
// `dirty`: bit (slot + 1) is set for the slot whose record this frame recomputed (the matched mode) or created.
__device__ __forceinline__ int mog2_pixel(Mog2Px& s, int& nmodes_io, float x0, float x1, float x2, const Mog2Args& a, unsigned& dirty,
                                          const float alphaT, const float alpha1, const float prune) {
  bool background = false, fitsPDF = false;
  int nmodes = nmodes_io;
  const int nNewModes = nmodes;
  float totalWeight = 0.f;
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode) {
    if (mode < nmodes) {  // nmodes shrinks inside the loop when a mode is pruned (reference quirk)
      float weight = alpha1 * s.w[mode] + prune;
      bool matched = false;
      if (!fitsPDF) {
        const float var = s.var[mode];
        const float d0 = s.m0[mode] - x0, d1 = s.m1[mode] - x1, d2 = s.m2[mode] - x2;
        const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
        if (totalWeight < a.TB && dist2 < a.Tb * var) background = true;
        if (dist2 < a.Tg * var) {
          fitsPDF = true;
          matched = true;
          weight += alphaT;
          const float k = div_rn(alphaT, weight);
          s.m0[mode] -= k * d0;
          s.m1[mode] -= k * d1;
          s.m2[mode] -= k * d2;
          float varnew = var + k * (dist2 - var);
          varnew = varnew > a.varMin ? varnew : a.varMin;
          varnew = varnew < a.varMax ? varnew : a.varMax;
          s.var[mode] = varnew;
          dirty |= 1u << s.sl[mode];  // the record of this mode's slot changed
        }
      }
      const bool pruned = weight < -prune;
      if (pruned) nmodes--;
      // The reference stores the weight at gmm[mode - swap_count] after the bubble; storing it first and letting it
      // travel with the swaps is the same thing.  The bubble compares the UNPRUNED weight, as the reference does.
      s.w[mode] = pruned ? 0.f : weight;
      if (matched) {
        bool moving = true;
#pragma unroll
        for (int i = mode; i > 0; --i) {
          moving = moving && !(weight < s.w[i - 1]);
          if (moving) mog2_swap(s, i, i - 1);
        }
      }
      totalWeight += pruned ? 0.f : weight;
    }
  }
  totalWeight = div_rn(1.f, totalWeight);
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode)
    if (mode < nmodes) s.w[mode] *= totalWeight;
  nmodes = nNewModes;  // sic: the pruned count is discarded
  if (!fitsPDF) {
    // the reference overwrites gmm[K-1] when the array is full, else appends: the new mode takes over the slot of the mode it
    // replaces, or the next free slot (slots are handed out in creation order)
    const int slot = (nmodes == kMog2K) ? s.sl[kMog2K - 1] : nmodes + 1;
    const int mode = (nmodes == kMog2K) ? kMog2K - 1 : nmodes++;
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) {
      if (k == mode) {
        s.w[k] = (nmodes == 1) ? 1.f : alphaT;
        s.m0[k] = x0, s.m1[k] = x1, s.m2[k] = x2;
        s.var[k] = a.varInit;
        s.sl[k] = slot;
      } else if (nmodes != 1 && k < nmodes - 1) {
        s.w[k] *= alpha1;
      }
    }
    dirty |= 1u << slot;
    bool moving = true;
#pragma unroll
    for (int i = kMog2K - 1; i > 0; --i) {
      if (i <= nmodes - 1) {
        moving = moving && !(alphaT < s.w[i - 1]);
        if (moving) mog2_swap(s, i, i - 1);
      }
    }
  }
  nmodes_io = nmodes;
  if (background) return 0;
  if (a.shadow) {
    if (mog2_shadow(s, nmodes, x0, x1, x2, a)) return a.shadow_val;
  }
  return 255;
}

// cv::BackgroundSubtractorMOG2::getBackgroundImage, one pixel, from the registers that already hold the model
__device__ __forceinline__ void mog2_background(const Mog2Px& s, int nmodes, float TB, int& b0, int& b1, int& b2) {
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, totalWeight = 0.f;
  bool stop = false;
#pragma unroll
  for (int g = 0; g < kMog2K; ++g) {
    if (g < nmodes && !stop) {
      const float w = s.w[g];
      v0 += w * s.m0[g];
      v1 += w * s.m1[g];
      v2 += w * s.m2[g];
      totalWeight += w;
      if (totalWeight > TB) stop = true;
    }
  }
  const float inv = div_rn(1.f, totalWeight);
  b0 = sat_u8(v0 * inv), b1 = sat_u8(v1 * inv), b2 = sat_u8(v2 * inv);
}

// lane-crossing OR for the sector-complete stores (DPP row operations, no LDS traffic): returns the OR over the lane's aligned
// group of 2 / 8 / 16 lanes.  quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140;
// lanes that left the kernel (past npix) read as 0 (bound_ctrl).
__device__ __forceinline__ void mog2_group_or(unsigned v, unsigned& or2, unsigned& or8, unsigned& or16) {
  v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
  or2 = v;
  v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
  or8 = v;
  v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);
  or16 = v;
}

// T consecutive frames of one pixel per lane (T = 1: the per-frame launch; 2 / 4 / 8: clip launches, bgs_process_clip_device).
// The model of a pixel is loaded once, updated T times in registers in frame order with exactly the statements of the reference,
// and what changed is written back once: results are those of T successive single-frame launches, bit for bit, because a pixel's
// update depends on nothing but its own model and its own input.
constexpr int kMog2ClipMax = 8;
struct Mog2ClipArgs {
  Mog2Args m;                  // frame / fg / bgimg / fg_bits point at the FIRST frame of the launch
  size_t frame_stride;         // bytes from one frame to the next (= pixels of the whole clip slab * 3), likewise below
  size_t fg_stride, bg_stride, bits_stride;  // bits_stride in 64-bit words
  float alphaT[kMog2ClipMax], alpha1[kMog2ClipMax], prune[kMog2ClipMax];  // per frame (the automatic rate changes with the frame count)
};

template <int T>
__device__ __forceinline__ void mog2_body(const Mog2Args& a, const size_t frame_stride, const size_t fg_stride, const size_t bg_stride, const size_t bits_stride,
                                          const float* alphaT, const float* alpha1, const float* prune) {
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 says which blocks share an XCD and its L2).  Give
    // each XCD one CONTIGUOUS eighth of the launch instead of every 8th block, so every XCD reads and writes a single
    // sequential run of the model (measured +16 % on HBM; placement only ever changes speed, never results).
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = blk * kBlock + threadIdx.x;  // this lane's pixel, launch-relative
  if (p0 >= a.npix) return;                      // wave-uniform whenever masks are bit-packed (npix % 64 == 0, checked by the host)
  const size_t sp = a.state_off + p0;            // index inside the model
  uint32_t pix[T];  // all T inputs of this pixel are requested before the model: they are what the first update waits for
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint8_t* f = a.frame + (size_t)t * frame_stride + p0 * 3;
    pix[t] = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16);
  }
  const Mog2Ptr mp = mog2_ptr(a.state, sp);
  const unsigned meta_in = *mp.meta;
  // Loads.  sparse < 2: everything, independent of the meta word (all the loads of a lane are in flight at once: what a busy
  // scene wants).  sparse >= 2: rank 0 / slot 0 at once (every pixel has them after its first frame), the rest only for the modes
  // this pixel has - those loads wait for the meta word, which pays when most pixels have one or two modes (quiet scenes).
  float wv[kMog2K];
  float4 rec[kMog2K];
  const int nm_in = mog2_meta_count(meta_in);
  const bool lazy = a.sparse >= 2;
  wv[0] = mp.w[0];
  rec[0] = mp.rec[0];
#pragma unroll
  for (int k = 1; k < kMog2K; ++k) {
    if (!lazy || k < nm_in) {
      wv[k] = mp.w[(size_t)k * kMog2Tile];
      rec[k] = mp.rec[(size_t)k * kMog2Tile];
    } else {
      wv[k] = 0.f;
      rec[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (a.stat && (blockIdx.x & a.stat_mask) == 0) {  // scene-sparsity sample for the engine's automatic choice between sparse 1 and 4
    const bool dense_wave = __any(nm_in >= kMog2K - 1);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      atomicAdd(a.stat, 1u);
      if (!dense_wave) atomicAdd(a.stat + 1, 1u);
    }
  }
  // records into rank order (what the reference's array order is)
  Mog2Px s;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) {
    const int f = (int)((meta_in >> (3 * r)) & 7u);
    s.sl[r] = f;
    s.w[r] = wv[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < kMog2K; ++q)
      if (f == q + 1) v = rec[q];
    s.var[r] = v.x, s.m0[r] = v.y, s.m1[r] = v.z, s.m2[r] = v.w;
  }
  unsigned dirty = 0;
  int nm = nm_in;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x0 = (float)(pix[t] & 0xffu), x1 = (float)((pix[t] >> 8) & 0xffu), x2 = (float)(pix[t] >> 16);
    const int raw = mog2_pixel(s, nm, x0, x1, x2, a, dirty, alphaT[t], alpha1[t], prune[t]);
    const int m = thr_bin(raw, a.thr, a.enable_thr);
    if (a.fg) a.fg[(size_t)t * fg_stride + p0] = (uint8_t)m;
    if (a.packed) store_packed_mask<1>(a.fg_bits + (size_t)t * bits_stride, p0, (uint32_t)(m != 0), true);
    if (a.want_bg) {
      int b0, b1, b2;
      mog2_background(s, nm, a.TB, b0, b1, b2);
      uint8_t* o = a.bgimg + (size_t)t * bg_stride + p0 * 3;
      o[0] = (uint8_t)b0, o[1] = (uint8_t)b1, o[2] = (uint8_t)b2;
    }
  }
  // what changed: bits 0..4 weight of rank r, bit 5 the meta word, bits 6..10 the record of slot s
  unsigned meta_out = 0;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) meta_out |= (unsigned)s.sl[r] << (3 * r);
  unsigned d = 0;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) d |= (unsigned)(s.w[r] != wv[r]) << r;
  d |= (unsigned)(meta_out != meta_in) << 5;
  d |= (dirty >> 1) << 6;
  if (a.sparse == 0) {
    d = 0x7ffu;
  } else if (a.complete) {
    // whole 32-byte sectors or nothing: 2 lanes share a sector of a record plane, 8 of a weight plane, 16 of the meta row
    unsigned d2, d8, d16;
    mog2_group_or(d, d2, d8, d16);
    d = (d2 & 0x7c0u) | (d8 & 0x1fu) | (d16 & 0x20u);
  }
#pragma unroll
  for (int r = 0; r < kMog2K; ++r)
    if ((d >> r) & 1u) mp.w[(size_t)r * kMog2Tile] = s.w[r];
  if ((d >> 5) & 1u) *mp.meta = (uint16_t)meta_out;
#pragma unroll
  for (int q = 0; q < kMog2K; ++q) {
    if ((d >> (6 + q)) & 1u) {
      // the record of slot q as it stands now: wherever the bubble left it (a slot this pixel does not own holds nothing anyone reads)
      float4 v = rec[q];
#pragma unroll
      for (int r = 0; r < kMog2K; ++r)
        if (s.sl[r] == q + 1) v = make_float4(s.var[r], s.m0[r], s.m1[r], s.m2[r]);
      mp.rec[(size_t)q * kMog2Tile] = v;
    }
  }
}

// grid: ceil(npix / kBlock) blocks of kBlock lanes, one pixel per lane
__global__ __launch_bounds__(kBlock) void mog2_update_kernel(const Mog2Args a) {
  mog2_body<1>(a, 0, 0, 0, 0, &a.alphaT, &a.alpha1, &a.prune);
}

template <int T>
__global__ __launch_bounds__(kBlock) void mog2_clip_kernel(const Mog2ClipArgs c) {
  mog2_body<T>(c.m, c.frame_stride, c.fg_stride, c.bg_stride, c.bits_stride, c.alphaT, c.alpha1, c.prune);
}

// (re)initialisation of a pixel range: bgmodel = zeros, modesUsed = 0 (BackgroundSubtractorMOG2::initialize)
__global__ __launch_bounds__(kBlock) void mog2_clear_kernel(const Mog2Args a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npix) return;
  const Mog2Ptr mp = mog2_ptr(a.state, a.state_off + p);
#pragma unroll
  for (int k = 0; k < kMog2K; ++k) {
    mp.w[(size_t)k * kMog2Tile] = 0.f;
    mp.rec[(size_t)k * kMog2Tile] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  *mp.meta = 0;
}

}  // namespace bgs
