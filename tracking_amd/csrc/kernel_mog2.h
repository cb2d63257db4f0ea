// kernel_mog2.h — K4: MixtureOfGaussianV2BGS (cv::BackgroundSubtractorMOG2 update + classify + wrapper threshold,
// optionally the getBackgroundImage pass) as ONE pointwise CDNA4 kernel.
//
// Replaces  MixtureOfGaussianV2BGS::process  package_bgs/MixtureOfGaussianV2BGS.cpp:56-62
//           (mog(img, fg, alpha); mog.getBackgroundImage(bg); cv::threshold(fg, 15))
// Algorithm: Zivkovic adaptive GMM as implemented by OpenCV 2.4 bgfg_gaussmix2.cpp MOG2Invoker.
//
// Layout: SoA planes in HBM, P = streams*pixels floats per plane:
//   w[k][P], var[k][P], mu[k][c][P] (k < K = 5, c < 3), nmodes u8[P].  Modes of a pixel are kept sorted by weight
//   (descending) exactly like the reference's per-pixel GMM array, so plane k holds every pixel's k-th heaviest mode.
// Mapping: one lane owns PX consecutive pixels -> every plane access is one PX*4-byte vector load/store per lane
//   (PX = 4: global_load_dwordx4, 1 KiB per wave instruction), the frame is PX*3 bytes per lane, the mask PX bytes.
//   The whole per-pixel model (25 floats) lives in VGPRs; K is a compile-time constant so the insertion sort is a
//   fully unrolled network of predicated swaps.  No LDS: the op is pointwise and HBM-bound
//   (206 B/pixel/frame algorithmic traffic vs ~400 VALU ops).
#pragma once
#include "bgs_device.h"

namespace bgs {

constexpr int kMog2K = 5;

struct Mog2Args {
  const uint8_t* frame;  // [P][3] interleaved BGR
  uint8_t* fg;           // [P] or null
  uint8_t* bgimg;        // [P][3] or null
  uint64_t* fg_bits;     // [P/64] or null
  float* state;          // model, layout below (updated in place)
  uint8_t* nmodes_planar;// planar layout only: [plane] bytes
  size_t plane;          // planar layout only: floats per plane (= streams * pixels of the engine)
  size_t state_off;      // first pixel of this launch inside the model
  size_t npix;           // pixels in this launch
  float alphaT, alpha1, prune;
  float Tb, TB, Tg, varInit, varMin, varMax, tau;
  int thr, enable_thr, shadow_val;
  int shadow, want_bg, packed;  // wave-uniform feature switches
  unsigned* stat;               // null, or 2 counters: sampled waves, sampled waves whose largest nmodes is below K-1 (auto mode)
  unsigned stat_mask;           // workgroups with (blockIdx.x & stat_mask) == 0 are sampled (~256 per launch)
  int sparse;                   // 0 dense; 1 skip the stores of planes nothing changed in; 2 also skip the loads of modes no pixel of the wave has;
                                // 4 the same per lane (4 pixels) instead of per wave: partial rows, traffic follows the live modes
  int xcd_swizzle;              // workgroups that share an XCD walk one contiguous eighth of the launch
};

// Model layouts.  25 float "planes" per pixel: w[k] = k, var[k] = 5+k, mu[k][c] = 10+3k+c, plus nmodes (u8).
//   TILED  (default): AoSoA — pixels are grouped in tiles of 256; a tile is 25 x 256 floats followed by 256 nmodes bytes
//            (25 856 B, contiguous).  One wave (PX = 4) owns one tile: it streams ONE contiguous 25 KB block in and out,
//            every access still a coalesced 1 KiB wave instruction.  With the XCD-aware block order each of the 8 XCDs
//            then reads and writes one sequential stream, which is what HBM likes best (measured: DESIGN.md §6).
//   PLANAR : 25 planes of P floats + one plane of P bytes (52 concurrent DRAM streams; kept for A/B measurements).
constexpr int kMog2Planes = 25;
constexpr int kMog2Tile = 256;                                            // pixels per tile
constexpr int kMog2TileFloats = kMog2Planes * kMog2Tile + kMog2Tile / 4;  // 6464 floats = 25 856 B

template <bool TILED>
__device__ __forceinline__ size_t mog2_plane_off(const Mog2Args& a, int p, size_t sp) {
  if constexpr (TILED)
    return (sp >> 8) * kMog2TileFloats + p * kMog2Tile + (sp & 255);
  else
    return (size_t)p * a.plane + sp;
}
template <bool TILED>
__device__ __forceinline__ uint8_t* mog2_nmodes(const Mog2Args& a, size_t sp) {
  if constexpr (TILED)
    return reinterpret_cast<uint8_t*>(a.state + (sp >> 8) * kMog2TileFloats + kMog2Planes * kMog2Tile) + (sp & 255);
  else
    return a.nmodes_planar + sp;
}

struct Mog2Px {
  float w[kMog2K], var[kMog2K], m0[kMog2K], m1[kMog2K], m2[kMog2K];
};

__device__ __forceinline__ void mog2_swap(Mog2Px& s, int i, int j) {
  float t;
  t = s.w[i], s.w[i] = s.w[j], s.w[j] = t;
  t = s.var[i], s.var[i] = s.var[j], s.var[j] = t;
  t = s.m0[i], s.m0[i] = s.m0[j], s.m0[j] = t;
  t = s.m1[i], s.m1[i] = s.m1[j], s.m1[j] = t;
  t = s.m2[i], s.m2[i] = s.m2[j], s.m2[j] = t;
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
          dirty |= 1u << mode;  // mean / variance of this mode changed
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
          if (moving) mog2_swap(s, i, i - 1), dirty |= 3u << (i - 1);
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
    const int mode = (nmodes == kMog2K) ? kMog2K - 1 : nmodes++;
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) {
      if (k == mode) {
        s.w[k] = (nmodes == 1) ? 1.f : alphaT;
        s.m0[k] = x0, s.m1[k] = x1, s.m2[k] = x2;
        s.var[k] = a.varInit;
        dirty |= 1u << k;
      } else if (nmodes != 1 && k < nmodes - 1) {
        s.w[k] *= alpha1;
      }
    }
    bool moving = true;
#pragma unroll
    for (int i = kMog2K - 1; i > 0; --i) {
      if (i <= nmodes - 1) {
        moving = moving && !(alphaT < s.w[i - 1]);
        if (moving) mog2_swap(s, i, i - 1), dirty |= 3u << (i - 1);
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

// grid: ceil(npix / PX / kBlock) blocks of kBlock lanes; npix % PX == 0 (the host picks PX = 1 otherwise).
template <int PX, bool TILED>
__global__ __launch_bounds__(kBlock) void mog2_update_kernel(const Mog2Args a) {
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 says which blocks share an XCD and its L2).  Give
    // each XCD one CONTIGUOUS eighth of the launch instead of every 8th block, so every XCD reads and writes a single
    // sequential run of the model (measured +16 % on HBM; placement only ever changes speed, never results).
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t g = blk * kBlock + threadIdx.x;
  const size_t p0 = g * PX;  // first pixel of this lane, launch-relative
  const bool active = p0 < a.npix;
  uint32_t bits = 0;
  if (active) {
    const size_t sp = a.state_off + p0;  // index inside the model
    constexpr int FB = (PX * 3 + 3) / 4 * 4;  // frame bytes per lane, rounded up to dwords
    Bytes<FB> pix;
    if constexpr (PX == 1) {
      const uint8_t* f = a.frame + p0 * 3;
      pix.w[0] = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16);
    } else if constexpr (PX == 2) {
      const uint16_t* f = reinterpret_cast<const uint16_t*>(a.frame + p0 * 3);
      pix.w[0] = (uint32_t)f[0] | ((uint32_t)f[1] << 16);
      pix.w[1] = f[2];
    } else {
      pix = load_bytes<PX * 3>(a.frame + p0 * 3);
    }
    uint8_t* const nmp = mog2_nmodes<TILED>(a, sp);
    uint32_t nmw;
    if constexpr (PX == 4)
      nmw = *reinterpret_cast<const uint32_t*>(nmp);
    else if constexpr (PX == 2)
      nmw = *reinterpret_cast<const uint16_t*>(nmp);
    else
      nmw = *nmp;
    // Data-dependent traffic (exact).  sparse >= 1: a plane is written back only if some pixel of the wave changed it.
    // sparse == 2: the algorithm never touches modes at index >= nmodes(pixel) except to create one AT index nmodes, so with
    // M = the largest nmodes in this wave only modes 0..min(M, K-1) are loaded at all.  That makes the plane loads depend on
    // the nmodes load (one extra memory round trip per wave): a clear win on sparse scenes (x1.9 on S_surv), a loss of ~8 %
    // when every mode is live, hence opt-in; mode 0 is requested before M is known so part of the latency overlaps.
    float st[kMog2Planes][PX];
    auto load_mode = [&](int k) {
      load_f<PX>(a.state + mog2_plane_off<TILED>(a, k, sp), st[k]);
      load_f<PX>(a.state + mog2_plane_off<TILED>(a, 5 + k, sp), st[5 + k]);
#pragma unroll
      for (int c = 0; c < 3; ++c) load_f<PX>(a.state + mog2_plane_off<TILED>(a, 10 + 3 * k + c, sp), st[10 + 3 * k + c]);
    };
    int nload = kMog2K;
    load_mode(0);
    // sparse == 4: the same two rules per LANE (its PX pixels) instead of per wave.  With L = the largest nmodes among the
    // lane's pixels, the lane loads modes 0..L-1 only and stores a plane only if one of its own pixels changed it.  The one
    // unloaded slot a pixel can still write is index L (a pixel with nmodes == L creating a mode): the lane's other pixels
    // keep stale, unread entries there, so that slot is written per pixel (scalar stores), never as a vector.
    const bool lanewise = a.sparse >= 4;
    int lane_need = kMog2K;
    if (a.sparse >= 2) {
      int lane_max = 0;
#pragma unroll
      for (int j = 0; j < PX; ++j) lane_max = max(lane_max, (int)((nmw >> (8 * j)) & 0xffu));
      if (lanewise) lane_need = lane_max - 1;
      int M = 0;
#pragma unroll
      for (int n = 1; n <= kMog2K; ++n)
        if (__any(lane_max >= n)) M = n;
      nload = min(M + 1, kMog2K);
    }
    if (a.stat && (blockIdx.x & a.stat_mask) == 0) {  // scene-sparsity sample for the engine's automatic choice between sparse 1 and 4
      int lane_max = 0;
#pragma unroll
      for (int j = 0; j < PX; ++j) lane_max = max(lane_max, (int)((nmw >> (8 * j)) & 0xffu));
      const bool dense_wave = __any(lane_max >= kMog2K - 1);
      if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(a.stat, 1u);
        if (!dense_wave) atomicAdd(a.stat + 1, 1u);
      }
    }
#pragma unroll
    for (int k = 1; k < kMog2K; ++k) {
      if (k < nload && k <= lane_need) {
        load_mode(k);
      } else {
#pragma unroll
        for (int j = 0; j < PX; ++j) st[k][j] = 0.f, st[5 + k][j] = 0.f, st[10 + 3 * k][j] = 0.f, st[11 + 3 * k][j] = 0.f, st[12 + 3 * k][j] = 0.f;
      }
    }

    uint32_t mask_word = 0, nm_out = 0;
    unsigned dirty_m = 0, dirty_w = 0;  // per mode: mean/variance changed, weight changed (any of this lane's pixels)
    Bytes<FB> bgout;
#pragma unroll
    for (int i = 0; i < FB / 4; ++i) bgout.w[i] = 0;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      Mog2Px s;
#pragma unroll
      for (int k = 0; k < kMog2K; ++k)
        s.w[k] = st[k][j], s.var[k] = st[5 + k][j], s.m0[k] = st[10 + 3 * k][j], s.m1[k] = st[11 + 3 * k][j], s.m2[k] = st[12 + 3 * k][j];
      int nm = (int)((nmw >> (8 * j)) & 0xffu);
      const float x0 = (float)pix.get(3 * j), x1 = (float)pix.get(3 * j + 1), x2 = (float)pix.get(3 * j + 2);
      float worig[kMog2K];
#pragma unroll
      for (int k = 0; k < kMog2K; ++k) worig[k] = s.w[k];
      const int raw = mog2_pixel(s, nm, x0, x1, x2, a, dirty_m, a.alphaT, a.alpha1, a.prune);
#pragma unroll
      for (int k = 0; k < kMog2K; ++k) dirty_w |= (unsigned)(s.w[k] != worig[k]) << k;
      const int m = thr_bin(raw, a.thr, a.enable_thr);
      mask_word |= (uint32_t)m << (8 * j);
      bits |= (uint32_t)(m != 0) << j;
      nm_out |= (uint32_t)nm << (8 * j);
      if (a.want_bg) {
        int b0, b1, b2;
        mog2_background(s, nm, a.TB, b0, b1, b2);
        bgout.set(3 * j, b0), bgout.set(3 * j + 1, b1), bgout.set(3 * j + 2, b2);
      }
#pragma unroll
      for (int k = 0; k < kMog2K; ++k)
        st[k][j] = s.w[k], st[5 + k][j] = s.var[k], st[10 + 3 * k][j] = s.m0[k], st[11 + 3 * k][j] = s.m1[k], st[12 + 3 * k][j] = s.m2[k];
    }
    // a plane is written back only if some pixel of the wave changed it (wave-uniform, so every store stays a full 1 KiB row)
    const bool all = !a.sparse;
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) {
      const bool dw = (dirty_w >> k) & 1u, dm = (dirty_m >> k) & 1u;
      if (lanewise && k > lane_need) {  // slot not loaded by this lane: only a pixel that now owns a mode here may write, and only its own element
        if (dw || dm) {
#pragma unroll
          for (int j = 0; j < PX; ++j)
            if ((int)((nm_out >> (8 * j)) & 0xffu) > k) {
              a.state[mog2_plane_off<TILED>(a, k, sp) + j] = st[k][j];
              a.state[mog2_plane_off<TILED>(a, 5 + k, sp) + j] = st[5 + k][j];
#pragma unroll
              for (int c = 0; c < 3; ++c) a.state[mog2_plane_off<TILED>(a, 10 + 3 * k + c, sp) + j] = st[10 + 3 * k + c][j];
            }
        }
        continue;
      }
      if (all || (lanewise ? dw : (bool)__any(dw))) store_f<PX>(a.state + mog2_plane_off<TILED>(a, k, sp), st[k]);
      if (all || (lanewise ? dm : (bool)__any(dm))) {
        store_f<PX>(a.state + mog2_plane_off<TILED>(a, 5 + k, sp), st[5 + k]);
#pragma unroll
        for (int c = 0; c < 3; ++c) store_f<PX>(a.state + mog2_plane_off<TILED>(a, 10 + 3 * k + c, sp), st[10 + 3 * k + c]);
      }
    }
    const bool nm_dirty = all || (lanewise ? nm_out != nmw : (bool)__any(nm_out != nmw));
    if constexpr (PX == 4) {
      if (nm_dirty) *reinterpret_cast<uint32_t*>(nmp) = nm_out;
      if (a.fg) *reinterpret_cast<uint32_t*>(a.fg + p0) = mask_word;
    } else if constexpr (PX == 2) {
      if (nm_dirty) *reinterpret_cast<uint16_t*>(nmp) = (uint16_t)nm_out;
      if (a.fg) *reinterpret_cast<uint16_t*>(a.fg + p0) = (uint16_t)mask_word;
    } else {
      if (nm_dirty) *nmp = (uint8_t)nm_out;
      if (a.fg) a.fg[p0] = (uint8_t)mask_word;
    }
    if (a.want_bg) {
      if constexpr (PX == 4) {
        store_bytes<12>(a.bgimg + p0 * 3, bgout);
      } else {
#pragma unroll
        for (int i = 0; i < PX * 3; ++i) a.bgimg[p0 * 3 + i] = (uint8_t)bgout.get(i);
      }
    }
  }
  if (a.packed) {
    // every lane of the wave takes part (inactive tail lanes contribute 0); npix % 64 == 0 is checked by the host
    store_packed_mask<PX>(a.fg_bits, p0, bits, active);
  }
}

// ---- clip launches: T consecutive frames of every stream in ONE launch (bgs_process_clip_device) -------------------------
// The model of a pixel is loaded once, updated T times in registers in frame order with exactly the statements of the
// single-frame kernel, and written back once: model traffic per frame drops from 201 B/pixel to 201/T, the frame and mask bytes
// stay (4 B/pixel/frame).  Results are those of T successive single-frame launches, bit for bit: a pixel's update depends on
// nothing but its own model and its own input.  One pixel per lane; npix % 64 == 0 when masks are bit-packed (then a wave is
// active or idle as a whole, which the cross-lane packing needs).
// Slots at index >= nmodes are all-zero in memory ever since mog2_clear (nmodes never shrinks), so the slots a lane does not
// load are exactly the zeros it holds for them; a mode created in such a slot during the clip is written back like any other
// changed plane.
constexpr int kMog2ClipMax = 8;
struct Mog2ClipArgs {
  Mog2Args m;                  // frame / fg / bgimg / fg_bits point at the FIRST frame of the launch
  size_t frame_stride;         // bytes from one frame to the next (= pixels of the whole clip slab * 3), likewise below
  size_t fg_stride, bg_stride, bits_stride;  // bits_stride in 64-bit words
  float alphaT[kMog2ClipMax], alpha1[kMog2ClipMax], prune[kMog2ClipMax];  // per frame (the automatic rate changes with the frame count)
};

template <bool TILED, int T>
__global__ __launch_bounds__(kBlock) void mog2_clip_kernel(const Mog2ClipArgs c) {
  const Mog2Args& a = c.m;
  size_t blk = blockIdx.x;
  if (a.xcd_swizzle) {
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  const size_t p0 = blk * kBlock + threadIdx.x;
  if (p0 >= a.npix) return;  // wave-uniform whenever the packing below runs (npix % 64 == 0)
  const size_t sp = a.state_off + p0;
  uint32_t pix[T];  // all T inputs of this pixel are requested before the model: they are what the first update waits for
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint8_t* f = a.frame + (size_t)t * c.frame_stride + p0 * 3;
    pix[t] = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16);
  }
  uint8_t* const nmp = mog2_nmodes<TILED>(a, sp);
  const int nm_in = *nmp;
  Mog2Px s;
  auto load_mode = [&](int k) {
    s.w[k] = a.state[mog2_plane_off<TILED>(a, k, sp)];
    s.var[k] = a.state[mog2_plane_off<TILED>(a, 5 + k, sp)];
    s.m0[k] = a.state[mog2_plane_off<TILED>(a, 10 + 3 * k, sp)];
    s.m1[k] = a.state[mog2_plane_off<TILED>(a, 11 + 3 * k, sp)];
    s.m2[k] = a.state[mog2_plane_off<TILED>(a, 12 + 3 * k, sp)];
  };
  load_mode(0);
  // the same data-dependent loads as the single-frame kernel: sparse >= 2 loads modes below the wave's largest count + 1,
  // sparse >= 4 below the lane's own count
  int nload = kMog2K, lane_need = kMog2K;
  if (a.sparse >= 2) {
    if (a.sparse >= 4) lane_need = nm_in - 1;
    int M = 0;
#pragma unroll
    for (int n = 1; n <= kMog2K; ++n)
      if (__any(nm_in >= n)) M = n;
    nload = min(M + 1, kMog2K);
  }
  if (a.stat && (blockIdx.x & a.stat_mask) == 0) {
    const bool dense_wave = __any(nm_in >= kMog2K - 1);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      atomicAdd(a.stat, 1u);
      if (!dense_wave) atomicAdd(a.stat + 1, 1u);
    }
  }
#pragma unroll
  for (int k = 1; k < kMog2K; ++k) {
    if (k < nload && k <= lane_need)
      load_mode(k);
    else
      s.w[k] = 0.f, s.var[k] = 0.f, s.m0[k] = 0.f, s.m1[k] = 0.f, s.m2[k] = 0.f;
  }
  float worig[kMog2K];
#pragma unroll
  for (int k = 0; k < kMog2K; ++k) worig[k] = s.w[k];
  unsigned dirty_m = 0;
  int nm = nm_in;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x0 = (float)(pix[t] & 0xffu), x1 = (float)((pix[t] >> 8) & 0xffu), x2 = (float)(pix[t] >> 16);
    const int raw = mog2_pixel(s, nm, x0, x1, x2, a, dirty_m, c.alphaT[t], c.alpha1[t], c.prune[t]);
    const int m = thr_bin(raw, a.thr, a.enable_thr);
    if (a.fg) a.fg[(size_t)t * c.fg_stride + p0] = (uint8_t)m;
    if (a.packed) store_packed_mask<1>(a.fg_bits + (size_t)t * c.bits_stride, p0, (uint32_t)(m != 0), true);
    if (a.want_bg) {
      int b0, b1, b2;
      mog2_background(s, nm, a.TB, b0, b1, b2);
      uint8_t* o = a.bgimg + (size_t)t * c.bg_stride + p0 * 3;
      o[0] = (uint8_t)b0, o[1] = (uint8_t)b1, o[2] = (uint8_t)b2;
    }
  }
  const bool all = !a.sparse, lanewise = a.sparse >= 4;
#pragma unroll
  for (int k = 0; k < kMog2K; ++k) {
    const bool dw = s.w[k] != worig[k], dm = (dirty_m >> k) & 1u;
    const bool unloaded = lanewise && k > lane_need;  // holds zeros unless a mode was created here: only then there is something to write
    const bool sw = unloaded ? ((dw || dm) && nm > k) : (all || (lanewise ? dw : (bool)__any(dw)));
    const bool sm = unloaded ? sw : (all || (lanewise ? dm : (bool)__any(dm)));
    if (sw) a.state[mog2_plane_off<TILED>(a, k, sp)] = s.w[k];
    if (sm) {
      a.state[mog2_plane_off<TILED>(a, 5 + k, sp)] = s.var[k];
      a.state[mog2_plane_off<TILED>(a, 10 + 3 * k, sp)] = s.m0[k];
      a.state[mog2_plane_off<TILED>(a, 11 + 3 * k, sp)] = s.m1[k];
      a.state[mog2_plane_off<TILED>(a, 12 + 3 * k, sp)] = s.m2[k];
    }
  }
  if (all || (lanewise ? nm != nm_in : (bool)__any(nm != nm_in))) *nmp = (uint8_t)nm;
}

// (re)initialisation of a pixel range: bgmodel = zeros, modesUsed = 0 (BackgroundSubtractorMOG2::initialize)
template <bool TILED>
__global__ __launch_bounds__(kBlock) void mog2_clear_kernel(const Mog2Args a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npix) return;
  const size_t sp = a.state_off + p;
#pragma unroll
  for (int q = 0; q < kMog2Planes; ++q) a.state[mog2_plane_off<TILED>(a, q, sp)] = 0.f;
  *mog2_nmodes<TILED>(a, sp) = 0;
}

}  // namespace bgs
