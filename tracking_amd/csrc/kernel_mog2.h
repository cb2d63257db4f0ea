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
//   The arithmetic is the reference's statement for statement.  In registers only the weights and the slot ids are kept in rank order
//   and take part in the reference's bubble; a mode's record is fetched from where it was loaded when its turn in the scan comes
//   (a select chain), and the one record a frame recomputes or creates goes back to its slot with a single store.
// SUMMARIES - reading fewer records, exactly.  A record is only ever consulted by two comparisons, dist2 < Tb var (background) and
//   dist2 < Tg var (match).  Each slot also has a 2-byte SUMMARY (round 3: 4 bytes): the 8-level bucket each channel's mean lies in
//   (5 bits each) and a variance class, maintained under the invariants 8 q_c - 2 <= mean_c <= 8 q_c + 9 and class 0 => var <= 32.
//   From the summary and the pixel's own buckets alone a lower bound of dist2 follows (mog2_reject: one v_sad_u8); when it exceeds
//   max(Tb, Tg) 32 (+ margins for float rounding) both comparisons are PROVEN false: such a mode is REJECTED without its record, and the
//   frame proceeds exactly as the reference would (a rejected mode's record is used by nothing else: only its weight decays).  The
//   per-frame launch loads the summaries of a pixel's live slots, then only the records of the modes that survive: on well-separated
//   modes (the saturating benchmark input) one record of five.  A summary is rewritten when its record changes AND stops satisfying
//   the invariants (two levels of hysteresis: quiet pixels rewrite it rarely).  Only the filter path maintains summaries: bit 15 of the
//   meta word says "this pixel's summaries cover its records"; every other path clears it when it changes a record, and the filter
//   path, finding it clear, loads all the pixel's records, rebuilds the summaries and sets it.  Shadow detection and the background
//   image read every mode's mean, so launches that deliver them take the count path; clip launches the eager one (over 4-8 frames
//   nearly every mode is matched by some frame).
//   Reads 3 + 2 + 20 + 10 (summaries) + 16 n (n = surviving modes, 1 on the benchmark input), writes 20 + 16 + 2 + 1 (+ 2 when a summary
//   is rewritten): 90 B/pixel on the benchmark input.
// Tiles: pixels are grouped in tiles of kMog2Tile (256); a tile is 5 weight planes (T dwords each), 5 summary planes (T 16-bit words each),
//   5 record planes (T float4 each) and T meta words = 112 T contiguous bytes.  One workgroup owns one tile, one lane one pixel: every access
//   is a coalesced wave instruction (dwordx4 for the records), and with the XCD-aware block order each XCD streams one contiguous
//   eighth of the model.
// Stores are SECTOR-COMPLETE (args.complete): HBM moves 32-byte sectors, so a lane also writes back an unchanged value of its own
//   when another lane of the same sector (8 lanes of a weight or summary plane, 16 of the meta row, 2 of a record plane when every
//   lane holds all its records) has something to write there - no partially written sector reaches the memory controller.
// No LDS: the op is pointwise and HBM-bound.
#pragma once
#include "bgs_device.h"

namespace bgs {

constexpr int kMog2K = 5;
#ifndef BGS_MOG2_TILE
#define BGS_MOG2_TILE 256
#endif
constexpr int kMog2Tile = BGS_MOG2_TILE;                                          // pixels per tile
constexpr size_t kMog2TileBytes = (size_t)kMog2Tile * (4 * kMog2K + 2 * kMog2K + 16 * kMog2K + 2);  // 112 B per pixel: 28 672 B
constexpr size_t kMog2SumOff = (size_t)kMog2Tile * 4 * kMog2K;                    // byte offset of summary plane 0 inside a tile
constexpr size_t kMog2RecOff = kMog2SumOff + (size_t)kMog2Tile * 2 * kMog2K;      // ... of record plane 0 (16-byte aligned)
constexpr size_t kMog2MetaOff = kMog2RecOff + (size_t)kMog2Tile * 16 * kMog2K;   // ... of the meta row
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
  unsigned* stat;               // null, or 3 counters of the sampled workgroups: record slots (5 per pixel), modes the pixels have, records the filter path loads (auto mode)
  unsigned stat_mask;           // workgroups with (blockIdx.x & stat_mask) == 0 are sampled (~256 per launch)
  int sparse;                   // 0 dense: every weight, summary, record and meta word loaded and written back (placement probe, A/B);
                                // 1 eager: every record loaded (no dependent loads), only what changed written;
                                // 2 count: a lane loads only the modes its pixel has; 4 filter: their summaries, then only the records that survive them
  int complete;                 // sector-complete stores (see above)
  int xcd_swizzle;              // workgroups that share an XCD walk one contiguous eighth of the launch
};

struct Mog2Ptr {
  float* w;        // weight of rank r at w[r * kMog2Tile]
  uint16_t* sum;   // summary of slot s at sum[s * kMog2Tile]: q0 | q1 << 5 | q2 << 10 | class << 15 (see below)
  float4* rec;     // record of slot s at rec[s * kMog2Tile]: {variance, mean0, mean1, mean2}
  uint16_t* meta;
};
__device__ __forceinline__ Mog2Ptr mog2_ptr(uint8_t* state, size_t sp) {
  uint8_t* tb = state + (sp / kMog2Tile) * kMog2TileBytes;
  const size_t in = sp % kMog2Tile;
  Mog2Ptr p;
  p.w = reinterpret_cast<float*>(tb) + in;
  p.sum = reinterpret_cast<uint16_t*>(tb + kMog2SumOff) + in;
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

// ---- summaries (see the header comment) -------------------------------------------------------------------------------------
// Round 4: 16 bits per slot (round 3: 32).  q_c = floor(mean_c) >> 3, five bits per channel - the 8-level bucket the mean lies in -
// and a variance class in bit 15.  Invariants of a live slot's summary:
//   8 q_c - 2 <= mean_c <= 8 q_c + 9 for every channel (the bucket, two levels of hysteresis either side), and
//   class 0  =>  var <= kMog2SumVar (= 32); class 1 says nothing about the variance (such a mode is never ruled out).
// Ten bytes less to read per pixel and frame than the byte-exact means of round 3; what it costs is resolving power: modes closer than
// ~20 grey levels in every channel are not told apart any more (round 3: ~13) - more records read there, never a different result.
constexpr float kMog2SumVar = 32.f, kMog2SumVarTight = 24.f;
__device__ __host__ __forceinline__ uint32_t mog2_summary(float var, float m0, float m1, float m2) {
  const bool inside = m0 >= 0.f && m0 < 256.f && m1 >= 0.f && m1 < 256.f && m2 >= 0.f && m2 < 256.f;  // (false for NaN: class 1)
  const uint32_t cls = (inside && var <= kMog2SumVarTight) ? 0u : 1u;
  const uint32_t q0 = inside ? (uint32_t)(int)m0 >> 3 : 0u, q1 = inside ? (uint32_t)(int)m1 >> 3 : 0u, q2 = inside ? (uint32_t)(int)m2 >> 3 : 0u;
  return q0 | (q1 << 5) | (q2 << 10) | (cls << 15);
}
// may the stored summary be kept for this (changed) record?  the invariants, plus "not uselessly loose": a class-1 summary is rewritten
// as soon as the record would get class 0
__device__ __host__ __forceinline__ bool mog2_summary_ok(uint32_t word, float var, float m0, float m1, float m2) {
  const float b0 = (float)((word & 0x1fu) << 3), b1 = (float)(((word >> 5) & 0x1fu) << 3), b2 = (float)(((word >> 10) & 0x1fu) << 3);
  const float e0 = m0 - b0, e1 = m1 - b1, e2 = m2 - b2;
  const bool means = (e0 >= -2.f && e0 <= 9.f) && (e1 >= -2.f && e1 <= 9.f) && (e2 >= -2.f && e2 <= 9.f);
  const bool tightable = var <= kMog2SumVarTight && m0 >= 0.f && m0 < 256.f && m1 >= 0.f && m1 < 256.f && m2 >= 0.f && m2 < 256.f;
  return (word >> 15) ? !tightable : (means && var <= kMog2SumVar);
}
// does the summary PROVE dist2 >= Tmax * var, i.e. both of the reference's comparisons false for this pixel value?
// In integers, from ONE v_sad_u8 (round 4).  `pixq` = the pixel's channels >> 3 in bytes 0..2, `word` a class-0 summary, unpacked to
// the same byte positions: S = sum_c |xq_c - q_c|.  x_c lies in bucket xq_c, mean_c within two levels of bucket q_c, so
// |x_c - mean_c| >= 8 |xq_c - q_c| - 9, hence sum_c |x_c - mean_c| >= 8 S - 27, and by Cauchy-Schwarz
// dist2 = sum_c (x_c - mean_c)^2 >= (8 S - 27)^2 / 3.  B = 8 S - 28 > 0 (one more for every float rounding on either side: the values
// are <= 2e5, relative error ~1e-7) and B^2 > 3 (Tmax kMog2SumVar + 1) with var <= kMog2SumVar proves dist2 > Tmax var + 1.
// tq = ceil(3 Tmax kMog2SumVar) + 3, clamped (a threshold that large rejects nothing).  The price of integers and buckets is a
// weaker bound when the modes differ in ONE channel only (factor 3) or by less than ~20 levels: more records read there.
__device__ __host__ __forceinline__ uint32_t mog2_tq(float Tmax) {
  const float c = 3.f * kMog2SumVar * Tmax;
  return c < 1.0e9f ? (uint32_t)(int)c + 4u : 0xffffffffu;  // > 3 (Tmax var + 1) for every var <= kMog2SumVar (NaN / negative thresholds: the clamp or 4, both safe)
}
__device__ __forceinline__ uint32_t mog2_pixq(uint32_t pix) { return (pix >> 3) & 0x1f1f1fu; }
__device__ __forceinline__ bool mog2_reject(uint32_t word, uint32_t pixq, uint32_t tq) {
  const uint32_t qb = (word & 0x1fu) | ((word & 0x3e0u) << 3) | ((word & 0x7c00u) << 6);
  const uint32_t S = __builtin_amdgcn_sad_u8(pixq, qb, 0u);
  const uint32_t B = 8u * S - 28u;  // (wraps for S < 4: caught by the first test)
  return S > 3u && !(word >> 15) && B * B > tq;
}

// One pixel's model in registers.  Weights and slot ids in rank order, as MOG2Invoker sees its array.  The records {var, mean0,
// mean1, mean2} in one of three orders (template parameter ORD of what follows):
//   kMog2Compact  rc[j] is the record of slot kj[j] for j < cnt, wherever the loads put it (filter kernel);
//   kMog2BySlot   rc[j] is the record of slot j (eager and count per-frame kernels): a mode's record is fetched when its turn in
//                 the scan comes (a select chain), the records themselves never move;
//   kMog2Ranked   rc[r] is the record of the mode at rank r and takes part in the bubble like the reference's array (clip kernels:
//                 gathered once per launch, then T frames index it statically).
enum { kMog2Compact = 0, kMog2BySlot = 1, kMog2Ranked = 2 };
struct Mog2Px {
  float w[kMog2K];
  int sl[kMog2K];  // slot + 1 of the mode at rank r (0: rank unused)
};
struct Mog2Recs {
  float4 rc[kMog2K];
  int kj[kMog2K], cnt;
};

template <int ORD>
__device__ __forceinline__ void mog2_swap(Mog2Px& s, Mog2Recs& R, int i, int j) {
  const float t = s.w[i];
  s.w[i] = s.w[j], s.w[j] = t;
  const int u = s.sl[i];
  s.sl[i] = s.sl[j], s.sl[j] = u;
  if constexpr (ORD == kMog2Ranked) {
    const float4 c = R.rc[i];
    R.rc[i] = R.rc[j], R.rc[j] = c;
  }
}

// the record of the mode with slot code `code` (slot + 1).  BYSLOT: rc[j] is slot j (a compare against constants)
template <bool BYSLOT>
__device__ __forceinline__ float4 mog2_pick(const Mog2Recs& R, int code) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < kMog2K; ++j) {
    const bool here = BYSLOT ? (code == j + 1) : (j < R.cnt && code == R.kj[j] + 1);
    const float4 c = R.rc[j];  // selects of VALUES, component by component: `if (here) v = R.rc[j]` became a select of addresses + a load, and the whole struct went to scratch
    v.x = here ? c.x : v.x, v.y = here ? c.y : v.y, v.z = here ? c.z : v.z, v.w = here ? c.w : v.w;
  }
  return v;
}
// the record of the mode at rank `mode` before this frame's update
template <int ORD>
__device__ __forceinline__ float4 mog2_rec_at(const Mog2Px& s, const Mog2Recs& R, int mode) {
  if constexpr (ORD == kMog2Ranked)
    return R.rc[mode];
  else
    return mog2_pick<ORD == kMog2BySlot>(R, s.sl[mode]);
}
// ... as it stands after this frame's update: the recomputed / created record is `out`, of slot code `hit`
template <bool BYSLOT>
__device__ __forceinline__ float4 mog2_pick_now(const Mog2Recs& R, int code, int hit, float4 out) {
  const float4 v = mog2_pick<BYSLOT>(R, code);
  const bool h = code == hit;
  return make_float4(h ? out.x : v.x, h ? out.y : v.y, h ? out.z : v.z, h ? out.w : v.w);
}

template <int ORD>
__device__ __forceinline__ float4 mog2_rec_now(const Mog2Px& s, const Mog2Recs& R, int mode, int hit, float4 out) {
  if constexpr (ORD == kMog2Ranked)
    return R.rc[mode];  // kept current
  else
    return mog2_pick_now<ORD == kMog2BySlot>(R, s.sl[mode], hit, out);
}

// detectShadowGMM of bgfg_gaussmix2.cpp, predicated form of its early returns
template <int ORD>
__device__ __forceinline__ bool mog2_shadow(const Mog2Px& s, const Mog2Recs& R, int hit, float4 out, int nmodes, float x0, float x1, float x2, const Mog2Args& a) {
  bool done = false, result = false;
  float tWeight = 0.f;
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode) {
    if (mode < nmodes && !done) {
      const float4 v = mog2_rec_now<ORD>(s, R, mode, hit, out);
      float num = 0.0f, den = 0.0f;
      num += x0 * v.y;
      den += v.y * v.y;
      num += x1 * v.z;
      den += v.z * v.z;
      num += x2 * v.w;
      den += v.w * v.w;
      if (den == 0) {
        done = true;
      } else {
        if (num <= den && num >= a.tau * den) {
          const float q = div_rn(num, den);
          float d2a = 0.0f, dD;
          dD = q * v.y - x0, d2a += dD * dD;
          dD = q * v.z - x1, d2a += dD * dD;
          dD = q * v.w - x2, d2a += dD * dD;
          if (d2a < a.Tb * v.x * q * q) result = true, done = true;
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
// Every frame recomputes (the matched mode) or creates exactly one record: `out`, of slot code `hit` (slot + 1).
// `rej`: bit (slot + 1) set = this mode's summary proved that neither comparison can hold for this pixel value (mog2_reject): the
// comparisons are skipped - their outcome is known - and the record, which is not even loaded, is not touched.
template <int ORD>
__device__ __forceinline__ int mog2_pixel(Mog2Px& s, Mog2Recs& R, int& nmodes_io, float x0, float x1, float x2, const Mog2Args& a, int& hit, float4& out,
                                          const float alphaT, const float alpha1, const float prune, const unsigned rej) {
  bool background = false, fitsPDF = false;
  int nmodes = nmodes_io;
  const int nNewModes = nmodes;
  float totalWeight = 0.f;
  hit = 0;
  out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode) {
    if (mode < nmodes) {  // nmodes shrinks inside the loop when a mode is pruned (reference quirk)
      float weight = alpha1 * s.w[mode] + prune;
      bool matched = false;
      if (!fitsPDF && !((rej >> s.sl[mode]) & 1u)) {
        const float4 v = mog2_rec_at<ORD>(s, R, mode);
        const float var = v.x;
        const float d0 = v.y - x0, d1 = v.z - x1, d2 = v.w - x2;
        const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
        if (totalWeight < a.TB && dist2 < a.Tb * var) background = true;
        if (dist2 < a.Tg * var) {
          fitsPDF = true;
          matched = true;
          weight += alphaT;
          const float k = div_rn(alphaT, weight);
          float varnew = var + k * (dist2 - var);
          varnew = varnew > a.varMin ? varnew : a.varMin;
          varnew = varnew < a.varMax ? varnew : a.varMax;
          out = make_float4(varnew, v.y - k * d0, v.z - k * d1, v.w - k * d2);
          hit = s.sl[mode];
          if constexpr (ORD == kMog2Ranked) R.rc[mode] = out;
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
          if (moving) mog2_swap<ORD>(s, R, i, i - 1);
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
        s.sl[k] = slot;
        if constexpr (ORD == kMog2Ranked) R.rc[k] = make_float4(a.varInit, x0, x1, x2);
      } else if (nmodes != 1 && k < nmodes - 1) {
        s.w[k] *= alpha1;
      }
    }
    out = make_float4(a.varInit, x0, x1, x2);
    hit = slot;
    bool moving = true;
#pragma unroll
    for (int i = kMog2K - 1; i > 0; --i) {
      if (i <= nmodes - 1) {
        moving = moving && !(alphaT < s.w[i - 1]);
        if (moving) mog2_swap<ORD>(s, R, i, i - 1);
      }
    }
  }
  nmodes_io = nmodes;
  if (background) return 0;
  if (a.shadow) {
    if (mog2_shadow<ORD>(s, R, hit, out, nmodes, x0, x1, x2, a)) return a.shadow_val;
  }
  return 255;
}

// The same pixel update in LOCK-STEP (round 4), for the per-frame kernels (records by slot or compacted; the clip kernels keep the
// form above, whose records travel with the bubble).  mog2_pixel is exact but DIVERGENT: the mode a pixel matches sits at a different
// rank in every lane, so each of the five unrolled rank iterations runs its record fetch (a select chain), distance, division and
// bubble for SOME lanes of the wave - up to five times the work of one pixel on a scene whose pixels are out of step (the benchmark
// input S_sat by construction; PMC: 620 vector instructions per pixel and frame, 79 % of the SIMDs' issue slots over the launch).
// Here everything that touches a record happens outside the rank loop:
//   1. every record the lane holds (`ncand` of them: all live slots, or what the summaries left) gives dist2 and the two comparisons
//      dist2 < Tb var, dist2 < Tg var ONCE - they depend on neither the rank nor the running weight sum - kept as bit masks by slot;
//   2. the rank loop only decays, prunes and sums the weights and looks the two bits of its rank's slot up: the background test reads
//      totalWeight as it stands when the scan reaches that rank, the first mode in rank order whose match bit is set takes alphaT;
//   3. the matched record is fetched and recomputed once, then the bubble runs from its rank.
// Same statements on the same operands for every float, in the same order wherever order matters (the weight sum).  A record behind
// the matched one in rank order is compared here although the reference never looks at it: a comparison has no side effect.
// `wave_cand`: wave-uniform upper bound of ncand (the candidate loop's trip count).
template <int ORD>
__device__ __forceinline__ int mog2_pixel_lockstep(Mog2Px& s, const Mog2Recs& R, const int ncand, const int wave_cand, int& nmodes_io, float x0, float x1, float x2,
                                                   const Mog2Args& a, int& hit, float4& out, const float alphaT, const float alpha1, const float prune) {
  static_assert(ORD == kMog2Compact || ORD == kMog2BySlot, "records by slot or compacted");
  bool background = false, fitsPDF = false;
  int nmodes = nmodes_io;
  const int nNewModes = nmodes;
  float totalWeight = 0.f;
  hit = 0;
  out = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned cm = 0, bgm = 0, mm = 0;  // bit (slot + 1): a candidate / dist2 < Tb var / dist2 < Tg var
#pragma unroll
  for (int j = 0; j < kMog2K; ++j) {
    if (j < wave_cand) {  // wave-uniform
      const float4 v = R.rc[j];
      const float d0 = v.y - x0, d1 = v.z - x1, d2 = v.w - x2;
      const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
      const int code = ORD == kMog2BySlot ? j + 1 : R.kj[j] + 1;
      const bool is = j < ncand;
      cm |= (unsigned)is << code;
      bgm |= (unsigned)(is && dist2 < a.Tb * v.x) << code;
      mm |= (unsigned)(is && dist2 < a.Tg * v.x) << code;
    }
  }
  float wmatch = 0.f;
  int rstar = 0;
#pragma unroll
  for (int mode = 0; mode < kMog2K; ++mode) {
    if (mode < nmodes) {  // nmodes shrinks inside the loop when a mode is pruned (reference quirk)
      float weight = alpha1 * s.w[mode] + prune;
      const int code = s.sl[mode];
      if (!fitsPDF && ((cm >> code) & 1u)) {
        if (totalWeight < a.TB && ((bgm >> code) & 1u)) background = true;
        if ((mm >> code) & 1u) {
          fitsPDF = true;
          weight += alphaT;
          wmatch = weight, rstar = mode, hit = code;
        }
      }
      const bool pruned = weight < -prune;
      if (pruned) nmodes--;
      s.w[mode] = pruned ? 0.f : weight;
      totalWeight += pruned ? 0.f : weight;
    }
  }
  if (fitsPDF) {
    const float4 v = wave_cand == 1 ? R.rc[0] : mog2_pick<ORD == kMog2BySlot>(R, hit);  // (wave-uniform choice; one candidate: it is the match)
    const float var = v.x;
    const float d0 = v.y - x0, d1 = v.z - x1, d2 = v.w - x2;
    const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
    const float k = div_rn(alphaT, wmatch);
    float varnew = var + k * (dist2 - var);
    varnew = varnew > a.varMin ? varnew : a.varMin;
    varnew = varnew < a.varMax ? varnew : a.varMax;
    out = make_float4(varnew, v.y - k * d0, v.z - k * d1, v.w - k * d2);
    // the reference's bubble, run at the matched rank: it compares the UNPRUNED matched weight with the stored weights in front
    bool moving = true;
    Mog2Recs none;  // (the records do not take part in the swaps)
#pragma unroll
    for (int i = kMog2K - 1; i > 0; --i) {
      if (i <= rstar) {
        moving = moving && !(wmatch < s.w[i - 1]);
        if (moving) mog2_swap<ORD>(s, none, i, i - 1);
      }
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
        s.sl[k] = slot;
      } else if (nmodes != 1 && k < nmodes - 1) {
        s.w[k] *= alpha1;
      }
    }
    out = make_float4(a.varInit, x0, x1, x2);
    hit = slot;
    bool moving = true;
    Mog2Recs none;
#pragma unroll
    for (int i = kMog2K - 1; i > 0; --i) {
      if (i <= nmodes - 1) {
        moving = moving && !(alphaT < s.w[i - 1]);
        if (moving) mog2_swap<ORD>(s, none, i, i - 1);
      }
    }
  }
  nmodes_io = nmodes;
  if (background) return 0;
  if (a.shadow) {
    if (mog2_shadow<ORD>(s, R, hit, out, nmodes, x0, x1, x2, a)) return a.shadow_val;
  }
  return 255;
}

// cv::BackgroundSubtractorMOG2::getBackgroundImage, one pixel, from the registers that already hold the model
template <int ORD>
__device__ __forceinline__ void mog2_background(const Mog2Px& s, const Mog2Recs& R, int hit, float4 out, int nmodes, float TB, int& b0, int& b1, int& b2) {
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, totalWeight = 0.f;
  bool stop = false;
#pragma unroll
  for (int g = 0; g < kMog2K; ++g) {
    if (g < nmodes && !stop) {
      const float4 v = mog2_rec_now<ORD>(s, R, g, hit, out);
      const float w = s.w[g];
      v0 += w * v.y;
      v1 += w * v.z;
      v2 += w * v.w;
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

// How a launch loads a pixel's model (one kernel instance each; results are identical, only the traffic differs):
//   EAGER   all weights and records at once, nothing waits for the meta word - right when most modes are needed anyway; every clip
//           launch.  With args.sparse == 0 also "dense": everything is written back too (the placement probe's traffic, A/B runs);
//   COUNT   rank 0 / slot 0 at once, the other modes' weights and records only for the modes the pixel has: one dependent round of
//           loads, far fewer bytes on quiet scenes (one or two modes per pixel);
//   FILTER  all weights and summaries at once, then only the records the summaries cannot rule out: one dependent round; pays when
//           a pixel's modes lie far apart (the saturating benchmark input: 1 record of 5).  Its sampled workgroups also count what
//           each way would load, for the engine's automatic choice.
enum { kMog2Eager = 1, kMog2Count = 2, kMog2Filter = 4 };

template <int MODE, int T>
__device__ __forceinline__ void mog2_body(const Mog2Args& a, const size_t frame_stride, const size_t fg_stride, const size_t bg_stride, const size_t bits_stride,
                                          const float* alphaT, const float* alpha1, const float* prune) {
  static_assert(T == 1 || MODE == kMog2Eager, "clip launches load eagerly");
  constexpr int ORD = MODE == kMog2Filter ? kMog2Compact : T > 1 ? kMog2Ranked : kMog2BySlot;
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
  const unsigned meta_raw = *mp.meta;
  const unsigned meta_in = meta_raw & 0x7fffu;
  const bool valid_in = (meta_raw >> 15) & 1u;
  const int nm_in = mog2_meta_count(meta_in);
  const bool dense = MODE == kMog2Eager && a.sparse == 0;
  const bool can_reject = !a.shadow && !a.want_bg;  // the shadow test and the background image read every mode's mean
  float wv[kMog2K];
  uint32_t sm[kMog2K];
  Mog2Recs R;
  R.cnt = kMog2K;
  unsigned rej = 0;  // bit (slot + 1): rejected by its summary for THIS frame (filter path)
  int wave_cand = kMog2K;  // wave-uniform bound of the records a lane holds (R.cnt): the candidate loop of mog2_pixel_lockstep
#pragma unroll
  for (int k = 0; k < kMog2K; ++k) R.kj[k] = k, sm[k] = 0u;
  // Every load below is UNCONDITIONAL per lane and sits in straight-line code: a lane that does not want plane k repeats a load it
  // does want (same cache line, no extra HBM traffic) and the value is discarded by a select afterwards.  Loads under a per-lane
  // `if` - and also loads under a chain of wave-uniform branches - made the compiler merge each result with the "not loaded" value
  // through register copies right behind the load, with s_waitcnt vmcnt(0) in front of them: the loads then went out one memory
  // round trip after the other (seen in the ISA of the first two versions of this path, and in round 2 in SuBSENSE's sample
  // prefetch).  So there is ONE wave-uniform two-way choice per round of loads: the short form when no lane of the wave needs
  // more, else all five.
  if constexpr (MODE == kMog2Eager) {
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) {
      wv[k] = mp.w[(size_t)k * kMog2Tile];
      R.rc[k] = mp.rec[(size_t)k * kMog2Tile];
      if (dense) sm[k] = mp.sum[(size_t)k * kMog2Tile];
    }
  } else if constexpr (MODE == kMog2Count) {
    wv[0] = mp.w[0];
    R.rc[0] = mp.rec[0];  // slot 0 exists from a pixel's first frame on
    int wave_nm = 0;
#pragma unroll
    for (int n = 1; n <= kMog2K; ++n)
      if (__any(nm_in >= n)) wave_nm = n;
    wave_cand = wave_nm <= 2 ? 2 : wave_nm;
    if (wave_nm <= 2) {
      const size_t idx = (size_t)(1 < nm_in ? 1 : 0) * kMog2Tile;
      wv[1] = mp.w[idx], R.rc[1] = mp.rec[idx];
#pragma unroll
      for (int k = 2; k < kMog2K; ++k) wv[k] = 0.f, R.rc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int k = 1; k < kMog2K; ++k) {
        const size_t idx = (size_t)(k < nm_in ? k : 0) * kMog2Tile;
        wv[k] = mp.w[idx], R.rc[k] = mp.rec[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) wv[k] = k < nm_in ? wv[k] : 0.f;
    R.cnt = nm_in;
  } else {
    // all five weights and summaries at once, whatever the pixel has: this path is chosen for scenes where pixels have several modes,
    // and not waiting for the meta word saves a whole round trip
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) wv[k] = mp.w[(size_t)k * kMog2Tile], sm[k] = mp.sum[(size_t)k * kMog2Tile];
    const uint32_t tq = mog2_tq(a.Tb > a.Tg ? a.Tb : a.Tg), pixq = mog2_pixq(pix[0]);
    unsigned need = 0;  // bit k: the record of slot k must be read
#pragma unroll
    for (int k = 0; k < kMog2K; ++k) {
      const bool live = k < nm_in;
      wv[k] = live ? wv[k] : 0.f;
      sm[k] = live ? sm[k] : 0u;
      const bool r = live && valid_in && can_reject && mog2_reject(sm[k], pixq, tq);
      rej |= (unsigned)r << (k + 1);
      need |= (unsigned)(live && !r) << k;
    }
    // the needed slots, compacted: the j-th load of a lane fetches its j-th needed record (a lane with fewer repeats its first)
    R.cnt = __popc(need);
    const int k_first = need ? __ffs(need) - 1 : 0;
    R.kj[0] = k_first;
    if (!__any(R.cnt > 1)) {
      wave_cand = 1;
      R.rc[0] = mp.rec[(size_t)k_first * kMog2Tile];
#pragma unroll
      for (int j = 1; j < kMog2K; ++j) R.rc[j] = make_float4(0.f, 0.f, 0.f, 0.f), R.kj[j] = k_first;
    } else {
      unsigned left = need & (need - 1);
#pragma unroll
      for (int j = 1; j < kMog2K; ++j) {
        R.kj[j] = left ? __ffs(left) - 1 : k_first;
        left &= left - 1;
      }
#pragma unroll
      for (int j = 0; j < kMog2K; ++j) R.rc[j] = mp.rec[(size_t)R.kj[j] * kMog2Tile];
    }
    if (a.stat && (blockIdx.x & a.stat_mask) == 0) {
      // per pixel: 5 record slots, the modes it has (what COUNT loads), the records the summaries leave (what FILTER loads);
      // sums over the wave from the ballots of the three bits of each count
      // (a pixel whose summaries were not valid had all its records loaded: what fresh summaries would have left is counted instead,
      // so that a single filter launch in the middle of another kernel's run can tell whether filtering would pay)
      int would = R.cnt;
      if (!valid_in && can_reject) {
        would = 0;
#pragma unroll
        for (int j = 0; j < kMog2K; ++j) {
          const float4 c = R.rc[j];
          if (j < R.cnt && !mog2_reject(mog2_summary(c.x, c.y, c.z, c.w), pixq, tq)) ++would;
        }
      }
      unsigned s_live = 0, s_need = 0;
#pragma unroll
      for (int bit = 0; bit < 3; ++bit) {
        s_live += (unsigned)__popcll(__ballot((nm_in >> bit) & 1)) << bit;
        s_need += (unsigned)__popcll(__ballot((would >> bit) & 1)) << bit;
      }
      const unsigned lanes = (unsigned)__popcll(__ballot(1));
      if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(a.stat, lanes * kMog2K);
        atomicAdd(a.stat + 1, s_live);
        atomicAdd(a.stat + 2, s_need);
      }
    }
  }
  Mog2Px s;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) s.sl[r] = (int)((meta_in >> (3 * r)) & 7u), s.w[r] = wv[r];
  if constexpr (ORD == kMog2Ranked) {  // clip launches: the records into rank order, once
    Mog2Recs Q;
#pragma unroll
    for (int r = 0; r < kMog2K; ++r) Q.rc[r] = mog2_pick<true>(R, s.sl[r]);
#pragma unroll
    for (int r = 0; r < kMog2K; ++r) R.rc[r] = Q.rc[r];
  }
  unsigned dirty = 0;  // bit (slot + 1): the record changed
  int nm = nm_in, hit = 0;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float x0 = (float)(pix[t] & 0xffu), x1 = (float)((pix[t] >> 8) & 0xffu), x2 = (float)(pix[t] >> 16);
    int raw;
    if constexpr (ORD == kMog2Ranked)
      raw = mog2_pixel<ORD>(s, R, nm, x0, x1, x2, a, hit, out, alphaT[t], alpha1[t], prune[t], rej);
    else
      raw = mog2_pixel_lockstep<ORD>(s, R, R.cnt, wave_cand, nm, x0, x1, x2, a, hit, out, alphaT[t], alpha1[t], prune[t]);
    const int m = thr_bin(raw, a.thr, a.enable_thr);
    if (a.fg) a.fg[(size_t)t * fg_stride + p0] = (uint8_t)m;
    if (a.packed) store_packed_mask<1>(a.fg_bits + (size_t)t * bits_stride, p0, (uint32_t)(m != 0), true);
    if (a.want_bg) {
      int b0, b1, b2;
      mog2_background<ORD>(s, R, hit, out, nm, a.TB, b0, b1, b2);
      uint8_t* o = a.bgimg + (size_t)t * bg_stride + p0 * 3;
      o[0] = (uint8_t)b0, o[1] = (uint8_t)b1, o[2] = (uint8_t)b2;
    }
    dirty |= 1u << hit;
  }
  // what changed: bits 0..4 weight of rank r, bit 5 the meta word, bits 6..10 the record of slot s, bits 11..15 its summary
  unsigned d = 0;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) d |= (unsigned)(s.w[r] != wv[r]) << r;
  d |= (dirty >> 1) << 6;
  bool valid_out = false;  // every launch changes a record of every pixel; only the filter path looks after the summaries
  if constexpr (MODE == kMog2Filter) {
    // Summaries.  The common case costs a select and a few compares: the one record that changed is `out`; its stored summary is
    // kept while it still covers the record (hysteresis: quiet pixels rewrite theirs rarely).  Rewrites and rebuilds (a pixel whose
    // summaries were not valid on entry had all its records loaded for this) take the slot loop - skipped by the whole wave otherwise.
    uint32_t old = 0;
#pragma unroll
    for (int q = 0; q < kMog2K; ++q) old = hit == q + 1 ? sm[q] : old;
    const bool keep = valid_in && hit <= nm_in && mog2_summary_ok(old, out.x, out.y, out.z, out.w);  // (a slot created by this launch has no summary yet)
    if (__any(!keep)) {
#pragma unroll
      for (int q = 0; q < kMog2K; ++q) {
        const bool changed = hit == q + 1;
        if ((changed && !keep) || (!valid_in && q < nm)) {
          const float4 v = mog2_pick_now<false>(R, q + 1, hit, out);
          sm[q] = mog2_summary(v.x, v.y, v.z, v.w);
          d |= 1u << (11 + q);
        }
      }
    }
    valid_out = true;
  }
  unsigned meta_out = valid_out ? 0x8000u : 0u;
#pragma unroll
  for (int r = 0; r < kMog2K; ++r) meta_out |= (unsigned)s.sl[r] << (3 * r);
  d |= (unsigned)(meta_out != meta_raw) << 5;
  if (dense) {
    d = 0xffffu;
  } else if (a.complete) {
    // whole 32-byte sectors or nothing: 8 lanes share a sector of a weight or summary plane, 16 one of the meta row, 2 one of a
    // record plane.  A lane can only complete a sector with a value it holds: summaries on the filter path, records on the eager path.
    unsigned d2, d8, d16;
    mog2_group_or(d, d2, d8, d16);
    unsigned c = (d8 & 0x1fu) | (d16 & 0x20u);
    if constexpr (MODE == kMog2Filter) c |= d16 & 0xf800u;  // it holds every summary of the pixel (live ones loaded or rebuilt, the rest unused); 16 lanes x 2 bytes share a sector
    if constexpr (MODE == kMog2Eager) c |= d2 & 0x7c0u;    // it holds every record
    d |= c;
  }
#pragma unroll
  for (int r = 0; r < kMog2K; ++r)
    if ((d >> r) & 1u) mp.w[(size_t)r * kMog2Tile] = s.w[r];
  if ((d >> 5) & 1u) *mp.meta = (uint16_t)meta_out;
  if constexpr (MODE == kMog2Eager) {
    // slot q as it stands: per-frame launch = as loaded, with this frame's record patched in; clip = wherever the bubble left it
    // (a slot the pixel does not own holds nothing anyone reads: zeros)
#pragma unroll
    for (int q = 0; q < kMog2K; ++q) {
      if ((d >> (11 + q)) & 1u) mp.sum[(size_t)q * kMog2Tile] = (uint16_t)sm[q];  // dense only
      if ((d >> (6 + q)) & 1u) {
        float4 v;
        if constexpr (T == 1) {
          const bool h = hit == q + 1;
          v = make_float4(h ? out.x : R.rc[q].x, h ? out.y : R.rc[q].y, h ? out.z : R.rc[q].z, h ? out.w : R.rc[q].w);
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int r = 0; r < kMog2K; ++r) {
            const bool h = s.sl[r] == q + 1;
            v.x = h ? R.rc[r].x : v.x, v.y = h ? R.rc[r].y : v.y, v.z = h ? R.rc[r].z : v.z, v.w = h ? R.rc[r].w : v.w;
          }
        }
        mp.rec[(size_t)q * kMog2Tile] = v;
      }
    }
  } else {
    // exactly one record per pixel and frame, at the slot it belongs to: one store with a per-lane address
    mp.rec[(size_t)(hit - 1) * kMog2Tile] = out;
    if constexpr (MODE == kMog2Filter) {
#pragma unroll
      for (int q = 0; q < kMog2K; ++q)
        if ((d >> (11 + q)) & 1u) mp.sum[(size_t)q * kMog2Tile] = (uint16_t)sm[q];
    }
  }
}

// grid: ceil(npix / kBlock) blocks of kBlock lanes, one pixel per lane
template <int MODE>
__global__ __launch_bounds__(kBlock) void mog2_update_kernel(const Mog2Args a) {
  mog2_body<MODE, 1>(a, 0, 0, 0, 0, &a.alphaT, &a.alpha1, &a.prune);
}

template <int T>
__global__ __launch_bounds__(kBlock) void mog2_clip_kernel(const Mog2ClipArgs c) {
  mog2_body<kMog2Eager, T>(c.m, c.frame_stride, c.fg_stride, c.bg_stride, c.bits_stride, c.alphaT, c.alpha1, c.prune);
}

// (re)initialisation of a pixel range: bgmodel = zeros, modesUsed = 0 (BackgroundSubtractorMOG2::initialize)
__global__ __launch_bounds__(kBlock) void mog2_clear_kernel(const Mog2Args a) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npix) return;
  const Mog2Ptr mp = mog2_ptr(a.state, a.state_off + p);
#pragma unroll
  for (int k = 0; k < kMog2K; ++k) {
    mp.w[(size_t)k * kMog2Tile] = 0.f;
    mp.sum[(size_t)k * kMog2Tile] = 0;
    mp.rec[(size_t)k * kMog2Tile] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  *mp.meta = 0;
}

}  // namespace bgs
