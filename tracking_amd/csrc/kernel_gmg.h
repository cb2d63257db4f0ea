// kernel_gmg.h — K6: GMG (cv::BackgroundSubtractorGMG per-pixel histogram update + Bayesian decision).
//
// Replaces  GMG::process  package_bgs/GMG.cpp:56  ((*fgbg)(img_input, img_foreground) with initializationFrames = 20,
// decisionThreshold = 0.7); algorithm: OpenCV 2.4 bgfg_gmg.cpp GMG_LoopBody (SURVEY.md App. B.4 — the least certain recall of
// the whole path; the assumptions G1..G7 are listed in DESIGN.md §5.2).  The median smoothing that follows is morph_box_kernel.
//
// Layout: planes of 8-byte RECORDS  rec {colour int32, weight f32} [F][P], nfeatures u8 [P]  (F = maxFeatures <= 64); entry f of every
// pixel lives in plane f, so each step of the per-pixel list walk is one coalesced 512-byte access per wave (round 4; rounds 1-3 kept
// colours and weights in separate planes: 19 load and 12 store instructions per pixel on a settled scene, now 11 and 12; a weight that
// changes while its colour stays goes back as the whole record - a 4-byte store into it leaves every other dword of the line untouched,
// i.e. partial sectors, and measured 4 % slower).  One lane owns one pixel; all list walks run to the largest
// count in the wave (wave-uniform trip counts via __any).
// Traffic is data-dependent: ~ (8 B read + 8 B written) x features of the pixel, up to 1 KiB/pixel; HBM-bound.
#pragma once
#include "bgs_device.h"

namespace bgs {

struct GmgArgs {
  const uint8_t* frame;  // [npix][C]
  uint8_t* raw;          // [npix] unsmoothed mask
  int2* rec;             // [F][plane] {colour, weight (float bits)}
  uint8_t* nfeat;        // [plane]
  size_t plane, state_off, npix;
  int F, C, levels, typical, update, normalize_now, fullStore;  // typical = frameNum >= init; normalize_now = frameNum == init - 1
  double lr, prior, thr;
};

// Short lists (the usual case: a static pixel sees a handful of quantised colours) are handled in registers: when no lane
// of the wave has more than kGmgFast - 1 features, the first kGmgFast entries are fetched with independent loads, the list
// operations below run on registers with static indices, and the entries are written back once.  Same arithmetic in the
// same order as the general path (gmg_pixel_lists), which stays for long lists.
constexpr int kGmgFast = 8;

__device__ __forceinline__ void gmg_fast(const GmgArgs& a, bool active, size_t sp, size_t p0, int nf, int color) {
  int c[kGmgFast];
  float w[kGmgFast];
  // All 16 loads are issued back to back and waited for once.  They are UNCONDITIONAL on purpose: written as `if (i < nf) load`
  // the compiler merged every loaded value with the default through a copy and put s_waitcnt vmcnt(0) behind each pair - eight
  // serial memory round trips (round 2, found in the ISA like SuBSENSE's prefetch).  Entries past the count re-read the last
  // valid one (same cache line, no extra traffic) and are zeroed afterwards.
  const int lastValid = max(nf - 1, 0);
#pragma unroll
  for (int i = 0; i < kGmgFast; ++i) {
    const int2 r = a.rec[(size_t)min(i, lastValid) * a.plane + sp];
    c[i] = r.x, w[i] = __int_as_float(r.y);
  }
#pragma unroll
  for (int i = 0; i < kGmgFast; ++i)
    if (!(active && i < nf)) c[i] = 0, w[i] = 0.f;
  int c_in[kGmgFast];  // as loaded: a colour is stored again only if it changed (round 3: on a settled pixel none does - the
  const int nf_in = nf;  // matched colour sits at the front - and rewriting them all was a third of the kernel's write traffic)
#pragma unroll
  for (int i = 0; i < kGmgFast; ++i) c_in[i] = c[i];
  int idx = -1;
  float wfound = 0.f;
#pragma unroll
  for (int i = 0; i < kGmgFast; ++i)
    if (i < nf && idx < 0 && c[i] == color) idx = i, wfound = w[i];
  bool isfg = false;
  if (a.typical) {
    const double wd = (double)wfound;
    const double num = __dmul_rn(wd, a.prior);
    const double den = __dadd_rn(num, __dmul_rn(__dsub_rn(1.0, wd), __dsub_rn(1.0, a.prior)));
    isfg = __dsub_rn(1.0, __ddiv_rn(num, den)) > a.thr;
  }
  if (a.update && active) {
    const double decay = a.typical ? __dsub_rn(1.0, a.lr) : 1.0;
    const float ins = a.typical ? (float)a.lr : 1.0f;
    const bool found = idx >= 0, full = !found && nf == a.F;
    const float front_w = found ? ins + (a.typical ? (float)__dmul_rn((double)wfound, decay) : wfound) : ins;
    const int shift_end = found ? idx : (full ? nf - 1 : -1);
    int prev_c = 0;
    float prev_w = 0.f;
#pragma unroll
    for (int i = 0; i < kGmgFast; ++i)
      if (i < nf && (a.typical || i <= shift_end)) {
        const int ci = c[i];
        float wi = w[i];
        if (a.typical) wi = (float)__dmul_rn((double)wi, decay);
        if (i <= shift_end) {
          c[i] = i == 0 ? color : prev_c;
          w[i] = i == 0 ? front_w : prev_w;
          prev_c = ci, prev_w = wi;
        } else {
          w[i] = wi;
        }
      }
    bool appended = false;
    if (!found && !full) {
#pragma unroll
      for (int i = 0; i < kGmgFast; ++i)
        if (i == nf) c[i] = color, w[i] = ins;
      ++nf;
      appended = true;
    }
    if (a.typical ? appended : (a.normalize_now != 0)) {
      float total = 0.0f;
#pragma unroll
      for (int i = 0; i < kGmgFast; ++i)
        if (i < nf) total += w[i];
      if (total != 0.0f) {
#pragma unroll
        for (int i = 0; i < kGmgFast; ++i)
          if (i < nf) w[i] = div_rn(w[i], total);
      }
    }
#pragma unroll
    for (int i = 0; i < kGmgFast; ++i)
      if (i < nf) {
        int2* r = a.rec + (size_t)i * a.plane + sp;
        if (a.fullStore || i >= nf_in || c[i] != c_in[i])
          *r = make_int2(c[i], __float_as_int(w[i]));
        else
          reinterpret_cast<float*>(r)[1] = w[i];
      }
    if (nf != nf_in) a.nfeat[sp] = (uint8_t)nf;
  }
  if (active) a.raw[p0] = isfg ? 255 : 0;
}

__global__ __launch_bounds__(kBlock) void gmg_kernel(const GmgArgs a) {
  const size_t p0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = p0 < a.npix;
  const size_t sp = a.state_off + (active ? p0 : 0);
  int nf = active ? (int)a.nfeat[sp] : 0;
  int color = 0;
  if (active) {
    unsigned feat = 0;
    for (int c = 0; c < a.C; ++c)  // G1: (int)((v - 0.0) * levels / (255.0 - 0.0)) << 8c, in double
      feat |= (unsigned)(int)__ddiv_rn(__dmul_rn((double)a.frame[p0 * a.C + c], (double)a.levels), 255.0) << (8 * c);
    color = (int)feat;
  }
  if (a.F >= kGmgFast && !__any(nf >= kGmgFast)) {  // wave-uniform
    gmg_fast(a, active, sp, p0, nf, color);
    return;
  }
  // pass 1: findFeature
  int idx = -1;
  float wfound = 0.f;
  for (int i = 0;; ++i) {
    const bool look = active && i < nf && idx < 0;
    if (!__any(look)) break;
    if (look) {
      const int2 r = a.rec[(size_t)i * a.plane + sp];
      if (r.x == color) idx = i, wfound = __int_as_float(r.y);
    }
  }
  bool isfg = false;
  bool appended = false;
  if (a.typical) {
    const double w = (double)wfound;  // 0 when the colour is not in the histogram
    const double num = __dmul_rn(w, a.prior);
    const double den = __dadd_rn(num, __dmul_rn(__dsub_rn(1.0, w), __dsub_rn(1.0, a.prior)));
    const double posterior = __ddiv_rn(num, den);  // G5
    isfg = __dsub_rn(1.0, posterior) > a.thr;
  }
  if (a.update) {
    const double decay = a.typical ? __dsub_rn(1.0, a.lr) : 1.0;  // G2 (training frames do not decay)
    const float ins = a.typical ? (float)a.lr : 1.0f;
    const bool found = idx >= 0, full = !found && nf == a.F;
    const float front_w = found ? ins + (a.typical ? (float)__dmul_rn((double)wfound, decay) : wfound) : ins;
    const int shift_end = found ? idx : (full ? nf - 1 : -1);  // entries 0..shift_end move down by one (G3)
    int prev_c = 0;
    float prev_w = 0.f;
    for (int i = 0;; ++i) {
      const bool act = active && i < nf && (a.typical || i <= shift_end);  // training frames touch only what moves
      if (!__any(act)) break;
      if (act) {
        int2* rp = a.rec + (size_t)i * a.plane + sp;
        const int2 r = *rp;
        const int c = r.x;
        float w = __int_as_float(r.y);
        if (a.typical) w = (float)__dmul_rn((double)w, decay);
        if (i <= shift_end) {
          *rp = make_int2(i == 0 ? color : prev_c, __float_as_int(i == 0 ? front_w : prev_w));
          prev_c = c, prev_w = w;
        } else if (a.fullStore) {
          *rp = make_int2(c, __float_as_int(w));
        } else {
          reinterpret_cast<float*>(rp)[1] = w;
        }
      }
    }
    if (active && !found && !full) {  // append
      a.rec[(size_t)nf * a.plane + sp] = make_int2(color, __float_as_int(ins));
      ++nf;
      appended = true;
    }
    // normalizeHistogram: after an append in normal operation; for every pixel on the last training frame
    const bool norm = active && (a.typical ? appended : (a.normalize_now != 0));
    float total = 0.0f;
    for (int i = 0;; ++i) {
      const bool act = norm && i < nf;
      if (!__any(act)) break;
      if (act) total += reinterpret_cast<const float*>(a.rec + (size_t)i * a.plane + sp)[1];
    }
    for (int i = 0;; ++i) {
      const bool act = norm && total != 0.0f && i < nf;
      if (!__any(act)) break;
      if (act) {
        float* wp = reinterpret_cast<float*>(a.rec + (size_t)i * a.plane + sp) + 1;
        *wp = div_rn(*wp, total);
      }
    }
    if (active) a.nfeat[sp] = (uint8_t)nf;  // G4: the count persists on training frames too
  }
  if (active) a.raw[p0] = isfg ? 255 : 0;
}

__global__ __launch_bounds__(kBlock) void gmg_clear_kernel(uint8_t* nfeat, size_t n) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) nfeat[i] = 0;
}

}  // namespace bgs
