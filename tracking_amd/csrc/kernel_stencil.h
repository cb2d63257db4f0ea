// kernel_stencil.h — the kernels with a spatial neighbourhood, staged through LDS.
//
//   K3b asbl_kernel   AdaptiveSelectiveBackgroundLearning::process  package_bgs/AdaptiveSelectiveBackgroundLearning.cpp:37-94
//                     gray -> |I-B| -> threshold -> 3x3 median (BORDER_REPLICATE) -> selective running average, one launch.
//   K7  lbsp_kernel   LBSP::computeRGBDescriptor / computeGrayscaleDescriptor  package_bgs/pl/LBSP.h:50-71
//                     with the bit order of LBSP_16bits_dbcross_{3ch3t,1ch}.i:27-43 and the per-centre threshold LUT of
//                     BackgroundSubtractorSuBSENSE.cpp:209-210, 227-228.
//   K9  morph / median kernels  (cv::erode / cv::dilate 3x3, cv::medianBlur k) for the post-processing chain of
//                     BackgroundSubtractorSuBSENSE.cpp:624-640.
//
// Mapping: one workgroup = one TW x TH output tile of one image (blockIdx.z = image of the batch); the tile plus its
// halo is loaded once into LDS with coalesced dword loads, every output is then computed from LDS.
#pragma once
#include "bgs_device.h"
#include "kernel_pointwise.h"

namespace bgs {

// ----------------------------------------------------------------------------------------------------- ASBL
struct AsblArgs {
  const uint8_t* frame;    // [S][rows][cols][C]
  const uint8_t* bg_in;    // [S][rows][cols] gray background (previous)
  uint8_t* bg_out;         // [S][rows][cols] updated background (other buffer of the ping-pong pair)
  uint8_t* fg;             // [S][rows][cols] or null
  uint8_t* bg_img;         // [S][rows][cols] or null (copy of bg_out for the caller)
  int rows, cols, thr, learn;
  double aL, bL, aD, bD;   // alphaLearn, 1-alphaLearn, alphaDetection, 1-alphaDetection
};

constexpr int kAsblTW = 64, kAsblTH = 16;  // 4 consecutive pixels per lane, 256 lanes

template <int C>
__device__ __forceinline__ int asbl_gray(const uint8_t* p) {
  if constexpr (C == 3)
    return gray_bgr(p[0], p[1], p[2]);
  else
    return p[0];
}

// thresholded |I - B| of one pixel, :50-62; also returns the two float images' values
__device__ __forceinline__ int asbl_raw(int gray, int bg8, int thr, float& i_f, float& b_f) {
  const float sf = (float)(1. / 255.);
  i_f = (float)gray * sf, b_f = (float)bg8 * sf;
  // saturate(|i/255 - b/255| * 255) == |i - b| for every byte pair (CPU test test_unit_absdiff_is_integer_absdiff)
  return abs(gray - bg8) > thr ? 1 : 0;
}

// the background update of one pixel, :69 / :83-86 (m = the pixel's mask value after the median); a foreground pixel of the detection
// phase keeps its value
__device__ __forceinline__ float asbl_update(float i_f, float b_f, int learn, int m, double aL, double bL, double aD, double bD) {
  if (learn) return add_weighted(i_f, aL, b_f, bL);                                              // :69  (MatExpr -> addWeighted)
  if (m == 0) return (float)__dadd_rn(__dmul_rn(aD, (double)i_f), __dmul_rn(bD, (double)b_f));  // :83-86 scalar double expression
  return b_f;
}

// One workgroup = 64 x 16 pixels.  Stage 1: every lane computes the thresholded difference of its 4 pixels (dword loads
// when the row pitch allows it) and keeps their float values; the 1-pixel halo ring is computed by the first 164 lanes.
// Stage 2: 3x3 majority from LDS with byte-wise SWAR sums (cells are 0/1), selective update, dword stores.
template <int C>
__global__ __launch_bounds__(kBlock) void asbl_kernel(const AsblArgs a) {
  constexpr int PITCH = kAsblTW + 8;  // tile at columns 4..67 (dword aligned), halo at 3 and 68
  __shared__ uint32_t rawd[kAsblTH + 2][PITCH / 4];
  uint8_t(*raw)[PITCH] = reinterpret_cast<uint8_t(*)[PITCH]>(rawd);
  const int x0 = blockIdx.x * kAsblTW, y0 = blockIdx.y * kAsblTH;
  const size_t img = (size_t)blockIdx.z * a.rows * a.cols;
  const uint8_t* frame = a.frame + img * C;
  const uint8_t* bg = a.bg_in + img;
  const bool vec = (a.cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.frame) | reinterpret_cast<uintptr_t>(a.bg_in) | reinterpret_cast<uintptr_t>(a.bg_out)) & 3) == 0;
  const int lx = (threadIdx.x % (kAsblTW / 4)) * 4, ly = threadIdx.x / (kAsblTW / 4);
  const int x = x0 + lx, y = y0 + ly;
  const bool inside = x < a.cols && y < a.rows;  // with vec, the 4 pixels are inside together
  float i_f[4], b_f[4];
  {
    uint8_t fb[4 * C], bb[4];
    const int yc = min(y, a.rows - 1);  // out-of-image cells replicate the nearest pixel (BORDER_REPLICATE)
    if (vec && inside) {
      const size_t p = (size_t)yc * a.cols + x;
      const uint32_t* fp = reinterpret_cast<const uint32_t*>(frame + p * C);
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const uint32_t w = fp[k];
        fb[4 * k] = (uint8_t)w, fb[4 * k + 1] = (uint8_t)(w >> 8), fb[4 * k + 2] = (uint8_t)(w >> 16), fb[4 * k + 3] = (uint8_t)(w >> 24);
      }
      const uint32_t w = *reinterpret_cast<const uint32_t*>(bg + p);
      bb[0] = (uint8_t)w, bb[1] = (uint8_t)(w >> 8), bb[2] = (uint8_t)(w >> 16), bb[3] = (uint8_t)(w >> 24);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const size_t p = (size_t)yc * a.cols + min(x + o, a.cols - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) fb[o * C + c] = frame[p * C + c];
        bb[o] = bg[p];
      }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int o = 0; o < 4; ++o) packed |= (uint32_t)asbl_raw(asbl_gray<C>(fb + o * C), bb[o], a.thr, i_f[o], b_f[o]) << (8 * o);
    rawd[ly + 1][1 + lx / 4] = packed;
  }
  if (threadIdx.x < 2 * (kAsblTW + 2) + 2 * kAsblTH) {  // halo ring: rows -1 and TH (66 cells each), columns -1 and TW (16 each)
    int hy, hx;
    const int t = threadIdx.x;
    if (t < 2 * (kAsblTW + 2))
      hy = t < kAsblTW + 2 ? -1 : kAsblTH, hx = (t % (kAsblTW + 2)) - 1;
    else
      hy = (t - 2 * (kAsblTW + 2)) % kAsblTH, hx = (t - 2 * (kAsblTW + 2)) < kAsblTH ? -1 : kAsblTW;
    const int yy = min(max(y0 + hy, 0), a.rows - 1), xx = min(max(x0 + hx, 0), a.cols - 1);
    const size_t p = (size_t)yy * a.cols + xx;
    float fi, fb2;
    raw[hy + 1][4 + hx] = (uint8_t)asbl_raw(asbl_gray<C>(frame + p * C), bg[p], a.thr, fi, fb2);
  }
  __syncthreads();
  if (!inside) return;
  // cv::medianBlur(k=3) of a {0,255} image = majority of the 9 cells; column sums of the 3 rows, byte-wise
  uint32_t s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) s0 += rawd[ly + dy][lx / 4] >> 24, s1 += rawd[ly + dy][lx / 4 + 1], s2 += rawd[ly + dy][lx / 4 + 2] & 0xffu;  // (bytes 0-2 / 69-71 of a row are never written)
  // columns x-1 .. x+4 are byte 3 of s0, bytes 0..3 of s1, byte 0 of s2
  const uint32_t left = s0, right = s2;
  const uint32_t cnt4 = s1 + ((s1 << 8) | left) + ((s1 >> 8) | (right << 24));  // per byte: c[o-1] + c[o] + c[o+1] (<= 9)
  uint32_t out_b = 0, out_m = 0;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int m = ((cnt4 >> (8 * o)) & 0xffu) >= 5 ? 255 : 0;
    const float bf = asbl_update(i_f[o], b_f[o], a.learn, m, a.aL, a.bL, a.aD, a.bD);
    out_b |= (uint32_t)sat_u8(bf * 255.f) << (8 * o);  // :92-94
    out_m |= (uint32_t)m << (8 * o);
  }
  const size_t p = img + (size_t)y * a.cols + x;
  if (vec) {
    *reinterpret_cast<uint32_t*>(a.bg_out + p) = out_b;
    if (a.bg_img && (reinterpret_cast<uintptr_t>(a.bg_img) & 3) == 0)
      *reinterpret_cast<uint32_t*>(a.bg_img + p) = out_b;
    else if (a.bg_img)
      for (int o = 0; o < 4; ++o) a.bg_img[p + o] = (uint8_t)(out_b >> (8 * o));
    if (a.fg && (reinterpret_cast<uintptr_t>(a.fg) & 3) == 0)
      *reinterpret_cast<uint32_t*>(a.fg + p) = out_m;
    else if (a.fg)
      for (int o = 0; o < 4; ++o) a.fg[p + o] = (uint8_t)(out_m >> (8 * o));
  } else {
    for (int o = 0; o < 4 && x + o < a.cols; ++o) {
      a.bg_out[p + o] = (uint8_t)(out_b >> (8 * o));
      if (a.bg_img) a.bg_img[p + o] = (uint8_t)(out_b >> (8 * o));
      if (a.fg) a.fg[p + o] = (uint8_t)(out_m >> (8 * o));
    }
  }
}

// ---- round 3: the update through a table, 1024-lane persistent workgroups (the form abl_kernel took in round 2).
// asbl_kernel above issues 56 vector instructions per pixel, most of them the float / double round trip of the background update
// (profiles/r02_byte_kernels_pmc.txt: 58 M wave-instructions per 8 x 4K launch = ~95 of its 133 us).  The updated byte is a pure
// function of (gray, background byte) for a fixed alpha and phase, so asbl_lut_kernel tabulates it with exactly the statements of
// asbl_kernel: rows 0..255 = background byte, column = gray value; row 256 = the byte a detection-phase FOREGROUND pixel keeps
// (its background only makes the float round trip of :92-94).  The gray conversion is two v_dot4_u32_u8 per pixel.
constexpr int kAsblLutRows = 257;
constexpr int kAsblSW = 256, kAsbl2Block = 1024;  // a wave's strip: 64 lanes x 4 consecutive pixels

// lut[b * 256 + g]; grid 257 x block 256
__global__ __launch_bounds__(kBlock) void asbl_lut_kernel(uint8_t* lut, int learn, double aL, double bL, double aD, double bD) {
  const int g = threadIdx.x, b = blockIdx.x;
  float i_f, b_f;
  if (b == 256) {
    asbl_raw(0, g, 0, i_f, b_f);
    lut[256 * 256 + g] = (uint8_t)sat_u8(asbl_update(i_f, b_f, learn, 255, aL, bL, aD, bD) * 255.f);
    return;
  }
  asbl_raw(g, b, 0, i_f, b_f);
  lut[b * 256 + g] = (uint8_t)sat_u8(asbl_update(i_f, b_f, learn, 0, aL, bL, aD, bD) * 255.f);
}

// The frame is cut into strips of 256 columns x R rows, one WAVE per strip: a lane owns 4 consecutive pixels (one dword of gray,
// background, mask) and walks down the rows with the thresholded differences of the row above, its own and the row below in
// registers; the 3x3 majority is a byte-wise sum of those three dwords plus the two neighbouring lanes' sums (DPP wave shifts), the
// columns left and right of the strip come from one extra pixel per row loaded by lanes 0 and 63.  No LDS besides the table, no
// barrier after it is loaded; the rows two and three ahead are in flight while a row is worked on.
// Requires cols % 4 == 0, cols >= 4, every image pointer 4-byte aligned (the host launches asbl_kernel otherwise).
template <int C>
struct AsblQuad {
  uint32_t w[C == 3 ? 3 : 1];  // the 4 pixels' frame bytes as they lie in memory
  uint32_t b;                  // their background bytes
  uint32_t hw, hb;             // lanes 0 / 63: the dword of frame / background bytes that holds the column beside the strip
};

// Row y of a strip: the lane's quad at column x, and for the strip's side columns the quad at hx (dword hk of its frame bytes).
// Coordinates are clamped into the image (BORDER_REPLICATE) - always the same loads, none in a divergent branch, so none is
// waited for before it is needed.
template <int C>
__device__ __forceinline__ AsblQuad<C> asbl_fetch(const AsblArgs& a, size_t img, int x, int hx, int hk, int y) {
  AsblQuad<C> q;
  const size_t row = img + (size_t)min(max(y, 0), a.rows - 1) * a.cols;
  const size_t p = row + min(max(x, 0), a.cols - 4), hp = row + hx;
  q.b = *reinterpret_cast<const uint32_t*>(a.bg_in + p);
  const uint32_t* fp = reinterpret_cast<const uint32_t*>(a.frame + p * C);
#pragma unroll
  for (int k = 0; k < (C == 3 ? 3 : 1); ++k) q.w[k] = fp[k];
  q.hb = *reinterpret_cast<const uint32_t*>(a.bg_in + hp);
  q.hw = reinterpret_cast<const uint32_t*>(a.frame + hp * C)[hk];
  return q;
}

template <int C>
__device__ __forceinline__ uint32_t asbl_gray4(const AsblQuad<C>& q) {
  if constexpr (C == 3)
    return gray_bgr_dword(q.w[0]) | (gray_bgr_dword(__builtin_amdgcn_alignbyte(q.w[1], q.w[0], 3)) << 8) | (gray_bgr_dword(__builtin_amdgcn_alignbyte(q.w[2], q.w[1], 2)) << 16) |
           (gray_bgr_dword(q.w[2] >> 8) << 24);
  else
    return q.w[0];
}

// bytes of 0 / 1: |gray - background| > thr for the 4 pixels
__device__ __forceinline__ uint32_t asbl_raw4(uint32_t g4, uint32_t b4, int thr) {
  uint32_t packed = 0;
#pragma unroll
  for (int o = 0; o < 4; ++o) packed |= (uint32_t)(abs((int)((g4 >> (8 * o)) & 0xffu) - (int)((b4 >> (8 * o)) & 0xffu)) > thr) << (8 * o);
  return packed;
}

template <int C>
__global__ __launch_bounds__(kAsbl2Block) void asbl_stream_kernel(const AsblArgs a, const uint8_t* __restrict__ lut, int nimg, int R) {
  __shared__ uint8_t T[kAsblLutRows * kAblLutStride];
  {
    const uint4* src = reinterpret_cast<const uint4*>(lut);
    for (int i = threadIdx.x; i < kAsblLutRows * 16; i += kAsbl2Block) {
      const uint4 v = src[i];
      uint32_t* dst = reinterpret_cast<uint32_t*>(T + (i >> 4) * kAblLutStride + (i & 15) * 16);
      dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t nsx = (a.cols + kAsblSW - 1) / kAsblSW, nby = (a.rows + R - 1) / R;
  const uint32_t per_img = nsx * nby, nstrips = (uint32_t)nimg * per_img;
  const uint32_t nwaves = gridDim.x * (kAsbl2Block / kWave);
  const size_t img_px = (size_t)a.rows * a.cols;
  for (uint32_t strip = blockIdx.x * (kAsbl2Block / kWave) + threadIdx.x / kWave; strip < nstrips; strip += nwaves) {
    const uint32_t im = strip / per_img, rem = strip % per_img;
    const int x0 = (int)(rem % nsx) * kAsblSW, y0 = (int)(rem / nsx) * R, y1 = min(y0 + R, a.rows);
    const int x = x0 + 4 * lane;
    const size_t img = im * img_px;
    // the column beside the strip (lanes 0..31: left, 32..63: right): the quad that holds it, which of its pixels it is (first / last),
    // replicated from the image's own first / last column at the image border
    const bool hlast = lane < 32 ? x0 > 0 : x0 + kAsblSW >= a.cols;
    const int hx = lane < 32 ? max(x0 - 4, 0) : min(x0 + kAsblSW, a.cols - 4);
    const int hk = (C == 3 && hlast) ? 2 : 0;
    auto side_raw = [&](const AsblQuad<C>& q) -> uint32_t {  // 0 / 1 for that one pixel
      uint32_t g;
      if constexpr (C == 3)
        g = gray_bgr_dword(hlast ? q.hw >> 8 : q.hw);
      else
        g = hlast ? q.hw >> 24 : q.hw & 0xffu;
      const uint32_t b = hlast ? q.hb >> 24 : q.hb & 0xffu;
      return (uint32_t)(abs((int)g - (int)b) > a.thr);
    };
    auto main_raw = [&](uint32_t g4, uint32_t b4) -> uint32_t {
      uint32_t r = asbl_raw4(g4, b4, a.thr);
      if (x >= a.cols) r >>= 24;  // a quad right of the image was loaded from the row's last quad: its neighbour reads byte 0 = the border pixel
      return r;
    };
    AsblQuad<C> q = asbl_fetch<C>(a, img, x, hx, hk, y0 - 1);
    uint32_t rp = main_raw(asbl_gray4<C>(q), q.b), hp = side_raw(q);
    q = asbl_fetch<C>(a, img, x, hx, hk, y0);
    uint32_t gc = asbl_gray4<C>(q), bc = q.b;
    uint32_t rc = main_raw(gc, bc), hc = side_raw(q);
    AsblQuad<C> qa = asbl_fetch<C>(a, img, x, hx, hk, y0 + 1), qb = asbl_fetch<C>(a, img, x, hx, hk, y0 + 2);
    // one output row; qn holds row y + 1 on entry and is refilled IN PLACE with row y + 3 (two register sets used in turn, below: a
    // rotation qa = qb would copy registers a load has just been issued for, and the copy waits for the load)
    auto row = [&](int y, AsblQuad<C>& qn) {
      const uint32_t gn = asbl_gray4<C>(qn), bn = qn.b;
      const uint32_t rn = main_raw(gn, bn), hn = side_raw(qn);
      qn = asbl_fetch<C>(a, img, x, hx, hk, y + 3);
      // cv::medianBlur(k=3) of a {0,255} image = majority of the 9 cells: column sums of the three rows (bytes <= 3), then the three columns
      const uint32_t s = rp + rc + rn, hs = hp + hc + hn;
      uint32_t left = __builtin_amdgcn_update_dpp(0u, s, 0x138, 0xf, 0xf, true) >> 24;  // wave_shr:1 = lane - 1's sums, its last column
      uint32_t right = __builtin_amdgcn_update_dpp(0u, s, 0x130, 0xf, 0xf, true) & 0xffu;  // wave_shl:1 = lane + 1's first column
      if (lane == 0) left = hs;
      if (lane == 63) right = hs;
      const uint32_t cnt4 = s + ((s << 8) | left) + ((s >> 8) | (right << 24));  // per byte: c[o-1] + c[o] + c[o+1] (<= 9)
      if (x < a.cols) {
        uint32_t out_b = 0, out_m = 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int m = ((cnt4 >> (8 * o)) & 0xffu) >= 5 ? 255 : 0;
          const int g = (gc >> (8 * o)) & 0xffu, b = (bc >> (8 * o)) & 0xffu;
          const int idx = (m != 0 && !a.learn) ? 256 * kAblLutStride + b : b * kAblLutStride + g;
          out_b |= (uint32_t)T[idx] << (8 * o);
          out_m |= (uint32_t)m << (8 * o);
        }
        const size_t p = img + (size_t)y * a.cols + x;
        *reinterpret_cast<uint32_t*>(a.bg_out + p) = out_b;
        if (a.bg_img) *reinterpret_cast<uint32_t*>(a.bg_img + p) = out_b;
        if (a.fg) *reinterpret_cast<uint32_t*>(a.fg + p) = out_m;
      }
      rp = rc, rc = rn, hp = hc, hc = hn, gc = gn, bc = bn;
    };
    for (int y = y0; y < y1; y += 2) {
      row(y, qa);
      if (y + 1 < y1) row(y + 1, qb);
    }
  }
}

// ----------------------------------------------------------------------------------------------------- LBSP
struct LbspArgs {
  const uint8_t* img;  // [S][rows][cols][C]
  uint16_t* desc;      // [S][rows][cols][C]
  int rows, cols;
  uint8_t lut[256];    // absolute threshold per centre value
};

// Round 2: the neighbourhood now comes out of the LDS tile with dword reads (LbspWin: 25 ds_read_b32 + 20 v_alignbyte + one
// v_perm per packed pair instead of 51 ds_read_u8) - and the kernel's time did not move (16 x 1080p in one launch: 0.234 ms,
// 142 Gpixel/s), nor did staging the output through LDS for whole-dword stores (0.28 ms).  With the compare compiled out the
// same launch takes 0.086 ms: the kernel is bound by ss_lbsp itself - 16 comparisons per channel at 2 packed instructions per
// neighbour, ~120 VALU instructions per pixel - which no access pattern changes.
constexpr int kLbspTW = 64, kLbspTH = 16;  // output tile; 256 lanes, each 4 rows of one column

// bit 15..0 -> (dx, dy) of LBSP_16bits_dbcross_3ch3t.i:27-43
__device__ __constant__ const int8_t kLbspDx[16] = {-1, 1, 1, -1, 1, 0, -1, 0, -2, 2, 2, -2, 0, 0, 2, -2};
__device__ __constant__ const int8_t kLbspDy[16] = {1, -1, 1, -1, 0, -1, 0, 1, -2, 2, -2, 2, 2, -2, 0, 0};

template <int C>
__global__ __launch_bounds__(kBlock) void lbsp_kernel(const LbspArgs a) {
  constexpr int HW = kLbspTW + 4, HH = kLbspTH + 4;   // tile + 2-pixel halo
  constexpr int ROWB = (HW * C + 3 + 3) / 4 * 4;       // LDS row pitch in bytes (room for the alignment shift)
  __shared__ uint32_t tile[HH][ROWB / 4];
  __shared__ uint8_t lut[256];
  lut[threadIdx.x] = a.lut[threadIdx.x];
  const int x0 = blockIdx.x * kLbspTW, y0 = blockIdx.y * kLbspTH;
  const size_t imgsz = (size_t)a.rows * a.cols * C;
  const uint8_t* img = a.img + (size_t)blockIdx.z * imgsz;
  // byte range of one halo'd row inside the image row: [(x0-2)*C, (x0+TW+2)*C), loaded as aligned dwords
  const long rb = (long)(x0 - 2) * C;
  for (int i = threadIdx.x; i < HH * (ROWB / 4); i += kBlock) {
    const int ry = i / (ROWB / 4), rd = i - ry * (ROWB / 4);
    const int y = min(max(y0 + ry - 2, 0), a.rows - 1);
    const long row0 = (long)y * a.cols * C;
    const long a0 = (row0 + rb) & ~3L;  // aligned-down start of this row's span (may be < 0 only for the very first bytes)
    const long off = a0 + 4L * rd;
    uint32_t v = 0;
    if (off >= 0 && off + 4 <= (long)imgsz)
      v = *reinterpret_cast<const uint32_t*>(img + off);
    else if (off < (long)imgsz && off + 4 > 0) {  // dword straddling the buffer edge: byte by byte
      for (int b = 0; b < 4; ++b)
        if (off + b >= 0 && off + b < (long)imgsz) v |= (uint32_t)img[off + b] << (8 * b);
    }
    tile[ry][rd] = v;
  }
  __syncthreads();
  const int lx = threadIdx.x % kLbspTW, lyq = threadIdx.x / kLbspTW;  // lyq in 0..3: rows lyq*4 .. lyq*4+3
  const int x = x0 + lx;
  if (x >= a.cols) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ly = lyq * 4 + r, y = y0 + ly;
    if (y >= a.rows) break;
    uint16_t* out = a.desc + (size_t)blockIdx.z * imgsz + ((size_t)y * a.cols + x) * C;
    const bool interior = x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2;  // LBSP::validateROI
    if (!interior) {
#pragma unroll
      for (int c = 0; c < C; ++c) out[c] = 0;
      continue;
    }
    LbspWin<C> win;  // dword reads of the LDS tile + v_alignbyte / v_perm packing (bgs_device.h) instead of 51 byte reads
    win.load(&tile[0][0], ROWB / 4, ly, lx, (int)(((long)(y0 - 2) * a.cols * C + rb) & 3L), (a.cols * C) & 3);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int ref = win.centre(c);
      uint32_t nb[8];  // two neighbours per dword, as ss_lbsp wants them (bit 15-k in the high half, bit 7-k in the low half)
      win.pack(c, nb);
      out[c] = (uint16_t)ss_lbsp(nb, ref, lut[ref]);
    }
  }
}

// ----------------------------------------------------------------------------------------------------- mask morphology
struct MorphArgs {
  const uint8_t* src;
  uint8_t* dst;
  int rows, cols, op, ksize;  // op 0 erode, 1 dilate (ksize x ksize box = (ksize-1)/2 iterations of 3x3), 2 median(ksize), 3 median(ksize) of a {0,255} mask
  const uint64_t* src_bits;   // op 3 only: when set, the input is this bit plane ([images][rows][W64], bit i of word w = pixel 64 w + i) instead of src
  int W64;
};

constexpr int kMorphTW = 64, kMorphTH = 4, kMorphMaxR = 7;  // median up to 15x15
constexpr int kBoxTW = 64, kBoxTH = 16;

// Box operations are separable: erode = min over the box, dilate = max, the median of a binary mask = (number of non-zero
// cells > half).  One vertical pass into LDS, one horizontal pass out of it: 2(2R+1) LDS reads per pixel instead of (2R+1)^2.
// n iterations of the 3x3 erode/dilate equal one (2n+1)x(2n+1) box: cells outside the image never take part
// (morphologyDefaultBorderValue) and the image is convex, so the iterated and the one-shot minimum run over the same cells
// (OpenCV itself folds iterations of a rectangular element into one larger element).  Median: BORDER_REPLICATE.
// OPT / RT: operation and radius as compile-time constants (-1: take them from the arguments).  The post-processing chains use
// a handful of shapes over and over (3x3 / 7x7 erode and dilate, binary medians 7 .. 13): with constant trip counts the two
// passes unroll and the per-cell `op` tests disappear.
template <int OPT, int RT>
__global__ __launch_bounds__(kBlock) void morph_box_kernel(const MorphArgs a) {
  constexpr int LW = kBoxTW + 2 * kMorphMaxR + 2;
  __shared__ uint8_t t[kBoxTH + 2 * kMorphMaxR][LW];
  __shared__ uint8_t v[kBoxTH][LW];
  const int R = RT >= 0 ? RT : a.ksize / 2, op = OPT >= 0 ? OPT : a.op;
  const int x0 = blockIdx.x * kBoxTW, y0 = blockIdx.y * kBoxTH;
  const size_t img = (size_t)blockIdx.z * a.rows * a.cols;
  const int HW = kBoxTW + 2 * R, HH = kBoxTH + 2 * R;
  for (int i = threadIdx.x; i < HW * HH; i += kBlock) {
    const int ly = i / HW, lx = i - ly * HW;
    const int y = y0 + ly - R, x = x0 + lx - R;
    uint8_t c;
    if (op == 3) {
      const int yc = min(max(y, 0), a.rows - 1), xc = min(max(x, 0), a.cols - 1);
      if (a.src_bits)
        c = (a.src_bits[((size_t)blockIdx.z * a.rows + yc) * a.W64 + (xc >> 6)] >> (xc & 63)) & 1ull;
      else
        c = a.src[img + (size_t)yc * a.cols + xc] != 0;
    } else {
      const bool in = y >= 0 && y < a.rows && x >= 0 && x < a.cols;
      c = in ? a.src[img + (size_t)y * a.cols + x] : (op == 0 ? 255 : 0);
    }
    t[ly][lx] = c;
  }
  __syncthreads();
  for (int task = threadIdx.x; task < HW * (kBoxTH / 4); task += kBlock) {  // vertical pass: one column, four output rows
    const int strip = task / HW, cx = task - strip * HW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ry = strip * 4 + r;
      int acc = op == 0 ? 255 : 0;
#pragma unroll
      for (int dy = 0; dy <= 2 * R; ++dy) {
        const int c = t[ry + dy][cx];
        acc = op == 0 ? min(acc, c) : op == 1 ? max(acc, c) : acc + c;
      }
      v[ry][cx] = (uint8_t)acc;
    }
  }
  __syncthreads();
  const int ly = threadIdx.x / (kBoxTW / 4), xs = (threadIdx.x % (kBoxTW / 4)) * 4;  // horizontal pass: four pixels of one row
  const int y = y0 + ly, x = x0 + xs;
  if (y >= a.rows || x >= a.cols) return;
  const int need = ((2 * R + 1) * (2 * R + 1)) / 2 + 1;
  uint32_t packed = 0;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    int acc = op == 0 ? 255 : 0;
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) {
      const int c = v[ly][xs + o + dx];
      acc = op == 0 ? min(acc, c) : op == 1 ? max(acc, c) : acc + c;
    }
    if (op == 3) acc = acc >= need ? 255 : 0;
    packed |= (uint32_t)acc << (8 * o);
  }
  uint8_t* out = a.dst + img + (size_t)y * a.cols + x;
  if (x + 3 < a.cols && ((img + (size_t)y * a.cols + x) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dst) & 3) == 0) {
    *reinterpret_cast<uint32_t*>(out) = packed;
  } else {
    for (int o = 0; o < 4 && x + o < a.cols; ++o) out[o] = (uint8_t)(packed >> (8 * o));
  }
}

// exact median of a u8 image by counting (BORDER_REPLICATE): the value v such that #(<= v) first reaches (k*k)/2 + 1
__global__ __launch_bounds__(kBlock) void median_kernel(const MorphArgs a) {
  __shared__ uint8_t t[kMorphTH + 2 * kMorphMaxR][kMorphTW + 2 * kMorphMaxR + 2];
  const int R = a.ksize / 2;
  const int x0 = blockIdx.x * kMorphTW, y0 = blockIdx.y * kMorphTH;
  const size_t img = (size_t)blockIdx.z * a.rows * a.cols;
  const int HW = kMorphTW + 2 * R, HH = kMorphTH + 2 * R;
  for (int i = threadIdx.x; i < HW * HH; i += kBlock) {
    const int ly = i / HW, lx = i - ly * HW;
    const int y = y0 + ly - R, x = x0 + lx - R;
    t[ly][lx] = a.src[img + (size_t)min(max(y, 0), a.rows - 1) * a.cols + min(max(x, 0), a.cols - 1)];
  }
  __syncthreads();
  const int lx = threadIdx.x % kMorphTW, ly = threadIdx.x / kMorphTW;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= a.cols || y >= a.rows) return;
  const int need = (a.ksize * a.ksize) / 2 + 1;
  int lo = 0, hi = 255;
  while (lo < hi) {  // binary search over the 8 bits
    const int mid = (lo + hi) >> 1;
    int cnt = 0;
    for (int dy = 0; dy <= 2 * R; ++dy)
      for (int dx = 0; dx <= 2 * R; ++dx) cnt += t[ly + dy][lx + dx] <= mid;
    if (cnt >= need)
      hi = mid;
    else
      lo = mid + 1;
  }
  a.dst[img + (size_t)y * a.cols + x] = (uint8_t)lo;
}

// byte mask -> bit mask (bit i of word j = pixel 64 j + i is non-zero): one wave ballot per word; n % 64 == 0
__global__ __launch_bounds__(kBlock) void mask_pack_kernel(const uint8_t* src, uint64_t* dst, size_t n) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long w = __ballot(p < n && src[p] != 0);
  if ((threadIdx.x & (kWave - 1)) == 0 && p < n) dst[p >> 6] = w;
}

// The same for `count` images of n pixels each when n is NOT a multiple of 64: image k owns words [k W, (k + 1) W), W = ceil(n / 64),
// and the bits of its last word past pixel n - 1 are zero.  One wave per word.
__global__ __launch_bounds__(kBlock) void mask_pack_ragged_kernel(const uint8_t* src, uint64_t* dst, size_t n, size_t W, size_t count) {
  const size_t word = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;  // wave-uniform
  if (word >= W * count) return;
  const size_t img = word / W, i = (word - img * W) * 64 + (threadIdx.x & (kWave - 1));
  const unsigned long long w = __ballot(i < n && src[img * n + i] != 0);
  if ((threadIdx.x & (kWave - 1)) == 0) dst[word] = w;
}

// host side: one launch over `count` images stored back to back
inline void morph_launch(const MorphArgs& a, int count, hipStream_t s) {
  if (a.op == 2)
    hipLaunchKernelGGL(median_kernel, dim3((a.cols + kMorphTW - 1) / kMorphTW, (a.rows + kMorphTH - 1) / kMorphTH, count), dim3(kBlock), 0, s, a);
  else {
    const dim3 grid((a.cols + kBoxTW - 1) / kBoxTW, (a.rows + kBoxTH - 1) / kBoxTH, count), block(kBlock);
    const int R = a.ksize / 2;
#define MORPH_CASE(OPV, RV) \
  if (a.op == OPV && R == RV) {                                                       \
    hipLaunchKernelGGL((morph_box_kernel<OPV, RV>), grid, block, 0, s, a);            \
    return;                                                                           \
  }
    MORPH_CASE(0, 1) MORPH_CASE(1, 1) MORPH_CASE(0, 3) MORPH_CASE(1, 3)                 // 3x3 and (3 iterations =) 7x7 erode / dilate
    MORPH_CASE(3, 3) MORPH_CASE(3, 4) MORPH_CASE(3, 5) MORPH_CASE(3, 6)                 // binary medians 7 (GMG), 9 (LOBSTER, SuBSENSE), 11, 13
#undef MORPH_CASE
    hipLaunchKernelGGL((morph_box_kernel<-1, -1>), grid, block, 0, s, a);
  }
}

}  // namespace bgs
