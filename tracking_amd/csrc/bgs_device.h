// bgs_device.h — device-side helpers shared by every kernel of libbgs_hip (gfx950 only).
//
// Numeric contract: the uint8 masks must equal the reference's bit for bit, so every
// float expression rounds exactly where the reference's does.  The library is compiled with
// -ffp-contract=off (no FMA fusion) and -fhip-fp32-correctly-rounded-divide-sqrt, so `/` and sqrt_rn() are IEEE
// correctly rounded (NOT HIP's __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is the 1-ulp native v_sqrt_f32);
// u8 conversion is round-half-to-even then clamp (cv::saturate_cast<uchar>(float)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bgs {

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kBlock = 256;     // 4 waves, one per SIMD

// IEEE correctly rounded f32 square root / division (see the header comment)
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

// cv::cvtColor(CV_BGR2GRAY) on 8U: fixed point, shift 14
__device__ __forceinline__ int gray_bgr(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14; }
// the same value from a dword whose bytes 0..2 are B, G, R (byte 3 is ignored): the three weights split into low and high bytes,
// two v_dot4_u32_u8 (1868 = 0x074C, 9617 = 0x2591, 4899 = 0x1323)
__device__ __forceinline__ uint32_t gray_bgr_dword(uint32_t px) {
  const uint32_t lo = __builtin_amdgcn_udot4(px, 0x0023914Cu, 1u << 13, false), hi = __builtin_amdgcn_udot4(px, 0x00132507u, 0u, false);
  return ((hi << 8) + lo) >> 14;
}

// cv::saturate_cast<uchar>(float): cvRound (half-to-even) then clamp
__device__ __forceinline__ int sat_u8(float v) {
  int i = __float2int_rn(v);
  return min(max(i, 0), 255);
}

// cv::threshold(..., THRESH_BINARY) when enabled, identity otherwise
__device__ __forceinline__ int thr_bin(int v, int thr, int enable) { return enable ? (v > thr ? 255 : 0) : v; }

// cv::addWeighted on 32F: (float)((double)a*alpha + (double)b*beta)   [the +gamma(0.0) is an identity for the non-negative values here]
__device__ __forceinline__ float add_weighted(float a, double alpha, float b, double beta) {
  return (float)__dadd_rn(__dmul_rn((double)a, alpha), __dmul_rn((double)b, beta));
}

// XCD-aware block order (speed only): workgroups are dealt round-robin over the 8 XCDs, so blockIdx % 8 says which blocks share
// an XCD and its L2.  Remap so that each XCD walks ONE contiguous eighth of the launch instead of every 8th block.
__device__ __forceinline__ size_t xcd_block(int enable) {
  size_t blk = blockIdx.x;
  if (enable) {
    const size_t per = gridDim.x >> 3, main = per << 3;
    if (blk < main) blk = (blk & 7) * per + (blk >> 3);
  }
  return blk;
}

// byte j (0..3) of a dword
__device__ __forceinline__ int byte_of(uint32_t w, int j) { return (int)((w >> (8 * j)) & 0xffu); }

// G consecutive pixels of C interleaved bytes, held as dwords in registers.
template <int NBYTES>
struct Bytes {
  static_assert(NBYTES % 4 == 0, "whole dwords");
  uint32_t w[NBYTES / 4];
  __device__ __forceinline__ int get(int i) const { return byte_of(w[i >> 2], i & 3); }
  __device__ __forceinline__ void set(int i, int v) { w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | ((uint32_t)v << (8 * (i & 3))); }
};

// Coalesced vector load/store of NBYTES per lane.  p must be 4-byte aligned (16-byte when NBYTES % 16 == 0).
template <int NBYTES>
__device__ __forceinline__ Bytes<NBYTES> load_bytes(const uint8_t* p) {
  Bytes<NBYTES> r;
  if constexpr (NBYTES % 16 == 0) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < NBYTES / 16; ++i) {
      uint4 v = q[i];
      r.w[4 * i] = v.x, r.w[4 * i + 1] = v.y, r.w[4 * i + 2] = v.z, r.w[4 * i + 3] = v.w;
    }
  } else {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < NBYTES / 4; ++i) r.w[i] = q[i];
  }
  return r;
}
template <int NBYTES>
__device__ __forceinline__ void store_bytes(uint8_t* p, const Bytes<NBYTES>& r) {
  if constexpr (NBYTES % 16 == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < NBYTES / 16; ++i) q[i] = make_uint4(r.w[4 * i], r.w[4 * i + 1], r.w[4 * i + 2], r.w[4 * i + 3]);
  } else {
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < NBYTES / 4; ++i) q[i] = r.w[i];
  }
}

// PX consecutive floats of one model plane
template <int PX>
__device__ __forceinline__ void load_f(const float* p, float (&d)[PX]) {
  if constexpr (PX == 4) {
    float4 v = *reinterpret_cast<const float4*>(p);
    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
  } else if constexpr (PX == 2) {
    float2 v = *reinterpret_cast<const float2*>(p);
    d[0] = v.x, d[1] = v.y;
  } else {
    d[0] = p[0];
  }
}
template <int PX>
__device__ __forceinline__ void store_f(float* p, const float (&d)[PX]) {
  if constexpr (PX == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
  } else if constexpr (PX == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
  } else {
    p[0] = d[0];
  }
}

// OR-combine the low `bits` of every lane's value into 64-bit words in pixel order and let the first lane
// of each group store it: packed foreground mask, bit i of word j = pixel 64 j + i.  Each lane holds PX mask
// bits (pixel order) in `nib`; 64/PX lanes share one word.
template <int PX>
__device__ __forceinline__ void store_packed_mask(uint64_t* words, size_t first_pixel_of_lane, uint32_t nib, bool lane_in_range) {
  constexpr int LANES = 64 / PX;  // lanes per 64-bit word
  const int lane = threadIdx.x & (kWave - 1);
  uint64_t v = (uint64_t)nib << (PX * (lane % LANES));
#pragma unroll
  for (int off = 1; off < LANES; off <<= 1) {
    uint32_t lo = __shfl_xor((uint32_t)v, off, kWave), hi = __shfl_xor((uint32_t)(v >> 32), off, kWave);
    v |= ((uint64_t)hi << 32) | lo;
  }
  if (lane % LANES == 0 && lane_in_range) words[first_pixel_of_lane >> 6] = v;
}

// LBSP descriptor of `ref` (threshold t) against the 16 neighbours of one channel, two neighbours per dword as 16-bit
// lanes: dword k = neighbour k (LBSP bit 15-k) in the high half, neighbour 8+k (bit 7-k) in the low half.
// |v - ref| > t  <=>  v outside [lo, hi] = [max(ref-t,0), min(ref+t,255)]  <=>  (v - lo) mod 2^16 > hi - lo, so one packed
// wrap-around subtract, one packed saturating subtract and a packed min give the two flags of a dword (v_pk_* on CDNA).
__device__ __forceinline__ unsigned ss_lbsp(const uint32_t (&nb)[8], int ref, int t) {
  const unsigned lo = (unsigned)max(ref - t, 0), w = (unsigned)min(ref + t, 255) - lo;
  const unsigned lo2 = lo * 0x10001u, w2 = w * 0x10001u, one = 0x10001u;
  unsigned acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    unsigned u, d, f;  // (the compiler turns the portable form of this into compare + select per half: twice the instructions)
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(u) : "v"(nb[k]), "v"(lo2));
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(u), "v"(w2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(d), "v"(one));
    acc = (acc << 1) | f;
  }
  return (acc & 0xffu) | ((acc >> 8) & 0xff00u);
}

// The 5x5 double-cross neighbourhood of one pixel out of a halo'd LDS tile, with dword reads.
// The tile rows hold interleaved C-channel bytes starting at an arbitrary byte alignment (`shift` of row ry, 0..3; the rows
// were filled with ALIGNED dword loads from the image, so the first wanted byte of a row sits `shift` bytes into it).  For tile
// pixel (ly, lx) - halo 2, so its own bytes start at shift + (lx + 2) * C of row ly + 2 - the window of row ry is the 5*C bytes
// of pixels lx .. lx+4 of that row: WD + 1 dword reads and WD v_alignbyte put them at byte 0 of w[r][...]
// (round 1 read every byte on its own: 51 ds_read_u8 + address arithmetic per pixel and channel set).
template <int C>
struct LbspWin {
  static constexpr int WD = (5 * C + 3) / 4;  // dwords per window row: 4 (BGR, 15 bytes), 2 (gray, 5 bytes)
  uint32_t w[5][WD];
  // row_dw: LDS row pitch in dwords; s0 / sstep: shift of tile row ry is (s0 + ry * sstep) & 3
  __device__ __forceinline__ void load(const uint32_t* tile, int row_dw, int ly, int lx, int s0, int sstep) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int ry = ly + r;  // tile rows ly .. ly+4 = image rows y-2 .. y+2
      const int b0 = ((s0 + ry * sstep) & 3) + lx * C, sh = b0 & 3;
      const uint32_t* q = tile + ry * row_dw + (b0 >> 2);
      uint32_t d[WD + 1];
#pragma unroll
      for (int j = 0; j <= WD; ++j) d[j] = q[j];
#pragma unroll
      for (int j = 0; j < WD; ++j) w[r][j] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], (uint32_t)sh);
    }
  }
  // byte (dx, dy, c) of the window, dx, dy in -2..2: compile-time position
  static constexpr int pos(int dx, int c) { return (dx + 2) * C + c; }
  __device__ __forceinline__ int centre(int c) const { return (int)((w[2][pos(0, c) >> 2] >> (8 * (pos(0, c) & 3))) & 0xffu); }
  // the 16 neighbours of channel c packed as ss_lbsp wants them: dword k = neighbour k << 16 | neighbour 8+k (one v_perm_b32 each)
  __device__ __forceinline__ void pack(int c, uint32_t (&nb)[8]) const {
    constexpr int dx[16] = {-1, 1, 1, -1, 1, 0, -1, 0, -2, 2, 2, -2, 0, 0, 2, -2}, dy[16] = {1, -1, 1, -1, 0, -1, 0, 1, -2, 2, -2, 2, 2, -2, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pa = pos(dx[k], c), pb = pos(dx[8 + k], c);
      // v_perm_b32 D, S0, S1, sel: selector 0-3 = bytes of S1, 4-7 = bytes of S0, 0x0c = constant 0
      const uint32_t sel = 0x0c000c00u | ((uint32_t)(4 + (pa & 3)) << 16) | (uint32_t)(pb & 3);
      nb[k] = __builtin_amdgcn_perm(w[dy[k] + 2][pa >> 2], w[dy[8 + k] + 2][pb >> 2], sel);
    }
  }
};

}  // namespace bgs
