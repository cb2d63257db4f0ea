// kernel_subsense.h — K7/K8/K9 for SuBSENSEBGS (package_bgs/pl/SuBSENSE.cpp:21-45 over BackgroundSubtractorSuBSENSE.cpp),
// BGR (C = 3, :437-584) and grayscale (C = 1, :306-434) paths.
//
//   ss_phase_a_kernel    per-pixel loop of operator()            BackgroundSubtractorSuBSENSE.cpp:437-584
//                        (thresholds :459-463, LBSP intra :465-466, sample consensus :469-497, rolling means :498-522,
//                         update decisions :508-551, feedback T/v/R :553-576, non-zero-desc count :577-578, last frame :579-582)
//   ss_phase_b_kernel    the sample writes decided in phase A (self update + neighbour diffusion), applied in raster order
//   ss_refresh_kernel    refreshModel                             :249-291   (also the model initialisation, :246)
//   ss_blink_kernel / ss_bits_* / flood kernels / ss_finish_kernel   post-processing chain :624-642 (on bit planes)
//   ss_downsample_kernel + ss_frame_level_kernel                  frame-level block :643-699 (runs on the device: no host sync)
//   ss_background_kernel getBackgroundImage                       :702-718
//
// Parallel-safe contract (the CPU restatement the tests compare against follows the same one): random draws are ss_rand(frame, pixel, slot)
// instead of the reference's sequential libc rand(); a frame is two phases (classify everything against the model as it
// stood at the start of the frame, then apply all sample writes, the highest source pixel index winning a conflict);
// a pixel reads its random neighbour's D_last / raw-segmentation means from the previous frame's copy.
// (Round 3: the SELF updates are physically stored by phase A - nobody else reads a pixel's samples there - which changes no
// result: see the comment at the store in ss_phase_a_kernel.)
//
// Background samples are RECORDS: colour and LBSP descriptor of one sample of one pixel together in 16 bytes (BGR: b g r 0 |
// d0 d1 | d2 0 | 0; 4 bytes for gray: c 0 | d), one vector load per sample in the sample-consensus loops, one store per model
// update.  Order (ss_rec): LOBSTER sample-major [nS][N]; SuBSENSE the first four samples sample-major, the rest pixel-major in
// double batches of eight, because its phase A reads a pixel's samples four and then eight at a time.
// How round 2 got there on 8 x 1080p S_surv (profiles/r02_subsense_phase_a_pmc.txt): planar colour / descriptor
// arrays (six loads per sample) -> these records (phase B's scattered writes one store each) -> the sample prefetch freed from a
// misplaced s_waitcnt (2.95 -> 2.56 ms; every earlier layout experiment had been latency-bound on it and said nothing) -> the L1's
// miss capacity as the limit, hence batches (2.40) -> rejection tests per batch, inter-LBSP per candidate pass (2.17) -> eight
// samples per trip (1.99): VALU-bound.
// Ten f32 maps [N] (+2 second copies), byte maps [N].  The current frame's 5x5 neighbourhood is staged through LDS once per
// workgroup (phase A: 64x32 pixel tile + halo 2, with an LDS work queue over its pixels; phase B: 64x16 targets; LOBSTER's phase A: 64x4).
//   lob_phase_a_kernel   LOBSTER's operator()                     BackgroundSubtractorLOBSTER.cpp:172-284 (shares phase B, refresh, background)
#pragma once
#include "bgs_device.h"

namespace bgs {

constexpr int kSsTW = 64, kSsTH = 4;  // one lane per pixel

// per-stream scalars that the frame-level block updates ON THE DEVICE
struct SsScalars {
  int framesSinceReset, cooldown, autoReset, doRefresh;
  float capLo, capHi, lastNZ;
  unsigned nzCount, totDiff;
  int pad[7];
};

struct SsArgs {
  const uint8_t* frame;  // [S][N][3]
  void* samples;         // records (SsSample<C>), order: ss_rec()
  int nSpad, pixelMajor; // records held per pixel (nS padded to 4 + whole groups of 8); 1: SuBSENSE's order, 0: sample-major planes
  float *R, *V, *T, *DlastOld, *DlastNew, *DminLT, *DminST, *RawLT, *RawSTOld, *RawSTNew, *FinLT, *FinST;  // [S][N]
  uint8_t *unstable, *blinks, *lastFG, *lastRaw, *lastRawBlink, *lastDilInv, *lastColor;                  // [S][N] ([S][N][3])
  uint16_t* lastDesc;    // [S][N][3]
  uint8_t* raw;          // [S][N] phase A's raw segmentation
  uint16_t* req;         // [S][N][2]
  uint8_t* lut;          // [S][256]
  SsScalars* sc;         // [S]
  float *dsLT, *dsST;    // [S][dsh][dsw][3]
  uint8_t* fg;           // [S][N] output mask (may be null)
  uint8_t* bgimg;        // [S][N][3] output background (may be null)
  int rows, cols, nS, nReq, nMinColor, nDescOff, nMov, lbspOff, use3x3, lrScaling, medK, refill, ipassMin;
  float relT, fLT, fST;
  unsigned frameIndex;
  int first;             // first stream of this launch (blockIdx.z is relative to it)
  int selfInA;           // 1: a pixel's SELF update is stored before phase B (SuBSENSE: by ss_feedback_kernel; round 3: by phase A itself); phase B then only applies the diffusion
  uint4* lastRec;        // [S][N] scratch of the full refresh (BGR): a pixel's last colour, foreground flag and descriptors as ONE 16-byte record
  uint32_t* ho;          // [S][N][2] hand-over from phase A to ss_feedback_kernel (round 4), see there
  SsScalars* scSnap;     // [S] the per-stream scalars as they stood when phase A ran (the frame-level block rewrites `sc` beside ss_feedback_kernel)
  const uint32_t* magic; // [1024] multipliers of ss_mod
};

__host__ __device__ __forceinline__ uint32_t ss_rand(uint32_t frame, uint32_t pixel, uint32_t draw) {
  uint32_t x = frame * 0x9E3779B1u;
  x ^= pixel + 0x85EBCA6Bu + (x << 6) + (x >> 2);
  x ^= (draw + 1u) * 0xC2B2AE35u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x >> 1;
}

// x % d for x < 2^31 (every ss_rand draw) and 1 <= d < kSsMagicN without a division: Granlund & Montgomery, "Division by invariant integers
// using multiplication" (1994), theorem 4.2 with N = 31: for l = ceil(log2 d) and m = ceil(2^(31 + l) / d) < 2^32,
// floor(x / d) = floor(m x / 2^(31 + l)) for every 0 <= x < 2^31.  m comes from a table built on the host (ss_magic);
// a runtime 32-bit `%` is ~30 instructions on this hardware and the per-pixel rules take five of them (:508-551).
constexpr int kSsMagicN = 576;  // covers every learning rate (<= 512, the upper cap) and the usual sample counts; larger divisors take the plain way (2.3 KB of LDS in phase A)
inline uint32_t ss_magic(uint32_t d) {  // host
  if (d < 2) return 0;
  int l = 0;
  while ((1u << l) < d) ++l;
  return (uint32_t)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
}
__device__ __forceinline__ uint32_t ss_mod(uint32_t x, uint32_t d, const uint32_t* magic) {
  if (d - 2u >= (uint32_t)(kSsMagicN - 2)) return d > 1u ? x % d : 0u;  // d = 1: 0; d >= kSsMagicN (a sample count that large): the plain way
  const uint32_t q = __umulhi(x, magic[d]) >> (31 - __clz((int)(d - 1u)));
  return x - q * d;
}

__device__ __constant__ const int8_t kSsPattern[7][7] = {{2, 4, 6, 7, 6, 4, 2},     {4, 8, 12, 14, 12, 8, 4},  {6, 12, 21, 25, 21, 12, 6}, {7, 14, 25, 28, 25, 14, 7},
                                                         {6, 12, 21, 25, 21, 12, 6}, {4, 8, 12, 14, 12, 8, 4}, {2, 4, 6, 7, 6, 4, 2}};  // RandUtils.h:13-25
__device__ __constant__ const int8_t kSsN3[8][2] = {{-1, 1}, {0, 1}, {1, 1}, {-1, 0}, {1, 0}, {-1, -1}, {0, -1}, {1, -1}};  // RandUtils.h:51-56
__device__ __constant__ const int8_t kSsN5[24][2] = {{-2, 2},  {-1, 2},  {0, 2},  {1, 2},  {2, 2},  {-2, 1},  {-1, 1},  {0, 1},  {1, 1},  {2, 1},  {-2, 0},  {-1, 0},
                                                     {1, 0},   {2, 0},   {-2, -1}, {-1, -1}, {0, -1}, {1, -1}, {2, -1}, {-2, -2}, {-1, -2}, {0, -2}, {1, -2}, {2, -2}};

// An update request, 16 bits: bit 15 valid | bits 5..14 the sample slot (0..1023: ./config/SuBSENSEBGS.xml may set any nBGSamples,
// SuBSENSE.cpp:69) | bits 0..4 the target, (dy + 2) * 5 + (dx + 2) relative to the source (12 = the source itself).
#define SS_REQ_VALID 0x8000u
constexpr int kSsMaxSamples = 1023;
__device__ __forceinline__ uint16_t ss_req(unsigned slot, int code) { return (uint16_t)(SS_REQ_VALID | (slot << 5) | (unsigned)code); }
__device__ __forceinline__ unsigned ss_req_slot(unsigned r) { return (r >> 5) & 0x3ffu; }

// One background sample: colour c[C] + descriptor d[C], packed (see the layout note at the top of the file).
template <int C>
struct SsSample;
template <>
struct SsSample<3> {
  uint4 v;
  static constexpr size_t kBytes = 16;
  __device__ __forceinline__ int color(int c) const { return (int)((v.x >> (8 * c)) & 0xffu); }
  __device__ __forceinline__ unsigned desc(int c) const { return c == 0 ? (v.y & 0xffffu) : c == 1 ? (v.y >> 16) : (v.z & 0xffffu); }
  __device__ __forceinline__ static SsSample make(const int (&col)[3], const unsigned (&d)[3]) {
    SsSample r;
    r.v = make_uint4((uint32_t)col[0] | ((uint32_t)col[1] << 8) | ((uint32_t)col[2] << 16), (d[0] & 0xffffu) | (d[1] << 16), d[2] & 0xffffu, 0u);
    return r;
  }
  __device__ __forceinline__ static SsSample load(const void* base, size_t rec) {
    SsSample r;
    r.v = reinterpret_cast<const uint4*>(base)[rec];
    return r;
  }
  __device__ __forceinline__ void store(void* base, size_t rec) const { reinterpret_cast<uint4*>(base)[rec] = v; }
};
template <>
struct SsSample<1> {
  uint32_t v;
  static constexpr size_t kBytes = 4;
  __device__ __forceinline__ int color(int) const { return (int)(v & 0xffu); }
  __device__ __forceinline__ unsigned desc(int) const { return v >> 16; }
  __device__ __forceinline__ static SsSample make(const int (&col)[1], const unsigned (&d)[1]) {
    SsSample r;
    r.v = (uint32_t)col[0] | (d[0] << 16);
    return r;
  }
  __device__ __forceinline__ static SsSample load(const void* base, size_t rec) {
    SsSample r;
    r.v = reinterpret_cast<const uint32_t*>(base)[rec];
    return r;
  }
  __device__ __forceinline__ void store(void* base, size_t rec) const { reinterpret_cast<uint32_t*>(base)[rec] = v; }
};

// Pins the point where a prefetched sample is first "used": an empty asm that reads and re-defines its registers.  Without it the
// compiler is free to copy the freshly loaded registers at once, which puts the s_waitcnt right behind the load.
__device__ __forceinline__ void ss_wait_here(SsSample<3>& s) { asm volatile("" : "+v"(s.v.x), "+v"(s.v.y), "+v"(s.v.z), "+v"(s.v.w)); }
__device__ __forceinline__ void ss_wait_here(SsSample<1>& s) { asm volatile("" : "+v"(s.v)); }

// Record index of sample k of pixel p (0..N) of absolute stream `stream`.
//   pixelMajor == 0 (LOBSTER): sample-major planes [nS][N].
//   pixelMajor == 1 (SuBSENSE): per stream, the FIRST batch (samples 0 .. kSsBatch-1) as sample-major planes [kSsBatch][N] - the
//     samples every pixel tests, and a quiet scene tests nothing else: neighbouring pixels' records share cache lines - followed
//     by the remaining nSpad - kSsBatch samples PIXEL-major [N][nSpad - kSsBatch]: a pixel that walks on fetches its next eight
//     samples as 128 contiguous bytes (ss_phase_a_kernel).
constexpr int kSsBatch = 4;
__device__ __forceinline__ size_t ss_rec(const SsArgs& a, int stream, size_t N, size_t p, int k) {
  if (!a.pixelMajor) return ((size_t)stream * (size_t)a.nS + (size_t)k) * N + p;
  const size_t base = (size_t)stream * N * (size_t)a.nSpad;
  return k < kSsBatch ? base + (size_t)k * N + p : base + (size_t)kSsBatch * N + p * (size_t)(a.nSpad - kSsBatch) + (size_t)(k - kSsBatch);
}

// ----------------------------------------------------------------------------------------------- phase A
// One workgroup owns a 64 x 32 pixel tile (8 pixels per lane) and works in three stages:
//   1. per pixel (static lane<->pixel map): thresholds from R / unstable, the three intra-LBSP descriptors -> LDS context;
//   2. the sample-consensus loop (:469-497).  Its trip count is 2 for a quiet pixel and nS = 50 for one that matches
//      nothing; with one pixel per lane a wave runs as long as its slowest pixel (measured: mean 7.7 trips per pixel, 26 per
//      64-pixel wave).  So the lanes pull pixels from a queue in LDS instead: a wave refills its idle lanes whenever
//      a quarter of them (kSsRefill) have finished, and every lane keeps its own sample index;
//   3. per pixel (static map again): rolling means, update requests, T / v / R feedback, stores.
// The result of a pixel does not depend on when or where it is processed: every model read is of start-of-frame state.
constexpr int kSsATH = 32, kSsAPix = kSsTW * kSsATH, kSsRefill = 16, kSsIpassMin = 16;
constexpr unsigned kSsBLdsPad = 22000;  // unused dynamic LDS of phase B beside the chain: 4 instead of 8 of its workgroups per CU (engine_subsense.h; BGS_SS_B_LDS_PAD)
constexpr int kSsParts = 1;  // parts a large batch is cut into (engine_subsense.h: ss_process); BGS_SS_PARTS
constexpr uint32_t kSsNotInterior = 0xffffffffu;


// SPLIT: the per-pixel rules behind the loop (:498-576) run in ss_feedback_kernel instead of stage 3 (see there for what was measured)
// QUEUE (BGR only): the inter-LBSP tests of a wave go through a per-wave work list in LDS and are computed by WHOEVER is free, 64 at a
// time (stage 2, "rounds"); the tile is 64 x 16 then (kSsQATH) so that the extra LDS still allows four workgroups per CU.
constexpr int kSsQATH = 16, kSsQList = 256;
#ifdef BGS_SS_STATS  // experiment builds only (tools/r04_ss_stats.sh): what the waves of stage 2 did, summed over the process
__device__ unsigned long long g_ss_stats[16];
#define SS_STAT(i, v) (st[i] += (unsigned)(v))
#else
#define SS_STAT(i, v) ((void)0)
#endif
template <int C, bool SPLIT, bool QUEUE>
__global__ __launch_bounds__(kBlock) void ss_phase_a_kernel(const SsArgs a) {
  static_assert(!QUEUE || C == 3, "the candidate queue is built for BGR records");
  constexpr int ATH = QUEUE ? kSsQATH : kSsATH, APIX = kSsTW * ATH;
  constexpr int HW = kSsTW + 4, HH = ATH + 4;
  constexpr int ROWB = (HW * C + 3 + 3) / 4 * 4;
  constexpr uint32_t maxColor = 255 * C, maxDesc = 16 * C;  // s_nColorMaxDataRange_*, s_nDescMaxDataRange_*
  constexpr int PPL = APIX / kBlock;
  __shared__ uint32_t tile[HH][ROWB / 4];
  __shared__ uint8_t lut[256];
  // per pixel: [0] intra0 | intra1 << 16, [1] intra2 | colorThr << 16, [2] descThr, replaced in stage 2 by
  // good | minDesc << 8 | minSum << 16 (3-dword stride: conflict-free for consecutive pixels)
  __shared__ uint32_t ctx[APIX][3];
  __shared__ unsigned nz_block, qhead;
  // QUEUE: what a worker lane needs of the lane that owns a candidate - its pixel's 48 LBSP neighbours as bytes (12 dwords: channel c
  // in words 4c .. 4c+3, neighbour k in byte k), its current colour, where its held samples start and whether they are the pixel-major
  // part - plus, per wave, the work list (lane << 3 | sample) and, per lane, the eight 16-bit results
  __shared__ uint32_t nbs[QUEUE ? kBlock : 1][12];
  __shared__ uint32_t hcur[QUEUE ? kBlock : 1], hrec[QUEUE ? kBlock : 1];
  __shared__ uint16_t hq[QUEUE ? kBlock : 2];
  __shared__ uint16_t qlist[QUEUE ? (kBlock / kWave) * kSsQList : 2];
  __shared__ __attribute__((aligned(16))) uint16_t qres[QUEUE ? kBlock * 8 : 8];
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const uint8_t* img = a.frame + (size_t)blockIdx.z * N * C;
  const int x0 = blockIdx.x * kSsTW, y0 = blockIdx.y * ATH;
  const long imgsz = (long)N * C, rb = (long)(x0 - 2) * C;
  for (int i = threadIdx.x; i < HH * (ROWB / 4); i += kBlock) {
    const int ry = i / (ROWB / 4), rd = i - ry * (ROWB / 4);
    const int y = min(max(y0 + ry - 2, 0), a.rows - 1);
    const long off = (((long)y * a.cols * C + rb) & ~3L) + 4L * rd;
    uint32_t v = 0;
    if (off >= 0 && off + 4 <= imgsz)
      v = *reinterpret_cast<const uint32_t*>(img + off);
    else if (off < imgsz && off + 4 > 0)
      for (int b = 0; b < 4; ++b)
        if (off + b >= 0 && off + b < imgsz) v |= (uint32_t)img[off + b] << (8 * b);
    tile[ry][rd] = v;
  }
  lut[threadIdx.x] = a.lut[(size_t)stream * 256 + threadIdx.x];
  if (threadIdx.x == 0) nz_block = 0, qhead = 0;
  __syncthreads();
  // current colour and the 16 LBSP neighbours (packed as ss_lbsp wants them) of tile pixel (ly, lx): dword reads of the LDS
  // tile + v_alignbyte / v_perm (bgs_device.h: LbspWin) instead of one ds_read_u8 per byte
  const int rowShift0 = (int)(((long)(y0 - 2) * a.cols * C + rb) & 3L), rowShiftStep = (a.cols * C) & 3;
  auto gather = [&](int ly, int lx, int (&cur)[C], uint32_t (&nb)[C][8]) {
    LbspWin<C> win;
    win.load(&tile[0][0], ROWB / 4, ly, lx, rowShift0, rowShiftStep);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      cur[c] = win.centre(c);
      win.pack(c, nb[c]);
    }
  };
  auto interior_of = [&](int x, int y) { return x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2; };  // LBSP::validateROI; border pixels are never touched

  __shared__ uint32_t magic[SPLIT ? 1 : kSsMagicN];  // ss_mod's multipliers (stage 3)
  if constexpr (!SPLIT) {
    for (int k = threadIdx.x; k < kSsMagicN; k += kBlock) magic[k] = a.magic[k];
  }
  unsigned nzcount = 0;
  if constexpr (SPLIT) {
    // ---- stage 1: thresholds and intra descriptors; and everything of :498-582 that needs the frame or the pixel's last colour /
    // descriptor - the distance to the last frame (:498), the new instability flag (:467), the non-zero-descriptor count (:577-578), the
    // last colour / descriptor themselves (:579-582) - so that the rest of the per-pixel rules can run in a kernel of its own
    // (ss_feedback_kernel) that needs neither the LDS tile nor this kernel's registers.  Pixels go through in half-batches: the loads of
    // four pixels are issued before the first of them is worked on.
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.scSnap[stream] = a.sc[stream];
    {
      constexpr int HB = PPL / 2;
  #pragma unroll
      for (int h = 0; h < 2; ++h) {
        float Rs[HB], rawLT[HB], rawST[HB], finLT[HB], finST[HB];
        uint32_t us[HB], lfg[HB], blk[HB], lcw[HB];
        uint2 ldw[HB];
  #pragma unroll
        for (int r = 0; r < HB; ++r) {  // all loads first
          const int q = (h * HB + r) * kBlock + threadIdx.x, x = x0 + (q % kSsTW), y = y0 + (q / kSsTW);
          const bool in = interior_of(x, y);
          const size_t i = in ? sN + (size_t)y * a.cols + x : sN;
          Rs[r] = a.R[i], us[r] = a.unstable[i];
          rawLT[r] = a.RawLT[i], rawST[r] = a.RawSTOld[i], finLT[r] = a.FinLT[i], finST[r] = a.FinST[i], lfg[r] = a.lastFG[i], blk[r] = a.blinks[i];
          if constexpr (C == 3) {  // 3 bytes / 3 words with one 4- / 8-byte load (both arrays are padded by 8 bytes)
            lcw[r] = *reinterpret_cast<const uint32_t*>(a.lastColor + i * 3);
            ldw[r] = *reinterpret_cast<const uint2*>(a.lastDesc + i * 3);
          } else {
            lcw[r] = a.lastColor[i], ldw[r] = make_uint2(a.lastDesc[i], 0u);
          }
        }
  #pragma unroll
        for (int r = 0; r < HB; ++r) {
          const int q = (h * HB + r) * kBlock + threadIdx.x, lx = q % kSsTW, ly = q / kSsTW;
          if (!interior_of(x0 + lx, y0 + ly)) {
            ctx[q][2] = kSsNotInterior;
            continue;
          }
          const float Rv = Rs[r];
          const int unst_old = (int)us[r];
          const int stabOff = a.nMinColor / 5;
          const uint32_t colorThr = (uint32_t)((Rv * (float)a.nMinColor) - (float)((!unst_old) * stabOff)) / (C == 1 ? 2 : 1);                  // :459 / :328 (trailing /2)
          const uint32_t descThr = (1u << ((uint32_t)floorf(Rv + 0.5f))) + (uint32_t)a.nDescOff + (uint32_t)(unst_old * a.nDescOff);        // :460
          int cur[C];
          uint32_t nb[C][8];
          gather(ly, lx, cur, nb);
          unsigned intra[3] = {0, 0, 0};
  #pragma unroll
          for (int c = 0; c < C; ++c) intra[c] = ss_lbsp(nb[c], cur[c], lut[cur[c]]);  // :465-466 / :331
          // the thresholds are only ever compared with distances <= 765: clamping them to 16 bits changes no comparison
          ctx[q][0] = intra[0] | (intra[1] << 16);
          ctx[q][1] = intra[2] | (min(colorThr, 0xffffu) << 16);
          ctx[q][2] = min(descThr, 0xffffu);
          {
            const size_t i = sN + (size_t)(y0 + ly) * a.cols + (x0 + lx);
            uint32_t l1 = 0, hd = 0;  // :498
            if constexpr (C == 3) {
              const uint32_t cw = (uint32_t)cur[0] | ((uint32_t)cur[1] << 8) | ((uint32_t)cur[2] << 16);
              l1 = __builtin_amdgcn_sad_u8(cw, lcw[r] & 0xffffffu, 0u);
              hd = (uint32_t)__popc((ldw[r].x ^ (intra[0] | (intra[1] << 16)))) + (uint32_t)__popc((ldw[r].y ^ intra[2]) & 0xffffu);
            } else {
              l1 = (uint32_t)abs((int)lcw[r] - cur[0]);
              hd = (uint32_t)__popc((ldw[r].x ^ intra[0]) & 0xffffu);
            }
            const uint32_t unst = (Rv > 3.0f || (rawLT[r] - finLT[r]) > 0.1f || (rawST[r] - finST[r]) > 0.1f) ? 1u : 0u;  // :467
            a.unstable[i] = (uint8_t)unst;
            a.ho[i * 2 + 1] = l1 | (hd << 10) | (unst << 16) | ((uint32_t)(lfg[r] != 0) << 17) | ((uint32_t)(blk[r] != 0) << 18);
            if constexpr (C == 3)
              nzcount += (__popc(intra[0]) + __popc(intra[1]) + __popc(intra[2])) >= 4;  // :577-578
            else
              nzcount += __popc(intra[0]) >= 2;  // :430-431
  #pragma unroll
            for (int c = 0; c < C; ++c) {  // :579-582
              a.lastDesc[i * C + c] = (uint16_t)intra[c];
              a.lastColor[i * C + c] = (uint8_t)cur[c];
            }
          }
        }
      }
    }
  } else {
    // ---- stage 1: thresholds and intra descriptors
    {
      float Rs[PPL];
      int us[PPL];
  #pragma unroll
      for (int r = 0; r < PPL; ++r) {  // all loads first
        const int q = r * kBlock + threadIdx.x, x = x0 + (q % kSsTW), y = y0 + (q / kSsTW);
        const bool in = interior_of(x, y);
        const size_t i = sN + (size_t)y * a.cols + x;
        Rs[r] = in ? a.R[i] : 1.0f, us[r] = in ? a.unstable[i] : 0;
      }
  #pragma unroll
      for (int r = 0; r < PPL; ++r) {
        const int q = r * kBlock + threadIdx.x, lx = q % kSsTW, ly = q / kSsTW;
        if (!interior_of(x0 + lx, y0 + ly)) {
          ctx[q][2] = kSsNotInterior;
          continue;
        }
        const float Rv = Rs[r];
        const int unst_old = us[r];
        const int stabOff = a.nMinColor / 5;
        const uint32_t colorThr = (uint32_t)((Rv * (float)a.nMinColor) - (float)((!unst_old) * stabOff)) / (C == 1 ? 2 : 1);                  // :459 / :328 (trailing /2)
        const uint32_t descThr = (1u << ((uint32_t)floorf(Rv + 0.5f))) + (uint32_t)a.nDescOff + (uint32_t)(unst_old * a.nDescOff);        // :460
        int cur[C];
        uint32_t nb[C][8];
        gather(ly, lx, cur, nb);
        unsigned intra[3] = {0, 0, 0};
  #pragma unroll
        for (int c = 0; c < C; ++c) intra[c] = ss_lbsp(nb[c], cur[c], lut[cur[c]]);  // :465-466 / :331
        // the thresholds are only ever compared with distances <= 765: clamping them to 16 bits changes no comparison
        ctx[q][0] = intra[0] | (intra[1] << 16);
        ctx[q][1] = intra[2] | (min(colorThr, 0xffffu) << 16);
        ctx[q][2] = min(descThr, 0xffffu);
      }
    }
  }
  __syncthreads();

  // ---- stage 2: sample consensus, lanes fed from the queue
  // rocprofv3 (round 2, 8 x 1080p S_surv): 21 samples are tested per pixel and 2.6 of them get as far as the inter-LBSP step.  The
  // rejection tests are written for instruction count - the three colour distances with v_sad_u8 on masked words, the three intra
  // descriptor distances from two XORs, the per-channel colour test folded into the bound on sd (sd >= cd, see below) - because at
  // the end of round 2 the kernel is VALU-bound (92 % of the SIMD cycles are vector issue).
  {
    const int lane = threadIdx.x & (kWave - 1);
    bool active = false, qempty = false, fresh = false, wide = false;
    uint32_t cand = 0;  // candidate bits of the held samples that no I pass has taken yet
    int q = 0, idx = 0, good = 0;
    uint32_t minDesc = maxDesc, minSum = maxColor, colorThr = 0, descThr = 0;
    int cur[C];
    unsigned intra[C];
    uint32_t nb[C][8];
    uint32_t curm[C];  // the current colour, channel c alone in byte c of a word (BGR); the gray value (gray)
    uint32_t iy = 0, iz = 0;  // intra descriptors as the records hold them: d0 | d1 << 16, d2
    // Records (ss_rec): the first batch sample-major, the rest PIXEL-major: a lane holds four samples on its pixel's first trip and
    // eight (128 contiguous bytes; 32 for gray) on every later one - memory requests that are all payload.  With one 16-byte record
    // per request the kernel ran at the L1's limit of outstanding misses (rocprofv3: TCP_PENDING_STALL 84 % of the cycles, 94 M line
    // requests of which a quarter of every 64-byte sector was used, 10.6 GB fetched per launch on 8 x 1080p); the sample ORDER is unchanged.
    constexpr int B = kSsBatch;
    size_t rec = 0, rnext = 0;  // first record of the samples this lane holds; the pixel's first pixel-major record
    SsSample<C> bt[B], nbt[B];
#pragma unroll
    for (int j = 0; j < B; ++j) bt[j] = SsSample<C>{}, nbt[j] = SsSample<C>{};
#pragma unroll
    for (int c = 0; c < C; ++c) cur[c] = 0, intra[c] = 0, curm[c] = 0;
#ifdef BGS_SS_STATS
    unsigned st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
      const unsigned long long idle = __ballot(!active);
      const int nidle = __popcll(idle);
      if (qempty && nidle == kWave) break;
      SS_STAT(0, 1), SS_STAT(1, kWave - nidle);  // trips, active lanes
      if (!qempty && nidle >= a.refill) {  // wave-uniform
        SS_STAT(2, 1);
        const int leader = __ffsll((long long)idle) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&qhead, (unsigned)nidle);
        base = (unsigned)__shfl((int)base, leader);
        if (base + (unsigned)nidle >= (unsigned)APIX) qempty = true;
        if (!active) {
          const unsigned my = base + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
          if (my < (unsigned)APIX && ctx[my][2] != kSsNotInterior) {
            q = (int)my;
            const int lx = q % kSsTW, ly = q / kSsTW;
            const uint32_t c0 = ctx[q][0], c1 = ctx[q][1];
            descThr = ctx[q][2], colorThr = c1 >> 16;
            intra[0] = c0 & 0xffffu;
            if constexpr (C == 3) intra[1] = c0 >> 16, intra[2] = c1 & 0xffffu;
            iy = c0, iz = c1 & 0xffffu;
            gather(ly, lx, cur, nb);
#pragma unroll
            for (int c = 0; c < C; ++c) curm[c] = (uint32_t)cur[c] << (8 * c);
            if constexpr (QUEUE) {  // the neighbours as bytes for whoever computes this pixel's inter-LBSP tests
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // words 0, 1: neighbours 0-3, 4-7 (the high halves of nb[4i .. 4i+3]); words 2, 3: neighbours 8-15 (the low halves)
                  const uint32_t h01 = __builtin_amdgcn_perm(nb[c][4 * i + 1], nb[c][4 * i], 0x0c0c0602u), h23 = __builtin_amdgcn_perm(nb[c][4 * i + 3], nb[c][4 * i + 2], 0x0c0c0602u);
                  const uint32_t l01 = __builtin_amdgcn_perm(nb[c][4 * i + 1], nb[c][4 * i], 0x0c0c0400u), l23 = __builtin_amdgcn_perm(nb[c][4 * i + 3], nb[c][4 * i + 2], 0x0c0c0400u);
                  w[i] = h01 | (h23 << 16), w[2 + i] = l01 | (l23 << 16);
                }
                *reinterpret_cast<uint4*>(&nbs[threadIdx.x][4 * c]) = make_uint4(w[0], w[1], w[2], w[3]);
              }
              hcur[threadIdx.x] = curm[0] | curm[1] | curm[2];
            }
            const size_t p = (size_t)(y0 + ly) * a.cols + (x0 + lx);
            rec = ss_rec(a, stream, N, p, 0), rnext = ss_rec(a, stream, N, p, B);
#pragma unroll
            for (int j = 0; j < B; ++j) bt[j] = SsSample<C>::load(a.samples, rec + j * N);
            idx = 0, good = 0, minDesc = maxDesc, minSum = maxColor;
            active = true, fresh = true;
          }
        }
      }
      // One trip = the samples a lane holds: batch 0 (four samples, bt) on a pixel's first trip, then EIGHT at a time (bt + nbt,
      // 128 contiguous bytes of the pixel-major part).  Two passes.  R: the cheap exact rejection tests on all held samples,
      // every lane busy -> candidate bits.  I: the inter-LBSP test (~150 instructions per channel set), one candidate per lane and
      // pass in sample order, stopping at the nReq-th match exactly like the reference's loop (:469).  Rejected samples and samples
      // behind the nReq-th match have no side effects, and min() does not care about order.
      // (round 4) A lane whose candidates were left over by the I passes below keeps its batch and its candidate bits (`fresh` false):
      // it neither loads nor re-tests anything until a pass has taken them.
      {
        const unsigned long long fm = __ballot(active && fresh), wm = __ballot(active && fresh && idx > 0);
        SS_STAT(3, fm != 0), SS_STAT(4, __popcll(fm)), SS_STAT(5, wm != 0), SS_STAT(6, __popcll(wm));
        (void)fm, (void)wm;
      }
      if (active && fresh) {  // :469-497 (BGR) / :334-357 (gray); an active lane always has good < nReq and idx < nS here
        wide = idx > 0;
        auto reject_bits = [&](const SsSample<C>(&bb)[B], int first) -> uint32_t {
          uint32_t bits = 0;
          if constexpr (C == 1) {
#pragma unroll
            for (int j = 0; j < B; ++j) bits |= (uint32_t)(first + j < a.nS && (uint32_t)abs(cur[0] - bb[j].color(0)) <= colorThr) << j;
          } else {
            // The reference walks the channels in order and drops the sample at the first failed test (:474-491); a sample is kept
            // only if every test passes, so the tests may run in any order.  Exact rejections first, from what costs least:
            // lower bounds with the intra half of the descriptor distance alone (dd >= intraD/2, and sd grows with dd).  The
            // per-channel colour test cd <= scColorThr is implied by the one on the bound lbsd = min(255, k + cd): either
            // lbsd = k + cd >= cd, or lbsd = 255 <= scColorThr and cd <= 255.
            const uint32_t totColorThr = colorThr * 3, totDescThr = descThr * 3, scColorThr = totColorThr / 2;
#pragma unroll
            for (int j = 0; j < B; ++j) {
              const uint32_t sx = bb[j].v.x, sy = bb[j].v.y, sz = bb[j].v.z;
              uint32_t cd[3], lbdd[3], lbsd[3];
              cd[0] = __builtin_amdgcn_sad_u8(curm[0], sx & 0x0000ffu, 0u);
              cd[1] = __builtin_amdgcn_sad_u8(curm[1], sx & 0x00ff00u, 0u);
              cd[2] = __builtin_amdgcn_sad_u8(curm[2], sx & 0xff0000u, 0u);
              const uint32_t xy = iy ^ sy;
              lbdd[0] = (uint32_t)__popc(xy & 0xffffu) >> 1, lbdd[1] = (uint32_t)__popc(xy >> 16) >> 1, lbdd[2] = (uint32_t)__popc(iz ^ sz) >> 1;
#pragma unroll
              for (int c = 0; c < 3; ++c) lbsd[c] = min((lbdd[c] >> 1) * (255 / 16) + cd[c], 255u);
              const bool ok = max(max(lbsd[0], lbsd[1]), lbsd[2]) <= scColorThr && lbdd[0] + lbdd[1] + lbdd[2] <= totDescThr && lbsd[0] + lbsd[1] + lbsd[2] <= totColorThr;
              bits |= (uint32_t)(ok && first + j < a.nS) << j;
            }
          }
          return bits;
        };
        cand = reject_bits(bt, idx);
        if (__any(wide)) {
          const uint32_t hi = reject_bits(nbt, idx + B);
          if (wide) cand |= hi << B;
        }
        fresh = false;
      }
      if constexpr (QUEUE) {
        // ROUNDS (round 4, BGS_SS_QUEUE=1; NOT the default: measured slower).  One candidate per lane and pass leaves the wave at
        // ~25 % lane use on a young model (profiles/r04_subsense_phase_a_pmc.txt: 2.65 M passes of ~230 vector instructions per
        // 8 x 1080p launch for 43 M candidates, half of the kernel's vector instructions).  Here every lane that has candidates puts some
        // on its wave's list and ALL 64 lanes work the list off, 64 entries at a time, each lane computing whichever candidate falls to
        // it from what its owner left in LDS (neighbours, colour, thresholds) and the sample re-read from memory (a cache hit: the
        // owner's load).  A pixel past its first batch enqueues up to four candidates at once - speculation: a candidate behind the
        // nReq-th match is computed in vain, but such pixels rarely match at all - a pixel on its first batch only as many as it still
        // needs (at most two).  Results come back as 16 bits per candidate; the owner applies them in sample order and stops at its
        // nReq-th match, exactly the reference's early exit.  Same integers as the direct form below (parity-tested).
        // What the counters said: the vector instruction count did not move (1 184 M -> 1 181 M per launch) and 3.9 M list passes ran
        // where 2.65 M direct ones had - a wave holds only ~20 candidates per trip (a third of its lanes are between pixels or have
        // none), so a list pass is still two thirds empty and pays ~60 instructions of LDS traffic on top; LDS bank conflicts x 5.
        // Young 8 x 1080p phase A 2.04 -> 2.42 ms, aged 1.06 -> 1.49 ms.  Filling the lanes needs candidates from SEVERAL trips of a
        // lane in flight at once, i.e. speculating across sample batches - not built.
        const int wbase = (int)(threadIdx.x & ~(kWave - 1));
        uint16_t* wl = qlist + (threadIdx.x / kWave) * kSsQList;
        const size_t recBase = (size_t)stream * N * (size_t)a.nSpad;
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (active) hrec[threadIdx.x] = (uint32_t)(rec - recBase), hq[threadIdx.x] = (uint16_t)((unsigned)q | (wide ? 0x8000u : 0u));
        for (;;) {
          const bool want = active && cand != 0 && good < a.nReq;
          if (!__any(want)) break;
          uint32_t enq = 0;
          if (want) {
            const int quota = wide ? 4 : min(a.nReq - good, 2);
            uint32_t x = cand;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k < quota && x) enq |= x & (0u - x), x &= x - 1u;
            cand &= ~enq;
          }
          unsigned n_list = 0;  // wave-uniform: at most 4 x 64 = kSsQList entries
#pragma unroll
          for (int b = 0; b < 2 * B; ++b) {
            const bool has = (enq >> b) & 1u;
            const unsigned long long mb = __ballot(has);
            if (has) wl[n_list + (unsigned)__popcll(mb & lt)] = (uint16_t)((lane << 3) | b);
            n_list += (unsigned)__popcll(mb);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          for (unsigned base = 0; base < n_list; base += kWave) {
            const bool valid = base + (unsigned)lane < n_list;
            const unsigned ent = valid ? (unsigned)wl[base + lane] : ((unsigned)lane << 3);
            const int ot = wbase + (int)(ent >> 3), j = (int)(ent & 7u);
            const unsigned oqw = hq[ot];
            const int oq = (int)(oqw & 0x7fffu);
            const size_t srec = recBase + hrec[ot] + ((oqw >> 15) ? (size_t)j : (size_t)j * N);
            const SsSample<3> sv = SsSample<3>::load(a.samples, valid ? srec : recBase);  // (unconditional, see the prefetch note: a valid address either way)
            const uint32_t c0w = ctx[oq][0], c1w = ctx[oq][1], oDescThr = ctx[oq][2], oColorThr = c1w >> 16;
            const uint32_t curw = hcur[ot];
            const uint32_t totColorThr = oColorThr * 3, totDescThr = oDescThr * 3, scColorThr = totColorThr / 2;
            uint32_t totDesc = 0, totSum = 0;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const uint4 w = *reinterpret_cast<const uint4*>(&nbs[ot][4 * c]);
              uint32_t nbf[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                nbf[k] = __builtin_amdgcn_perm(w.x, w.z, 0x0c000c00u | ((uint32_t)(4 + k) << 16) | (uint32_t)k);
                nbf[4 + k] = __builtin_amdgcn_perm(w.y, w.w, 0x0c000c00u | ((uint32_t)(4 + k) << 16) | (uint32_t)k);
              }
              const int cc = (int)((curw >> (8 * c)) & 0xffu), bcc = sv.color(c);
              const uint32_t ic = c == 0 ? (c0w & 0xffffu) : c == 1 ? (c0w >> 16) : (c1w & 0xffffu);
              const uint32_t cd = (uint32_t)abs(cc - bcc);
              const uint32_t intraD = (uint32_t)__popc(ic ^ sv.desc(c));
              const unsigned inter = ss_lbsp(nbf, bcc, lut[bcc]);
              const uint32_t interD = (uint32_t)__popc(inter ^ sv.desc(c));
              const uint32_t dd = (intraD + interD) / 2;
              uint32_t sd = (dd / 2) * (255 / 16) + cd;
              sd = sd < 255 ? sd : 255;
              ok = ok && sd <= scColorThr;
              totDesc += dd, totSum += sd;
            }
            const bool m = ok && !(totDesc > totDescThr || totSum > totColorThr);
            if (valid) qres[ot * 8 + j] = (uint16_t)(m ? (totDesc | (totSum << 6)) : 0xffffu);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (enq) {
            const uint4 rr = *reinterpret_cast<const uint4*>(&qres[threadIdx.x * 8]);
            const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
            for (int b = 0; b < 2 * B; ++b) {
              const uint32_t r = (rw[b >> 1] >> (16 * (b & 1))) & 0xffffu;
              if (((enq >> b) & 1u) && good < a.nReq && r != 0xffffu) {
                const uint32_t mDesc = r & 0x3fu, mSum = r >> 6;
                minDesc = minDesc > mDesc ? mDesc : minDesc;
                minSum = minSum > mSum ? mSum : minSum;
                good++;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();  // (the next round rewrites the list and the results)
        }
      }
      // I passes: one candidate per lane and pass.  (Round 4 tried to fill the idle lanes of a sparse pass by splitting each candidate
      // over three lanes, one channel each, the helpers reading the owner's neighbours from the frame tile: parity-green, but a
      // sub-pass of 21 candidates cost ~150 vector + 25 LDS instructions against ~230 for a whole pass, and 40 % of the passes hold 22-42
      // candidates, i.e. need two: -4 % vector instructions, +90 M bank-conflict cycles, no time gained - profiles/r04_subsense_phase_a_pmc.txt.)
      if constexpr (!QUEUE) {
        for (;;) {
          const bool go = active && cand != 0 && good < a.nReq;
          const unsigned long long gm = __ballot(go);
          if (!gm) break;
          // DENSE PASSES (round 4): a pass costs the same ~230 vector instructions for one candidate as for 64, and once the lanes
          // with two or three candidates in their batch are the only ones left it ran at a few lanes.  So a pass that fewer than
          // ipassMin lanes would join is put off while some other active lane has something else to do (its next batch, or finishing
          // and handing its lane to a new pixel): the waiting lanes keep batch and candidate bits and join the passes of the next trip.
          // Every trip still makes progress - either a pass runs, or a lane without candidates moves on - and each pixel still sees
          // its samples in order up to its nReq-th match: the same integers for any ipassMin (1 = every pass at once, the round-3 form).
          // Not once the tile's queue is empty: no new pixel can join then, a trip is a memory round trip, and waiting only strings
          // more of them together (measured: the aged model, whose pixels mostly take one trip, lost 13 % without this condition).
          if (!qempty && __popcll(gm) < a.ipassMin && __any(active && !go)) {
            SS_STAT(9, 1), SS_STAT(10, __popcll(gm));
            break;
          }
          SS_STAT(7, 1), SS_STAT(8, __popcll(gm));
          SsSample<C> smp = bt[0];
          if (go) {
            const int j = __ffs((int)cand) - 1;
            cand &= cand - 1;
  #pragma unroll
            for (int jj = 1; jj < B; ++jj)
              if (j == jj) smp = bt[jj];
  #pragma unroll
            for (int jj = 0; jj < B; ++jj)
              if (j == B + jj) smp = nbt[jj];
          }
          bool matched = false;
          uint32_t mDesc = 0, mSum = 0;
          if (go) {
            if constexpr (C == 1) {
              const int bcc = smp.color(0);
              const unsigned bdc = smp.desc(0);
              const uint32_t cd = (uint32_t)abs(cur[0] - bcc);
              const uint32_t intraD = (uint32_t)__popc(intra[0] ^ bdc);
              const unsigned inter = ss_lbsp(nb[0], bcc, lut[bcc]);
              const uint32_t dd = (intraD + (uint32_t)__popc(inter ^ bdc)) / 2;
              if (dd <= descThr) {
                uint32_t sd = (dd / 4) * (255 / 16) + cd;
                sd = sd < 255 ? sd : 255;
                if (sd <= colorThr) matched = true, mDesc = dd, mSum = sd;
              }
            } else {
              const uint32_t totColorThr = colorThr * 3, totDescThr = descThr * 3, scColorThr = totColorThr / 2;
              uint32_t totDesc = 0, totSum = 0;
              bool ok = true;
  #pragma unroll
              for (int c = 0; c < 3; ++c) {
                const int bcc = smp.color(c);
                const uint32_t cd = (uint32_t)abs(cur[c] - bcc);
                const uint32_t intraD = (uint32_t)__popc(intra[c] ^ smp.desc(c));
                const unsigned inter = ss_lbsp(nb[c], bcc, lut[bcc]);
                const uint32_t interD = (uint32_t)__popc(inter ^ smp.desc(c));
                const uint32_t dd = (intraD + interD) / 2;
                uint32_t sd = (dd / 2) * (255 / 16) + cd;
                sd = sd < 255 ? sd : 255;
                ok = ok && sd <= scColorThr;
                totDesc += dd, totSum += sd;
              }
              if (ok && !(totDesc > totDescThr || totSum > totColorThr)) matched = true, mDesc = totDesc, mSum = totSum;
            }
          }
          if (go && matched) {
            minDesc = minDesc > mDesc ? mDesc : minDesc;
            minSum = minSum > mSum ? mSum : minSum;
            good++;
          }
        }
      }
      if (active && !(cand != 0 && good < a.nReq)) {  // (a lane with candidates left waits for the next trip's passes)
        idx += wide ? 2 * B : B;
        if (good < a.nReq && idx < a.nS) {  // not done: the next eight samples (the first trip leaves the sample-major part here)
          rec = wide ? rec + 2 * B : rnext;
#pragma unroll
          for (int j = 0; j < B; ++j) bt[j] = SsSample<C>::load(a.samples, rec + j), nbt[j] = SsSample<C>::load(a.samples, rec + B + j);
          fresh = true;
        } else {
          ctx[q][2] = (uint32_t)good | (minDesc << 8) | (minSum << 16);
          active = false;
        }
      }
    }
#ifdef BGS_SS_STATS
    if (lane == 0)
      for (int i = 0; i < 12; ++i) atomicAdd(&g_ss_stats[i], (unsigned long long)st[i]);
#endif
  }
  __syncthreads();

  if constexpr (SPLIT) {
    // ---- stage 3: the outcome of the loop goes to memory: the raw segmentation (the post-processing chain starts from it) and the
    // hand-over word of ss_feedback_kernel, which applies :498-576 beside that chain
  #pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int qq = r * kBlock + threadIdx.x, lx = qq % kSsTW, ly = qq / kSsTW;
      const int x = x0 + lx, y = y0 + ly;
      if (x < a.cols && y < a.rows) {
        const size_t i = sN + (size_t)y * a.cols + x;
        const uint32_t res = ctx[qq][2];
        const bool in = interior_of(x, y);
        a.raw[i] = (in && (int)(res & 0xffu) < a.nReq) ? 255 : 0;
        if (in) a.ho[i * 2] = res;
      }
    }
  } else {
    // ---- stage 3: everything after the loop, :498-582
    const SsScalars sc = a.sc[stream];
    auto at = [&](int ry, int rx, int c) -> int {
      const uint8_t* rowp = reinterpret_cast<const uint8_t*>(tile[ry]);
      const int shift = (int)(((long)(y0 + ry - 2) * a.cols * C + rb) & 3L);
      return rowp[shift + rx * C + c];
    };
  #pragma unroll 2
    for (int r = 0; r < PPL; ++r) {
      const int qq = r * kBlock + threadIdx.x, lx = qq % kSsTW, ly = qq / kSsTW;
      const int x = x0 + lx, y = y0 + ly;
      if (interior_of(x, y)) {
        const size_t p = (size_t)y * a.cols + x, i = sN + p;
        const uint32_t fr = a.frameIndex, pi = (uint32_t)p;
        float Rv = a.R[i], Vv = a.V[i], Tv = a.T[i];
        const float rawLT_old = a.RawLT[i], rawST_old = a.RawSTOld[i], finLT = a.FinLT[i], finST = a.FinST[i];
        const float dlast_old = a.DlastOld[i], dminLT_old = a.DminLT[i], dminST_old = a.DminST[i];
        const int lastfg = a.lastFG[i], blink = a.blinks[i];
        int lastc[C], cur[C];
        unsigned lastd[C], intra[C];
  #pragma unroll
        for (int c = 0; c < C; ++c) lastc[c] = a.lastColor[i * C + c], lastd[c] = a.lastDesc[i * C + c], cur[c] = at(ly + 2, lx + 2, c);
        const int unst = (Rv > 3.0f || (rawLT_old - finLT) > 0.1f || (rawST_old - finST) > 0.1f) ? 1 : 0;  // :467
        // the random neighbour of the background branch (:526-551) depends only on `unst`
        const bool use3 = a.use3x3 && !unst;
        int xn, yn;
        {
          const uint32_t r4 = ss_rand(fr, pi, 4);
          if (use3) {
            const int rr = (int)(r4 % 8u);
            xn = x + kSsN3[rr][0], yn = y + kSsN3[rr][1];
          } else {
            const int rr = (int)(r4 % 24u);
            xn = x + kSsN5[rr][0], yn = y + kSsN5[rr][1];
          }
          xn = min(max(xn, 2), a.cols - 3), yn = min(max(yn, 2), a.rows - 3);
        }
        const size_t j = sN + (size_t)yn * a.cols + xn;
        const float nbrLast = a.DlastOld[j], nbrRaw = a.RawSTOld[j];  // previous frame's copy (contract)
        const uint32_t c0 = ctx[qq][0], c1 = ctx[qq][1], res = ctx[qq][2];
        intra[0] = c0 & 0xffffu;
        if constexpr (C == 3) intra[1] = c0 >> 16, intra[2] = c1 & 0xffffu;
        const int good = (int)(res & 0xffu);
        const uint32_t minDesc = (res >> 8) & 0xffu, minSum = res >> 16;
        uint32_t l1 = 0, hd = 0;
  #pragma unroll
        for (int c = 0; c < C; ++c) {
          l1 += (uint32_t)abs(lastc[c] - cur[c]);
          hd += (uint32_t)__popc((lastd[c] ^ intra[c]) & 0xffffu);
        }
        const float fLT = a.fLT, fST = a.fST;
        const float normLast = ((float)l1 / maxColor + (float)hd / maxDesc) / 2;  // :498
        a.DlastNew[i] = dlast_old * (1.0f - fST) + normLast * fST;
        a.unstable[i] = (uint8_t)unst;
        float dminLT = dminLT_old, dminST = dminST_old, rawLT = rawLT_old, rawST = rawST_old;
        bool isfg;
        uint16_t reqSelf = 0, reqNbr = 0;
        if (good < a.nReq) {  // foreground :500-515
          float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2 + (float)(a.nReq - good) / a.nReq;
          nm = nm > 1.0f ? 1.0f : nm;
          dminLT = dminLT * (1.0f - fLT) + nm * fLT;
          dminST = dminST * (1.0f - fST) + nm * fST;
          rawLT = rawLT * (1.0f - fLT) + fLT;
          rawST = rawST * (1.0f - fST) + fST;
          isfg = true;
          if (sc.cooldown && (ss_rand(fr, pi, 0) % 2u) == 0) reqSelf = ss_req(ss_mod(ss_rand(fr, pi, 1), (uint32_t)a.nS, magic), 12);
        } else {  // background :516-552
          const float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2;
          dminLT = dminLT * (1.0f - fLT) + nm * fLT;
          dminST = dminST * (1.0f - fST) + nm * fST;
          rawLT = rawLT * (1.0f - fLT);
          rawST = rawST * (1.0f - fST);
          isfg = false;
          const uint32_t lr = (uint32_t)ceilf(Tv);  // (the reference computes these in size_t; every value fits 31 bits)
          if (ss_mod(ss_rand(fr, pi, 2), lr, magic) == 0) reqSelf = ss_req(ss_mod(ss_rand(fr, pi, 3), (uint32_t)a.nS, magic), 12);
          const uint32_t nrand = ss_rand(fr, pi, 5);
          if (ss_mod(nrand, use3 ? lr : (lr / 2 + 1), magic) == 0 || (nbrRaw > 0.995f && nbrLast < 0.010f && ss_mod(nrand, (uint32_t)sc.capLo, magic) == 0))
            reqNbr = ss_req(ss_mod(ss_rand(fr, pi, 6), (uint32_t)a.nS, magic), (yn - y + 2) * 5 + (xn - x + 2));
        }
        a.DminLT[i] = dminLT, a.DminST[i] = dminST, a.RawLT[i] = rawLT, a.RawSTNew[i] = rawST;
        a.raw[i] = isfg ? 255 : 0;
        a.req[i * 2] = reqSelf, a.req[i * 2 + 1] = reqNbr;
        // Round 3: the self update is stored HERE.  No other pixel reads this pixel's samples in phase A (every pixel tests its own
        // model only, and this lane is done with it), so the write cannot be seen early; a diffusion request of another source for
        // the same slot is ordered against it in phase B exactly as before (the request stays in a.req: an earlier source's loses
        // there, a later source's is applied after this launch and wins).  It takes half of the scattered 16-byte writes out of
        // phase B, whose only limit they are, into a kernel that is bound by vector issue.
        if (a.selfInA && reqSelf) {
          unsigned dsc[C];
  #pragma unroll
          for (int c = 0; c < C; ++c) dsc[c] = intra[c];
          SsSample<C>::make(cur, dsc).store(a.samples, ss_rec(a, stream, N, p, (int)ss_req_slot(reqSelf)));
        }
        // feedback :553-576
        const float dmin_min = dminLT < dminST ? dminLT : dminST, dmin_max = dminLT > dminST ? dminLT : dminST;
        if (lastfg || (dmin_min < 0.1f && isfg)) {
          if (Tv < sc.capHi) Tv += div_rn(0.5f, dmin_max * Vv);
        } else if (Tv > sc.capLo)
          Tv -= div_rn(0.25f * Vv, dmin_max);
        if (Tv < sc.capLo)
          Tv = sc.capLo;
        else if (Tv > sc.capHi)
          Tv = sc.capHi;
        if (dmin_max > 0.1f && blink)
          Vv += 1.0f;
        else if (Vv > 0.1f) {
          Vv -= lastfg ? 0.1f / 4 : unst ? 0.1f / 2 : 0.1f;
          if (Vv < 0.1f) Vv = 0.1f;
        }
        const float pw = 1.0f + dmin_min * 2;
        if ((double)Rv < __dmul_rn((double)pw, (double)pw))  // std::pow(float, int) is a double in C++11; the square is exact
          Rv += 0.01f * (Vv - 0.1f);
        else {
          Rv -= div_rn(0.01f, Vv);
          if (Rv < 1.0f) Rv = 1.0f;
        }
        a.R[i] = Rv, a.V[i] = Vv, a.T[i] = Tv;
        if constexpr (C == 3)
          nzcount += (__popc(intra[0]) + __popc(intra[1]) + __popc(intra[2])) >= 4;  // :577-578
        else
          nzcount += __popc(intra[0]) >= 2;  // :430-431
  #pragma unroll
        for (int c = 0; c < C; ++c) {  // :579-582
          a.lastDesc[i * C + c] = (uint16_t)intra[c];
          a.lastColor[i * C + c] = (uint8_t)cur[c];
        }
      } else if (x < a.cols && y < a.rows) {
        const size_t i = sN + (size_t)y * a.cols + x;
        a.raw[i] = 0;
        a.req[i * 2] = 0, a.req[i * 2 + 1] = 0;
        a.DlastNew[i] = a.DlastOld[i], a.RawSTNew[i] = a.RawSTOld[i];
      }
    }
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) nzcount += __shfl_xor((int)nzcount, o);
  if ((threadIdx.x & (kWave - 1)) == 0 && nzcount) atomicAdd(&nz_block, nzcount);
  __syncthreads();
  if (threadIdx.x == 0 && nz_block) atomicAdd(&a.sc[stream].nzCount, nz_block);
}

// ----------------------------------------------------------------------------------------------- feedback (:498-576), round 4
// Everything BackgroundSubtractorSuBSENSE::operator() does to a pixel after its sample-consensus loop that is not the frame itself:
// rolling means of the distances and of the raw segmentation (:498-522), the update decisions (:508-551: which of its own samples, which
// neighbour's), the T / v / R feedback (:553-576).  Rounds 1-3 ran it as stage 3 of phase A, where it was a third of a kernel that is
// bound by vector issue (~700 of ~2 200 lane-instructions per quiet pixel: seven counter hashes, five run-time `%`, up to eight
// correctly rounded divisions).  Nothing in it needs the frame tile or phase A's registers, so it is a pointwise kernel of its own now,
// launched on the side stream in front of phase B - beside the post-processing chain, which only needs `raw`.  What phase A hands over
// is 8 bytes per pixel (`ho`):
//   word 0  good | minDesc << 8 | minSum << 16         the outcome of the loop (:469-497)
//   word 1  l1 | hd << 10 | unstable << 16 | lastFG != 0 << 17 | blink != 0 << 18
//           l1 / hd: L1 colour distance and Hamming descriptor distance to the pixel's last frame (:498); the three flags are this
//           frame's inputs of the rules, captured before the chain rewrites the maps they come from (m_oLastFGMask, m_oBlinksFrame;
//           `unstable` needs the final-segmentation means that ss_finish_kernel updates)
// and the per-stream scalars as they stood at the start of the frame (scSnap: the frame-level block rewrites `sc` beside this kernel).
// The statements, their order and every operand are those of stage 3 before: same floats, same draws.  x % d with d in 1..1023 is exact
// through ss_mod.  grid: (ceil(cols / 256), rows, streams).
template <int C>
__global__ __launch_bounds__(kBlock) void ss_feedback_kernel(const SsArgs a) {
  constexpr uint32_t maxColor = 255 * C, maxDesc = 16 * C;
  __shared__ uint32_t magic[kSsMagicN];
  for (int k = threadIdx.x; k < kSsMagicN; k += kBlock) magic[k] = a.magic[k];
  __syncthreads();
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
  if (x >= a.cols) return;
  const size_t p = (size_t)y * a.cols + x, i = sN + p;
  if (!(x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2)) {  // LBSP::validateROI: border pixels are never touched
    a.req[i * 2] = 0, a.req[i * 2 + 1] = 0;
    a.DlastNew[i] = a.DlastOld[i], a.RawSTNew[i] = a.RawSTOld[i];
    return;
  }
  const SsScalars sc = a.scSnap[stream];
  const uint32_t fr = a.frameIndex, pi = (uint32_t)p;
  const uint2 hw = *reinterpret_cast<const uint2*>(a.ho + i * 2);
  float Rv = a.R[i], Vv = a.V[i], Tv = a.T[i];
  const float rawLT_old = a.RawLT[i], rawST_old = a.RawSTOld[i];
  const float dlast_old = a.DlastOld[i], dminLT_old = a.DminLT[i], dminST_old = a.DminST[i];
  const int good = (int)(hw.x & 0xffu);
  const uint32_t minDesc = (hw.x >> 8) & 0xffu, minSum = hw.x >> 16;
  const uint32_t l1 = hw.y & 0x3ffu, hd = (hw.y >> 10) & 0x3fu;
  const int unst = (int)((hw.y >> 16) & 1u), lastfg = (int)((hw.y >> 17) & 1u), blink = (int)((hw.y >> 18) & 1u);
  // the random neighbour of the background branch (:526-551) depends only on `unst`
  const bool use3 = a.use3x3 && !unst;
  int xn, yn;
  {
    const uint32_t r4 = ss_rand(fr, pi, 4);
    if (use3) {
      const int rr = (int)(r4 % 8u);
      xn = x + kSsN3[rr][0], yn = y + kSsN3[rr][1];
    } else {
      const int rr = (int)(r4 % 24u);
      xn = x + kSsN5[rr][0], yn = y + kSsN5[rr][1];
    }
    xn = min(max(xn, 2), a.cols - 3), yn = min(max(yn, 2), a.rows - 3);
  }
  const size_t j = sN + (size_t)yn * a.cols + xn;
  const float nbrLast = a.DlastOld[j], nbrRaw = a.RawSTOld[j];  // previous frame's copy (contract)
  const float fLT = a.fLT, fST = a.fST;
  const float normLast = ((float)l1 / maxColor + (float)hd / maxDesc) / 2;  // :498
  a.DlastNew[i] = dlast_old * (1.0f - fST) + normLast * fST;
  float dminLT = dminLT_old, dminST = dminST_old, rawLT = rawLT_old, rawST = rawST_old;
  bool isfg;
  uint16_t reqSelf = 0, reqNbr = 0;
  if (good < a.nReq) {  // foreground :500-515
    float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2 + (float)(a.nReq - good) / a.nReq;
    nm = nm > 1.0f ? 1.0f : nm;
    dminLT = dminLT * (1.0f - fLT) + nm * fLT;
    dminST = dminST * (1.0f - fST) + nm * fST;
    rawLT = rawLT * (1.0f - fLT) + fLT;
    rawST = rawST * (1.0f - fST) + fST;
    isfg = true;
    if (sc.cooldown && (ss_rand(fr, pi, 0) % 2u) == 0) reqSelf = ss_req(ss_mod(ss_rand(fr, pi, 1), (uint32_t)a.nS, magic), 12);
  } else {  // background :516-552
    const float nm = ((float)minSum / maxColor + (float)minDesc / maxDesc) / 2;
    dminLT = dminLT * (1.0f - fLT) + nm * fLT;
    dminST = dminST * (1.0f - fST) + nm * fST;
    rawLT = rawLT * (1.0f - fLT);
    rawST = rawST * (1.0f - fST);
    isfg = false;
    const uint32_t lr = (uint32_t)ceilf(Tv);  // (the reference computes these in size_t; every value fits 31 bits)
    if (ss_mod(ss_rand(fr, pi, 2), lr, magic) == 0) reqSelf = ss_req(ss_mod(ss_rand(fr, pi, 3), (uint32_t)a.nS, magic), 12);
    const uint32_t nrand = ss_rand(fr, pi, 5);
    if (ss_mod(nrand, use3 ? lr : (lr / 2 + 1), magic) == 0 || (nbrRaw > 0.995f && nbrLast < 0.010f && ss_mod(nrand, (uint32_t)sc.capLo, magic) == 0))
      reqNbr = ss_req(ss_mod(ss_rand(fr, pi, 6), (uint32_t)a.nS, magic), (yn - y + 2) * 5 + (xn - x + 2));
  }
  a.DminLT[i] = dminLT, a.DminST[i] = dminST, a.RawLT[i] = rawLT, a.RawSTNew[i] = rawST;
  *reinterpret_cast<uint32_t*>(a.req + i * 2) = (uint32_t)reqSelf | ((uint32_t)reqNbr << 16);
  // The SELF update is stored here (round 3: by phase A).  Nobody reads a pixel's samples between its consensus loop and phase B, so
  // the write cannot be seen early; a diffusion request of another source for the same slot is ordered against it in phase B exactly
  // as before (the request stays in a.req: an earlier source's loses there, a later source's is applied by phase B, after this
  // launch, and wins).  It keeps half of the scattered 16-byte writes out of phase B, whose only limit they are.
  if (a.selfInA && reqSelf) {
    int cur[C];
    unsigned dsc[C];
    if constexpr (C == 3) {  // what phase A left as the pixel's last colour / descriptor IS this frame's
      const uint32_t cw = *reinterpret_cast<const uint32_t*>(a.lastColor + i * 3);
      const uint2 dw = *reinterpret_cast<const uint2*>(a.lastDesc + i * 3);
      cur[0] = (int)(cw & 0xffu), cur[1] = (int)((cw >> 8) & 0xffu), cur[2] = (int)((cw >> 16) & 0xffu);
      dsc[0] = dw.x & 0xffffu, dsc[1] = dw.x >> 16, dsc[2] = dw.y & 0xffffu;
    } else {
      cur[0] = a.lastColor[i], dsc[0] = a.lastDesc[i];
    }
    SsSample<C>::make(cur, dsc).store(a.samples, ss_rec(a, stream, N, p, (int)ss_req_slot(reqSelf)));
  }
  // feedback :553-576
  const float dmin_min = dminLT < dminST ? dminLT : dminST, dmin_max = dminLT > dminST ? dminLT : dminST;
  if (lastfg || (dmin_min < 0.1f && isfg)) {
    if (Tv < sc.capHi) Tv += div_rn(0.5f, dmin_max * Vv);
  } else if (Tv > sc.capLo)
    Tv -= div_rn(0.25f * Vv, dmin_max);
  if (Tv < sc.capLo)
    Tv = sc.capLo;
  else if (Tv > sc.capHi)
    Tv = sc.capHi;
  if (dmin_max > 0.1f && blink)
    Vv += 1.0f;
  else if (Vv > 0.1f) {
    Vv -= lastfg ? 0.1f / 4 : unst ? 0.1f / 2 : 0.1f;
    if (Vv < 0.1f) Vv = 0.1f;
  }
  const float pw = 1.0f + dmin_min * 2;
  if ((double)Rv < __dmul_rn((double)pw, (double)pw))  // std::pow(float, int) is a double in C++11; the square is exact
    Rv += 0.01f * (Vv - 0.1f);
  else {
    Rv -= div_rn(0.01f, Vv);
    if (Rv < 1.0f) Rv = 1.0f;
  }
  a.R[i] = Rv, a.V[i] = Vv, a.T[i] = Tv;
}

// ----------------------------------------------------------------------------------------------- LOBSTER, phase A
// BackgroundSubtractorLOBSTER::operator() (package_bgs/pl/BackgroundSubtractorLOBSTER.cpp:172-284): the sample-consensus
// test of SuBSENSE with fixed thresholds and no feedback maps; update requests go through the same phase B.
// a.nMinColor = nColorDistThreshold, a.nDescOff = nDescDistThreshold; a.lastColor / a.lastDesc point at SCRATCH planes here
// (what a requesting pixel will write: its current colour and intra descriptor) - LOBSTER's own last-frame images are
// only read by refreshModel.  Learning rate = BGSLOBSTER_DEFAULT_LEARNING_RATE (LOBSTER.cpp:36 passes none).
template <int C>
__global__ __launch_bounds__(kBlock) void lob_phase_a_kernel(const SsArgs a) {
  constexpr int HW = kSsTW + 4, HH = kSsTH + 4;
  constexpr int ROWB = (HW * C + 3 + 3) / 4 * 4;
  constexpr uint32_t kLearningRate = 16;
  __shared__ uint32_t tile[HH][ROWB / 4];
  __shared__ uint8_t lut[256];
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const uint8_t* img = a.frame + (size_t)blockIdx.z * N * C;
  const int x0 = blockIdx.x * kSsTW, y0 = blockIdx.y * kSsTH;
  const long imgsz = (long)N * C, rb = (long)(x0 - 2) * C;
  for (int i = threadIdx.x; i < HH * (ROWB / 4); i += kBlock) {
    const int ry = i / (ROWB / 4), rd = i - ry * (ROWB / 4);
    const int y = min(max(y0 + ry - 2, 0), a.rows - 1);
    const long off = (((long)y * a.cols * C + rb) & ~3L) + 4L * rd;
    uint32_t v = 0;
    if (off >= 0 && off + 4 <= imgsz)
      v = *reinterpret_cast<const uint32_t*>(img + off);
    else if (off < imgsz && off + 4 > 0)
      for (int b = 0; b < 4; ++b)
        if (off + b >= 0 && off + b < imgsz) v |= (uint32_t)img[off + b] << (8 * b);
    tile[ry][rd] = v;
  }
  lut[threadIdx.x] = a.lut[(size_t)stream * 256 + threadIdx.x];
  __syncthreads();
  const int lx = threadIdx.x % kSsTW, ly = threadIdx.x / kSsTW;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= a.cols || y >= a.rows) return;
  const size_t p = (size_t)y * a.cols + x, i = sN + p;
  if (!(x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2)) {  // outside LBSP::validateROI: never foreground, never updated
    a.raw[i] = 0;
    a.req[i * 2] = 0, a.req[i * 2 + 1] = 0;
    return;
  }
  int cur[C];
  uint32_t nb[C][8];
  {
    LbspWin<C> win;
    win.load(&tile[0][0], ROWB / 4, ly, lx, (int)(((long)(y0 - 2) * a.cols * C + rb) & 3L), (a.cols * C) & 3);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      cur[c] = win.centre(c);
      win.pack(c, nb[c]);
    }
  }
  const uint32_t colorThr = (uint32_t)a.nMinColor, descThr = (uint32_t)a.nDescOff;
  const uint32_t descThr3 = descThr * 3, colorThr3 = colorThr * 3, scDesc = descThr3 / 2, scColor = colorThr3 / 2;  // :225-228
  size_t rec = ss_rec(a, stream, N, p, 0);
  const size_t rstep = a.pixelMajor ? 1 : N;
  int good = 0, idx = 0;
  int bc[C];
  unsigned bd[C];
  SsSample<C> smp = SsSample<C>::load(a.samples, rec), nsmp = smp;
  while (good < a.nReq && idx < a.nS) {  // :192-205 (gray) / :241-258 (BGR); sample idx+1 is in flight while idx is tested (the loop is latency-bound)
    rec += (idx + 1 < a.nS) ? rstep : 0;  // unconditional, see phase A: the loaded registers must BE nsmp for the load to stay in flight
    nsmp = SsSample<C>::load(a.samples, rec);
#pragma unroll
    for (int c = 0; c < C; ++c) bc[c] = smp.color(c), bd[c] = smp.desc(c);
    if constexpr (C == 1) {
      const int bcc = bc[0];
      const uint32_t cd = (uint32_t)abs(cur[0] - bcc);
      if (cd <= colorThr / 2) {
        const unsigned inter = ss_lbsp(nb[0], bcc, lut[bcc]);
        if ((uint32_t)__popc(inter ^ bd[0]) <= descThr) good++;
      }
    } else {
      uint32_t totC = 0, totD = 0;
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (ok) {
          const int bcc = bc[c];
          const uint32_t cd = (uint32_t)abs(cur[c] - bcc);
          if (cd > scColor) {
            ok = false;
          } else {
            const unsigned inter = ss_lbsp(nb[c], bcc, lut[bcc]);
            const uint32_t dd = (uint32_t)__popc(inter ^ bd[c]);
            if (dd > scDesc)
              ok = false;
            else
              totC += cd, totD += dd;
          }
        }
      if (ok && totD <= descThr3 && totC <= colorThr3) good++;
    }
    idx++;
    ss_wait_here(nsmp);
    smp = nsmp;
  }
  uint16_t reqSelf = 0, reqNbr = 0;
  if (good >= a.nReq) {
    const uint32_t fr = a.frameIndex, pi = (uint32_t)p;
    if ((ss_rand(fr, pi, 0) % kLearningRate) == 0) reqSelf = ss_req(ss_rand(fr, pi, 1) % (uint32_t)a.nS, 12);  // :209-214 / :262-269
    if ((ss_rand(fr, pi, 2) % kLearningRate) == 0) {                                                             // :215-222 / :270-279
      const int r = (int)(ss_rand(fr, pi, 3) % 8u);
      const int xn = min(max(x + kSsN3[r][0], 2), a.cols - 3), yn = min(max(y + kSsN3[r][1], 2), a.rows - 3);
      reqNbr = ss_req(ss_rand(fr, pi, 4) % (uint32_t)a.nS, (yn - y + 2) * 5 + (xn - x + 2));
    }
    if (reqSelf | reqNbr) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        a.lastColor[i * C + c] = (uint8_t)cur[c];
        a.lastDesc[i * C + c] = (uint16_t)ss_lbsp(nb[c], cur[c], lut[cur[c]]);
      }
    }
  }
  a.raw[i] = good < a.nReq ? 255 : 0;  // :207 / :260
  a.req[i * 2] = reqSelf, a.req[i * 2 + 1] = reqNbr;
}

// Round 4: LOBSTER's phase A with the LANES FED FROM A QUEUE (the form SuBSENSE's phase A took in round 2).  lob_phase_a_kernel above
// gives every lane one pixel and walks the samples in lock step: a wave runs as many trips as its slowest pixel needs, and on a scene
// with a few percent of foreground nearly every wave holds a pixel that matches nothing and walks all 35 samples (8 x 1080p, S_surv:
// 2.36 - 2.53 ms although the average pixel tests six).  Here a workgroup owns a 64 x 16 tile (four pixels per lane), a lane that has
// finished its pixel takes the next one from the tile's queue, and a trip tests one sample per active lane: the cheap part (colour
// distances, exact: a sample matches only if EVERY test of :192-205 / :241-258 passes, so their order is free) for everybody, the
// inter-LBSP part only when some lane's sample got that far.  What a pixel leaves behind is its match count in LDS; the update
// requests and the colour / descriptor a requesting pixel hands to phase B are made afterwards, densely, for the whole tile.
// Same results as lob_phase_a_kernel (BGS_LOB_QUEUE=0), whole-model parity tests.
constexpr int kLobATH = 16;
template <int C>
__global__ __launch_bounds__(kBlock) void lob_phase_a_queue_kernel(const SsArgs a) {
  constexpr int ATH = kLobATH, APIX = kSsTW * ATH, PPL = APIX / kBlock;
  constexpr int HW = kSsTW + 4, HH = ATH + 4;
  constexpr int ROWB = (HW * C + 3 + 3) / 4 * 4;
  constexpr uint32_t kLearningRate = 16;
  __shared__ uint32_t tile[HH][ROWB / 4];
  __shared__ uint8_t lut[256];
  __shared__ uint8_t res[APIX];  // matches found for the pixel (stage 2 -> stage 3)
  __shared__ unsigned qhead;
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const uint8_t* img = a.frame + (size_t)blockIdx.z * N * C;
  const int x0 = blockIdx.x * kSsTW, y0 = blockIdx.y * ATH;
  const long imgsz = (long)N * C, rb = (long)(x0 - 2) * C;
  for (int i = threadIdx.x; i < HH * (ROWB / 4); i += kBlock) {
    const int ry = i / (ROWB / 4), rd = i - ry * (ROWB / 4);
    const int y = min(max(y0 + ry - 2, 0), a.rows - 1);
    const long off = (((long)y * a.cols * C + rb) & ~3L) + 4L * rd;
    uint32_t v = 0;
    if (off >= 0 && off + 4 <= imgsz)
      v = *reinterpret_cast<const uint32_t*>(img + off);
    else if (off < imgsz && off + 4 > 0)
      for (int b = 0; b < 4; ++b)
        if (off + b >= 0 && off + b < imgsz) v |= (uint32_t)img[off + b] << (8 * b);
    tile[ry][rd] = v;
  }
  lut[threadIdx.x] = a.lut[(size_t)stream * 256 + threadIdx.x];
  if (threadIdx.x == 0) qhead = 0;
  __syncthreads();
  const int rowShift0 = (int)(((long)(y0 - 2) * a.cols * C + rb) & 3L), rowShiftStep = (a.cols * C) & 3;
  auto gather = [&](int ly, int lx, int (&cur)[C], uint32_t (&nb)[C][8]) {
    LbspWin<C> win;
    win.load(&tile[0][0], ROWB / 4, ly, lx, rowShift0, rowShiftStep);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      cur[c] = win.centre(c);
      win.pack(c, nb[c]);
    }
  };
  auto interior_of = [&](int x, int y) { return x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2; };  // LBSP::validateROI
  const uint32_t colorThr = (uint32_t)a.nMinColor, descThr = (uint32_t)a.nDescOff;
  const uint32_t descThr3 = descThr * 3, colorThr3 = colorThr * 3, scDesc = descThr3 / 2, scColor = colorThr3 / 2;  // :225-228
  const size_t rstep = a.pixelMajor ? 1 : N;

  // ---- stage 2: the sample loop (:192-205 gray / :241-258 BGR), lanes fed from the queue
  {
    const int lane = threadIdx.x & (kWave - 1);
    bool active = false, qempty = false;
    int q = 0, idx = 0, good = 0;
    int cur[C];
    uint32_t nb[C][8];
    uint32_t curm[C];
    size_t rec = ss_rec(a, stream, N, 0, 0);  // (a valid record for lanes that never get a pixel: their loads are unconditional)
    SsSample<C> smp = SsSample<C>{}, nsmp = SsSample<C>{};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      cur[c] = 0, curm[c] = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) nb[c][k] = 0;
    }
    for (;;) {
      const unsigned long long idle = __ballot(!active);
      const int nidle = __popcll(idle);
      if (qempty && nidle == kWave) break;
      if (!qempty && nidle >= a.refill) {  // wave-uniform
        const int leader = __ffsll((long long)idle) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&qhead, (unsigned)nidle);
        base = (unsigned)__shfl((int)base, leader);
        if (base + (unsigned)nidle >= (unsigned)APIX) qempty = true;
        if (!active) {
          const unsigned my = base + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
          const int lx = (int)(my % kSsTW), ly = (int)(my / kSsTW);
          if (my < (unsigned)APIX && interior_of(x0 + lx, y0 + ly)) {
            q = (int)my;
            gather(ly, lx, cur, nb);
#pragma unroll
            for (int c = 0; c < C; ++c) curm[c] = (uint32_t)cur[c] << (8 * c);
            rec = ss_rec(a, stream, N, (size_t)(y0 + ly) * a.cols + (x0 + lx), 0);
            smp = SsSample<C>::load(a.samples, rec);
            idx = 0, good = 0;
            active = true;
          }
        }
      }
      // sample idx + 1 is in flight while idx is tested (the loaded registers must BE nsmp for the load to stay in flight)
      rec += (active && idx + 1 < a.nS) ? rstep : 0;
      nsmp = SsSample<C>::load(a.samples, rec);
      bool cand = false;
      if (active) {
        if constexpr (C == 1) {
          cand = (uint32_t)abs(cur[0] - smp.color(0)) <= colorThr / 2;
        } else {
          const uint32_t sx = smp.v.x;
          const uint32_t c0 = __builtin_amdgcn_sad_u8(curm[0], sx & 0x0000ffu, 0u), c1 = __builtin_amdgcn_sad_u8(curm[1], sx & 0x00ff00u, 0u),
                         c2 = __builtin_amdgcn_sad_u8(curm[2], sx & 0xff0000u, 0u);
          cand = max(max(c0, c1), c2) <= scColor && c0 + c1 + c2 <= colorThr3;
        }
      }
      if (__any(cand)) {  // wave-uniform: the expensive part only when some lane's sample got this far
        if (cand) {
          if constexpr (C == 1) {
            const int bcc = smp.color(0);
            const unsigned inter = ss_lbsp(nb[0], bcc, lut[bcc]);
            if ((uint32_t)__popc(inter ^ smp.desc(0)) <= descThr) good++;
          } else {
            uint32_t totD = 0;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const int bcc = smp.color(c);
              const unsigned inter = ss_lbsp(nb[c], bcc, lut[bcc]);
              const uint32_t dd = (uint32_t)__popc(inter ^ smp.desc(c));
              ok = ok && dd <= scDesc;
              totD += dd;
            }
            if (ok && totD <= descThr3) good++;
          }
        }
      }
      if (active) {
        idx++;
        if (good >= a.nReq || idx >= a.nS) {
          res[q] = (uint8_t)good;
          active = false;
        }
      }
      ss_wait_here(nsmp);
      smp = nsmp;
    }
  }
  __syncthreads();

  // ---- stage 3: the segmentation and the update requests (:207-222 / :260-279), every pixel of the tile
#pragma unroll 1
  for (int r = 0; r < PPL; ++r) {
    const int qq = r * kBlock + threadIdx.x, lx = qq % kSsTW, ly = qq / kSsTW;
    const int x = x0 + lx, y = y0 + ly;
    const bool inimg = x < a.cols && y < a.rows, in = inimg && interior_of(x, y);
    const size_t p = (size_t)(inimg ? y : 0) * a.cols + (inimg ? x : 0), i = sN + p;
    uint16_t reqSelf = 0, reqNbr = 0;
    const int good = in ? (int)res[qq] : 0;
    if (in && good >= a.nReq) {
      const uint32_t fr = a.frameIndex, pi = (uint32_t)p;
      if ((ss_rand(fr, pi, 0) % kLearningRate) == 0) reqSelf = ss_req(ss_rand(fr, pi, 1) % (uint32_t)a.nS, 12);  // :209-214 / :262-269
      if ((ss_rand(fr, pi, 2) % kLearningRate) == 0) {                                                             // :215-222 / :270-279
        const int rr = (int)(ss_rand(fr, pi, 3) % 8u);
        const int xn = min(max(x + kSsN3[rr][0], 2), a.cols - 3), yn = min(max(y + kSsN3[rr][1], 2), a.rows - 3);
        reqNbr = ss_req(ss_rand(fr, pi, 4) % (uint32_t)a.nS, (yn - y + 2) * 5 + (xn - x + 2));
      }
    }
    const bool need = (reqSelf | reqNbr) != 0;
    if (__any(need)) {  // what a requesting pixel will write: its current colour and intra descriptor
      int cur[C];
      uint32_t nb[C][8];
      gather(min(ly, ATH - 1), lx, cur, nb);
      if (need) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          a.lastColor[i * C + c] = (uint8_t)cur[c];
          a.lastDesc[i * C + c] = (uint16_t)ss_lbsp(nb[c], cur[c], lut[cur[c]]);
        }
      }
    }
    if (inimg) {
      a.raw[i] = (in && good < a.nReq) ? 255 : 0;  // :207 / :260; outside LBSP::validateROI: never foreground, never updated
      a.req[i * 2] = reqSelf, a.req[i * 2 + 1] = reqNbr;
    }
  }
}

// ----------------------------------------------------------------------------------------------- phase B
// Applies the sample writes phase A decided, in the reference's order: sources in raster order, a source's self update before its
// neighbour diffusion, a later write to the same (pixel, sample slot) replacing an earlier one.
// Round 1 PULLED: every pixel looked at the requests of the 25 sources around it (50 LDS reads and tests per pixel, 0.47 ms on
// 8 x 1080p - the second-largest kernel of the step - for writes that a few percent of the pixels make).  Now the requests are
// PUSHED: a workgroup owns a 64 x 16 tile of TARGETS, collects the requests of the tile + halo whose target lies inside the tile into
// a list in LDS, and one lane per list entry decides whether its request is the last one in that order for its (target, slot) - it
// looks at the up to 50 requests that could aim at the same target - and if so performs the write.  Same result, the work follows
// the number of requests instead of the number of pixels.
// Round 3 (profiles/r03_subsense_phase_b_pmc.txt: 0.7 ms on the aged model, where half of the pixels make a request; only 0.5 + 0.5 GB
// of HBM traffic, 400 lane-instructions per pixel, waves waiting 65 % of their cycles): the colour / descriptor of a source is loaded
// by the lane that performs its write (one 4-byte and one 8-byte load) instead of being staged in LDS behind the request load it
// depended on; list slots are handed out per wave (ballot + one LDS atomic) instead of per request; every target counts the requests
// aimed at it while the list is built, and only the requests of targets with more than one - a quarter of them on the aged model -
// go through the search for a later request to the same slot (a second, dense list); away from the image border that search tests
// 25 instead of 50 requests (a source's self update aims at itself, its diffusion at one of the 24 pixels around it; the other
// combinations only arise when a target is clamped into the image).
constexpr int kSsBTH = 16;

// is request `key` (slot `slot`, target (tly, tlx) of the halo'd tile) the last one in the reference's order for that target and slot?
template <int HW, bool INTERIOR>
__device__ __forceinline__ bool ss_req_is_last(const uint32_t (*rq)[HW], int tly, int tlx, uint32_t slot, int key) {
  bool last = true;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int sy = tly + dy, sx = tlx + dx;  // inside the halo'd tile: the target is inside the tile
      const uint32_t both = rq[sy][sx];
      const uint32_t aimed = SS_REQ_VALID | (slot << 5) | (uint32_t)(12 - 5 * dy - dx);  // a request of (sy, sx) for this target and slot
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        if (INTERIOR && qq != ((dy == 0 && dx == 0) ? 0 : 1)) continue;
        if (((both >> (16 * qq)) & 0xffffu) == aimed && (sy * HW + sx) * 2 + qq > key) last = false;
      }
    }
  return last;
}

// append to a list in LDS, one atomic per wave; every lane of the wave must make the call
template <typename T>
__device__ __forceinline__ void ss_list_push(T* list, unsigned* n, bool push, T value, int lane) {
  const unsigned long long mask = __ballot(push);
  if (mask) {  // (wave-uniform)
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(n, (unsigned)__popcll(mask));
    base = (unsigned)__shfl((int)base, 0, kWave);
    if (push) list[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = value;
  }
}

template <int C>
__global__ __launch_bounds__(kBlock) void ss_phase_b_kernel(const SsArgs a) {
  constexpr int HW = kSsTW + 4, HH = kSsBTH + 4;
  static_assert(kSsTW * kSsBTH / 4 == kBlock, "one dword of target counters per lane");
  __shared__ uint32_t rq[HH][HW];          // both requests of a source pixel in one dword
  __shared__ uint16_t list[HH * HW * 2];   // ly << 8 | lx << 1 | q of every request whose target is in this tile
  __shared__ uint16_t list2[HH * HW * 2];  // those whose target has other requests too
  __shared__ uint32_t cnt[kBlock];         // requests aimed at each target of the tile, one byte each (<= 26)
  __shared__ unsigned nlist, nlist2;
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const int x0 = blockIdx.x * kSsTW, y0 = blockIdx.y * kSsBTH;
  const int lane = threadIdx.x & (kWave - 1);
  if (threadIdx.x == 0) nlist = 0, nlist2 = 0;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  // Round 4: a workgroup's requests are loaded in ONE go (the loop below used to load its dword at the top of each of its six trips and
  // then ballot on it: six memory round trips in a row) and the lists hold 16-bit entries (28 -> 17 KB of LDS: 8 instead of 5
  // workgroups per CU).  Neither moved the launch time on the aged model (0.37 - 0.41 ms on 8 x 1080p): it is the ~8 M scattered
  // 16-byte writes, i.e. DRAM row activations, that take the time, not the waves' latency chains.
  constexpr int NT = (HH * HW + kBlock - 1) / kBlock;
  uint32_t vv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = t * kBlock + (int)threadIdx.x;
    const int ly = i / HW, lx = i - ly * HW;
    const int y = y0 + ly - 2, x = x0 + lx - 2;
    vv[t] = 0;
    if (i < HH * HW && y >= 2 && y < a.rows - 2 && x >= 2 && x < a.cols - 2) vv[t] = *reinterpret_cast<const uint32_t*>(a.req + (sN + (size_t)y * a.cols + x) * 2);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {  // (every lane makes every trip: the ballots see whole waves)
    const int i = t * kBlock + (int)threadIdx.x;
    const int ly = i / HW, lx = i - ly * HW;
    const uint32_t v = vv[t];
    if (i < HH * HW) rq[ly][lx] = v;
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // self request first, then the neighbour request
      const uint32_t r = (v >> (16 * q)) & 0xffffu;
      const int code = (int)(r & 0x1fu), tly = ly + code / 5 - 2, tlx = lx + code % 5 - 2;  // code = (dy + 2) * 5 + (dx + 2) of the target
      const bool aimed = (r & SS_REQ_VALID) && tly >= 2 && tly < 2 + kSsBTH && tlx >= 2 && tlx < 2 + kSsTW;
      if (aimed) {  // (self updates count too: they compete for their slot even when phase A has already stored them)
        const int t = (tly - 2) * kSsTW + (tlx - 2);
        atomicAdd(&cnt[t >> 2], 1u << (8 * (t & 3)));
      }
      const bool push = aimed && !(q == 0 && a.selfInA);
      ss_list_push(list, &nlist, push, (uint16_t)((ly << 8) | (lx << 1) | q), lane);
    }
  }
  __syncthreads();
  auto write = [&](int ly, int lx, int tly, int tlx, uint32_t slot) {
    // what a requesting source writes: its current colour / intra descriptor (phase A left them in lastColor / lastDesc)
    const size_t src = sN + (size_t)(y0 + ly - 2) * a.cols + (size_t)(x0 + lx - 2);
    int col[C];
    unsigned dsc[C];
    if constexpr (C == 3) {  // one 4-byte and one 8-byte load (no alignment needed; 8 spare bytes behind both maps: ss_refresh_one)
      typedef uint32_t __attribute__((aligned(1))) u32u;
      typedef uint64_t __attribute__((aligned(2))) u64u;
      const uint32_t cw = *reinterpret_cast<const u32u*>(a.lastColor + src * 3);
      const uint64_t dw = *reinterpret_cast<const u64u*>(a.lastDesc + src * 3);
      col[0] = (int)(cw & 0xffu), col[1] = (int)((cw >> 8) & 0xffu), col[2] = (int)((cw >> 16) & 0xffu);
      dsc[0] = (unsigned)(dw & 0xffffu), dsc[1] = (unsigned)((dw >> 16) & 0xffffu), dsc[2] = (unsigned)((dw >> 32) & 0xffffu);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) col[c] = a.lastColor[src * C + c], dsc[c] = a.lastDesc[src * C + c];
    }
    const size_t p = (size_t)(y0 + tly - 2) * a.cols + (size_t)(x0 + tlx - 2);
    SsSample<C>::make(col, dsc).store(a.samples, ss_rec(a, stream, N, p, (int)slot));  // one 16-byte (4-byte) store per update
  };
  const unsigned n = nlist;
  for (unsigned e0 = 0; e0 < n; e0 += kBlock) {  // the only request of its target: write; else: second list
    const unsigned e = e0 + threadIdx.x;
    bool contested = false;
    uint16_t ent = 0;
    if (e < n) {
      ent = list[e];
      const int ly = (int)(ent >> 8), lx = (int)((ent >> 1) & 0x7fu), q = (int)(ent & 1u);
      const uint32_t r = (rq[ly][lx] >> (16 * q)) & 0xffffu;
      const int code = (int)(r & 0x1fu), tly = ly + code / 5 - 2, tlx = lx + code % 5 - 2;
      const int t = (tly - 2) * kSsTW + (tlx - 2);
      contested = ((cnt[t >> 2] >> (8 * (t & 3))) & 0xffu) > 1u;
      if (!contested) write(ly, lx, tly, tlx, ss_req_slot(r));
    }
    ss_list_push(list2, &nlist2, contested, ent, lane);
  }
  __syncthreads();
  const unsigned n2 = nlist2;
  // no source of this tile (+ halo) had a diffusion target clamped into the image: x +- 2, y +- 2 of every source lie in [2, cols - 3] x [2, rows - 3]
  const bool interior = x0 - 4 >= 2 && x0 + kSsTW + 3 <= a.cols - 3 && y0 - 4 >= 2 && y0 + kSsBTH + 3 <= a.rows - 3;
  for (unsigned e = threadIdx.x; e < n2; e += kBlock) {
    const unsigned ent = list2[e];
    const int ly = (int)(ent >> 8), lx = (int)((ent >> 1) & 0x7fu), q = (int)(ent & 1u);
    const uint32_t r = (rq[ly][lx] >> (16 * q)) & 0xffffu;
    const int code = (int)(r & 0x1fu), tly = ly + code / 5 - 2, tlx = lx + code % 5 - 2;
    const uint32_t slot = ss_req_slot(r);
    const int key = (ly * HW + lx) * 2 + q;  // position in the reference's order of writes
    const bool last = interior ? ss_req_is_last<HW, true>(rq, tly, tlx, slot, key) : ss_req_is_last<HW, false>(rq, tly, tlx, slot, key);
    if (last) write(ly, lx, tly, tlx, slot);
  }
}

// ----------------------------------------------------------------------------------------------- refreshModel :249-291
// getRandSamplePosition's scan of the 7x7 pattern (RandUtils.h:28-48) is a pure function of r = 1 + rand % 512: tabulated at
// compile time, entry r - 1 = (x_sample << 4) | y_sample.
struct SsPosTab {
  uint8_t v[512];
};
constexpr SsPosTab ss_make_pos_tab() {
  constexpr int P[7][7] = {{2, 4, 6, 7, 6, 4, 2},     {4, 8, 12, 14, 12, 8, 4},  {6, 12, 21, 25, 21, 12, 6}, {7, 14, 25, 28, 25, 14, 7},
                           {6, 12, 21, 25, 21, 12, 6}, {4, 8, 12, 14, 12, 8, 4}, {2, 4, 6, 7, 6, 4, 2}};  // = kSsPattern
  SsPosTab t{};
  for (int e = 0; e < 512; ++e) {
    int r = 1 + e, xs = 0, ys = 0;
    bool stop = false;
    for (xs = 0; xs < 7 && !stop; ++xs)
      for (ys = 0; ys < 7; ++ys) {
        r -= P[ys][xs];
        if (r <= 0) {
          stop = true;
          break;
        }
      }
    if (stop) --xs;  // the goto leaves x_sample un-incremented
    t.v[e] = (uint8_t)((xs << 4) | ys);
  }
  return t;
}
__device__ __constant__ const SsPosTab kSsPosTab = ss_make_pos_tab();

// one refreshed sample: slot (start + m) % nS of pixel p takes the colour / descriptor of a random 7x7 neighbour (if that one is background)
template <int C>
__device__ __forceinline__ void ss_refresh_one(const SsArgs& a, int stream, size_t N, size_t sN, uint32_t p, int x, int y, int m, int start) {
  const int t = kSsPosTab.v[ss_rand(a.frameIndex, p, 16u + (uint32_t)m) % 512u];
  const int xs = min(max((t >> 4) + x - 3, 2), a.cols - 3), ys = min(max((t & 15) + y - 3, 2), a.rows - 3);
  const size_t j = sN + (size_t)ys * a.cols + xs;
  if (!a.lastFG[j]) {
    int col[C];
    unsigned dsc[C];
    if constexpr (C == 3) {
      // the neighbour's 3 colour bytes and 3 descriptor words as ONE 4-byte and ONE 8-byte load (global loads need no alignment on
      // this hardware; the engine allocates 8 spare bytes behind both maps): 3 instead of 7 scattered loads per sample
      typedef uint32_t __attribute__((aligned(1))) u32u;
      typedef uint64_t __attribute__((aligned(2))) u64u;
      const uint32_t cw = *reinterpret_cast<const u32u*>(a.lastColor + j * 3);
      const uint64_t dw = *reinterpret_cast<const u64u*>(a.lastDesc + j * 3);
      col[0] = (int)(cw & 0xffu), col[1] = (int)((cw >> 8) & 0xffu), col[2] = (int)((cw >> 16) & 0xffu);
      dsc[0] = (unsigned)(dw & 0xffffu), dsc[1] = (unsigned)((dw >> 16) & 0xffffu), dsc[2] = (unsigned)((dw >> 32) & 0xffffu);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) col[c] = a.lastColor[j * C + c], dsc[c] = a.lastDesc[j * C + c];
    }
    int slot = start + m;
    slot = slot >= a.nS ? slot - a.nS : slot;
    SsSample<C>::make(col, dsc).store(a.samples, ss_rec(a, stream, N, p, slot));
  }
}

// The full refresh reads, per sample, a RANDOM neighbour's foreground byte, colour and descriptors - three arrays, three scattered
// loads; its counters (round 3) put it at the vector memory path's rate for scattered loads (13 per wave).  One
// streaming pass packs them first: colour | (lastFG != 0) << 24, d0 | d1 << 16, d2 - the sample record itself with the flag in its
// spare byte - so that a sample costs ONE scattered 16-byte load (round 4).
__global__ __launch_bounds__(kBlock) void ss_lastrec_pack_kernel(const SsArgs a) {
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= N) return;
  typedef uint32_t __attribute__((aligned(1))) u32u;
  typedef uint64_t __attribute__((aligned(2))) u64u;
  const uint32_t cw = *reinterpret_cast<const u32u*>(a.lastColor + (sN + p) * 3);
  const uint64_t dw = *reinterpret_cast<const u64u*>(a.lastDesc + (sN + p) * 3);
  a.lastRec[sN + p] = make_uint4((cw & 0x00ffffffu) | (a.lastFG[sN + p] ? 0x01000000u : 0u), (uint32_t)dw, (uint32_t)(dw >> 32) & 0xffffu, 0u);
}

// mode 0: unconditional full refresh (initialisation, frac = 1); mode 1: 10 % refresh if the frame-level block asked for it.
// Full refresh of SuBSENSE's layout (FAST): a workgroup owns 16 pixels, lane (pixel q, column s) of the 16 x 16 writes plane s of the
// sample-major first batch (s < 4) and the pixel-major records 4 + s, 4 + s + 16, ...: every store instruction of a wave is four
// runs of 256 contiguous bytes (round 2: one lane walked a pixel's 50 samples serially, each store 768 bytes from its neighbour's,
// each sample through a <= 49-step scan of the pattern: 26 ms for 8 x 1080p, 0.06 of what the 13 GB it writes need).
// Otherwise (10 % refresh: 5 samples; LOBSTER's sample-major planes): one lane per pixel.
constexpr int kSsRefreshGroups = 8;
template <int C, bool FAST>
__global__ __launch_bounds__(kBlock) void ss_refresh_kernel(const SsArgs a, int mode) {
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols, sN = (size_t)stream * N;
  if (mode == 1 && !a.sc[stream].doRefresh) return;
  if constexpr (!FAST) {  // a bounded grid walks the pixels (the per-frame launch that usually finds doRefresh == 0 stays small)
    for (uint32_t p = blockIdx.x * (uint32_t)kBlock + threadIdx.x; p < N; p += gridDim.x * (uint32_t)kBlock) {
      if (mode == 1) a.T[sN + p] = 1.0f;  // m_oUpdateRateFrame = cv::Scalar(1.0f), every pixel (:682)
      const int x = (int)(p % (uint32_t)a.cols), y = (int)(p / (uint32_t)a.cols);
      if (!(x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2)) continue;
      if (a.lastFG[sN + p]) continue;  // bForceFGUpdate = false
      const int nRefresh = mode == 1 ? (int)(0.1f * a.nS) : a.nS;
      const int start = mode == 1 ? (int)(ss_rand(a.frameIndex, 0xFFFFFFFFu, 0) % (uint32_t)a.nS) : 0;
      for (int m = 0; m < nRefresh; ++m) ss_refresh_one<C>(a, stream, N, sN, p, x, y, m, start);
    }
    return;
  }
  // the position table in LDS: 64 lanes with 64 different table indices are one LDS access there, and one more scattered global load
  // on the vector memory path otherwise - the kernel's limit (17 scattered loads per wave: counters in DESIGN.md 7c)
  __shared__ uint8_t tab[512];
  tab[threadIdx.x] = kSsPosTab.v[threadIdx.x], tab[threadIdx.x + 256] = kSsPosTab.v[threadIdx.x + 256];
  __syncthreads();
  // (a workgroup walks kSsRefreshGroups groups of 16 pixels: with one group per workgroup the 1 M tiny workgroups of an 8 x 1080p
  // launch kept only 1.6 waves per SIMD resident - SQ_WAVE_CYCLES / duration - and the kernel ran at 2 TB/s of its stores)
  for (int grp = 0; grp < kSsRefreshGroups; ++grp) {
  const uint32_t p = (blockIdx.x * (uint32_t)kSsRefreshGroups + (uint32_t)grp) * 16u + (threadIdx.x >> 4);
  if (p >= N) continue;
  if (mode == 1) a.T[sN + p] = 1.0f;  // m_oUpdateRateFrame = cv::Scalar(1.0f), every pixel (:682)
  const int x = (int)(p % (uint32_t)a.cols), y = (int)(p / (uint32_t)a.cols);
  // BGR, mode 0 (initialisation): this kernel writes EVERY record of the stream - zeros where no sample goes (the two border rows and
  // columns, the padding slots behind sample nS - 1, a sample whose neighbour is foreground) - so that the engine need not clear the
  // 13 GB of an 8 x 1080p model first (round 4: that hipMemsetAsync alone took as long as two thirds of this kernel)
  const bool fill = C == 3 && mode == 0;
  const bool live = x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2 && !a.lastFG[sN + p];  // LBSP::validateROI; bForceFGUpdate = false
  if (!live && !fill) continue;
  if constexpr (FAST) {
    // A lane's samples are s (first batch, s < 4), 4 + s, 20 + s, 36 + s, ...: the first four go through three STAGES - position
    // table, the neighbours' foreground bytes, the neighbours' colour / descriptor - each stage's loads issued together (one sample
    // after the other was three dependent memory round trips per sample, ~10 in a row per lane: the kernel was latency-bound at 0.3
    // of the rate its 13.8 GB of stores need).  Loads are unconditional; a sample that does not exist or whose neighbour is
    // foreground reads the pixel's own entries and stores nothing.
    const int s = threadIdx.x & 15;
    int mm[4], jx[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mm[k] = k == 0 ? s : kSsBatch + s + 16 * (k - 1);
      ok[k] = k == 0 ? s < kSsBatch : mm[k] < a.nS;
      jx[k] = tab[ss_rand(a.frameIndex, p, 16u + (uint32_t)(ok[k] ? mm[k] : 0)) % 512u];
    }
    // per-stream bases once, 32-bit offsets per sample (N < 2^31), the record assembled from the loaded words as they are (a
    // record is the colour dword with byte 3 cleared, d0 | d1 << 16, d2): ~60 instead of ~170 instructions per sample
    uint32_t jo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xs = min(max((jx[k] >> 4) + x - 3, 2), a.cols - 3), ys = min(max((jx[k] & 15) + y - 3, 2), a.rows - 3);
      jo[k] = (uint32_t)ys * (uint32_t)a.cols + (uint32_t)xs;
    }
    if constexpr (C == 3) {
      // ONE scattered 16-byte load per sample (ss_lastrec_pack_kernel): the neighbour's record with its foreground flag in byte 3
      const uint4* lr = a.lastRec + sN;
      uint4 rv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) rv[k] = lr[jo[k]];
      // ss_rec of this pixel's records: the first batch sample-major (plane s at + s * N), the rest pixel-major
      uint4* recs = reinterpret_cast<uint4*>(a.samples) + (size_t)stream * N * (size_t)a.nSpad;
      uint4* mine = recs + (size_t)kSsBatch * N + (size_t)p * (size_t)(a.nSpad - kSsBatch) - kSsBatch;  // + m for m >= kSsBatch
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool put = live && ok[k] && !(rv[k].x >> 24);
        if (put || (fill && (k == 0 ? s < kSsBatch : mm[k] < a.nSpad))) {
          const uint4 v = put ? make_uint4(rv[k].x & 0x00ffffffu, rv[k].y, rv[k].z, 0u) : make_uint4(0u, 0u, 0u, 0u);
          if (k == 0)
            recs[(size_t)s * N + p] = v;
          else
            mine[mm[k]] = v;
        }
      }
      if (fill) {  // (more than 52 samples: the loop below writes the samples, this one the zeros it leaves out)
        for (int m = kSsBatch + s + 48; m < a.nSpad; m += 16)
          if (!live || m >= a.nS) mine[m] = make_uint4(0u, 0u, 0u, 0u);
      }
      if (!live) continue;
    } else {
      const uint8_t* fgS = a.lastFG + sN;
      (void)fgS;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) ss_refresh_one<C>(a, stream, N, sN, p, x, y, mm[k], 0);
    }
    for (int m = kSsBatch + s + 48; m < a.nS; m += 16) ss_refresh_one<C>(a, stream, N, sN, p, x, y, m, 0);  // (more than 52 samples)
  }
  }
}

// LastColor of the first frame: interior pixels only (:229-243); lastDesc comes from lbsp_kernel
template <int C>
__global__ __launch_bounds__(kBlock) void ss_init_lastcolor_kernel(const SsArgs a) {
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols;
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= N) return;
  const int x = (int)(p % a.cols), y = (int)(p / a.cols);
  const bool in = x >= 2 && x < a.cols - 2 && y >= 2 && y < a.rows - 2;
#pragma unroll
  for (int c = 0; c < C; ++c) a.lastColor[((size_t)stream * N + p) * C + c] = in ? a.frame[((size_t)blockIdx.z * N + p) * C + c] : 0;
}

// ----------------------------------------------------------------------------------------------- post-processing :624-642
// blink maps :624-627
template <int G>  // pixels per lane: 16 (one dwordx4 per map) when the launch allows it, else 1
__global__ __launch_bounds__(kBlock) void ss_blink_kernel(const SsArgs a, size_t count) {
  const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * G;
  if (i >= count) return;
  const size_t g = (size_t)a.first * a.rows * a.cols + i;
  if constexpr (G == 16) {
    const uint4 raw = *reinterpret_cast<const uint4*>(a.raw + g), last = *reinterpret_cast<const uint4*>(a.lastRaw + g);
    const uint4 lb = *reinterpret_cast<const uint4*>(a.lastRawBlink + g);
    const uint4 blink = make_uint4(raw.x ^ last.x, raw.y ^ last.y, raw.z ^ last.z, raw.w ^ last.w);
    *reinterpret_cast<uint4*>(a.blinks + g) = make_uint4(blink.x | lb.x, blink.y | lb.y, blink.z | lb.z, blink.w | lb.w);
    *reinterpret_cast<uint4*>(a.lastRawBlink + g) = blink;
    *reinterpret_cast<uint4*>(a.lastRaw + g) = raw;
  } else {
    const uint8_t raw = a.raw[g], blink = raw ^ a.lastRaw[g];
    a.blinks[g] = blink | a.lastRawBlink[g];
    a.lastRawBlink[g] = blink;
    a.lastRaw[g] = raw;
  }
}

// Per-stream constants of a freshly constructed model, handed over BY VALUE in the kernel arguments (no host buffer has to
// outlive the launch, so the first frame needs no stream synchronisation): the 256-entry LBSP threshold LUT and the
// frame-level scalars.  One workgroup per stream of the launch.
struct SsLut256 {
  uint8_t v[256];
};
__global__ __launch_bounds__(kBlock) void ss_init_consts_kernel(uint8_t* lut_all, SsScalars* sc_all, const SsLut256 lut, const SsScalars sc0, int first) {
  const int stream = first + blockIdx.x;
  lut_all[(size_t)stream * 256 + threadIdx.x] = lut.v[threadIdx.x];
  if (sc_all && threadIdx.x == 0) sc_all[stream] = sc0;
}

// flood fill from (0,0) (:630) on BIT-PACKED rows: one 64-bit word = 64 pixels, one wave = one 64x64 tile, lane = row.
//   mbits[y][w] bit i = (mask(y, 64w+i) == seed value)      rbits = pixels reached so far
// Inside a row the fill is carry arithmetic ((m + r) ripples a seed through its run of 1s); between rows it is a wave
// shuffle; a tile relaxes to its fixed point in registers, the host relaunches until no tile changed (the wavefront
// crosses one tile per launch, from every side at once when the whole outer ring is background - always the case after
// the morphological close of a mask whose 2-pixel border is empty).
__device__ __forceinline__ uint64_t ss_hfill(uint64_t r, uint64_t m) {
  r &= m;
  const uint64_t up = m & ~(m + r);  // run bits at and above each seed
  const uint64_t rr = __brevll(r), mr = __brevll(m);
  const uint64_t dn = __brevll(mr & ~(mr + rr));
  return r | up | dn;
}

// The same along a column of the tile (lane = row): reached[y] = r[y] | (m[y] & reached[y -+ 1]) for all 64 bit columns at once, as
// two log-step scans over the lanes with (G, P) = (reached, passable): a segment of rows hands down G | (P & G_above), P & P_above.
// Round 2 walked one row per trip (up to 63 trips for an empty tile filled from its top edge); this is 6 + 6 steps.
__device__ __forceinline__ uint64_t ss_shfl64(uint64_t v, int src_lane) {
  return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), src_lane, kWave) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src_lane, kWave);
}
__device__ __forceinline__ uint64_t ss_vfill(uint64_t r, uint64_t m, int lane) {
  uint64_t gd = r & m, pd = m, gu = gd, pu = m;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const bool has_up = lane >= d, has_dn = lane + d < kWave;
    const uint64_t gs = ss_shfl64(gd, lane - d), ps = ss_shfl64(pd, lane - d);  // (out-of-range source lanes wrap: masked by has_*)
    const uint64_t gt = ss_shfl64(gu, lane + d), pt = ss_shfl64(pu, lane + d);
    gd |= has_up ? (pd & gs) : 0, pd &= has_up ? ps : 0;
    gu |= has_dn ? (pu & gt) : 0, pu &= has_dn ? pt : 0;
  }
  return gd | gu;
}

__global__ __launch_bounds__(kBlock) void ss_flood_pack_kernel(const uint8_t* mask, uint64_t* mbits, uint64_t* rbits, int rows, int cols, int W64) {
  // one wave per word: lane = pixel; padded pixels (x >= cols) are 0 in mbits
  const int img = blockIdx.z;
  const size_t wid = (size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;  // word index inside the image
  if (wid >= (size_t)rows * W64) return;
  const int y = (int)(wid / W64), w = (int)(wid % W64), x = w * 64 + (threadIdx.x & (kWave - 1));
  const uint8_t* im = mask + (size_t)img * rows * cols;
  const uint8_t seed = im[0];
  const bool bit = x < cols && im[(size_t)y * cols + x] == seed;
  const unsigned long long b = __ballot(bit);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    mbits[(size_t)img * rows * W64 + wid] = b;
    rbits[(size_t)img * rows * W64 + wid] = 0;
  }
}

// seeds: the origin; or the whole outer ring when every ring pixel has the seed value (then all of it is 4-connected to the origin)
__global__ __launch_bounds__(kBlock) void ss_flood_seed_kernel(const uint64_t* mbits, uint64_t* rbits, int rows, int cols, int W64) {
  const size_t base = (size_t)blockIdx.x * rows * W64;
  const uint64_t* m = mbits + base;
  uint64_t* r = rbits + base;
  const uint64_t lastmask = (cols % 64) ? ((1ull << (cols % 64)) - 1) : ~0ull;
  const int lastbit = (cols - 1) % 64;
  int bad = 0;
  for (int i = threadIdx.x; i < W64; i += kBlock) {
    const uint64_t full = (i == W64 - 1) ? lastmask : ~0ull;
    bad |= (m[i] & full) != full || (m[(size_t)(rows - 1) * W64 + i] & full) != full;
  }
  for (int y = threadIdx.x; y < rows; y += kBlock) bad |= !(m[(size_t)y * W64] & 1ull) || !((m[(size_t)y * W64 + W64 - 1] >> lastbit) & 1ull);
  const int ring_ok = !__syncthreads_or(bad);
  if (!ring_ok) {
    if (threadIdx.x == 0) r[0] = 1ull;  // floodFill always repaints the seed pixel itself
    return;
  }
  for (int i = threadIdx.x; i < W64; i += kBlock) {
    const uint64_t full = (i == W64 - 1) ? lastmask : ~0ull;
    r[i] = full;
    r[(size_t)(rows - 1) * W64 + i] = full;
  }
  __syncthreads();
  for (int y = threadIdx.x + 1; y < rows - 1; y += kBlock) {
    atomicOr((unsigned long long*)&r[(size_t)y * W64], 1ull);
    atomicOr((unsigned long long*)&r[(size_t)y * W64 + W64 - 1], 1ull << lastbit);
  }
}

// One relaxation of one 64x64 tile (one wave, lane = row): halos from the neighbouring tiles as they stand in memory, then the
// tile runs to its fixed point in registers.  Returns true (wave-uniform) if any row of the tile gained pixels.
// COHERENT: reads of the reached set bypass the CU's vector L1 (agent-scope atomic loads) - needed when the SAME kernel relaxes
// tiles repeatedly (ss_flood_finish_kernel); across kernel launches the caches are invalidated anyway.
template <bool COHERENT>
__device__ __forceinline__ bool ss_flood_tile(const uint64_t* mb, uint64_t* rb, int rows, int W64, int ty, int w, int lane) {
  auto rd = [&](size_t idx) -> uint64_t {
    if constexpr (COHERENT)
      return __hip_atomic_load(reinterpret_cast<unsigned long long*>(rb + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      return rb[idx];
  };
  const int y = ty * 64 + lane;
  const bool in = y < rows;
  const uint64_t m = in ? mb[(size_t)y * W64 + w] : 0;
  const uint64_t r0 = in ? rd((size_t)y * W64 + w) : 0;
  uint64_t side = 0;
  if (in && w > 0 && (rd((size_t)y * W64 + w - 1) >> 63)) side |= 1ull;
  if (in && w < W64 - 1 && (rd((size_t)y * W64 + w + 1) & 1ull)) side |= 1ull << 63;
  uint64_t vert = 0;
  if (lane == 0 && y > 0 && in) vert = rd((size_t)(y - 1) * W64 + w);
  if (lane == 63 && y + 1 < rows) vert = rd((size_t)(y + 1) * W64 + w);
  uint64_t r = ss_hfill(r0 | ((side | vert) & m), m);
  for (;;) {  // whole columns, then whole rows, until neither adds a pixel (the least fixed point, whatever the order of the steps)
    const uint64_t rn = ss_hfill(ss_vfill(r, m, lane), m);
    const bool ch = rn != r;
    r = rn;
    if (!__any(ch)) break;
  }
  const bool grew = in && r != r0;
  if (grew) {
    if constexpr (COHERENT)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(rb + (size_t)y * W64 + w), r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      rb[(size_t)y * W64 + w] = r;
  }
  return __any(grew);
}

// The flood fill runs WITHOUT a host round trip: the host enqueues a fixed batch of kSsFloodBatch relaxation launches plus one
// ss_flood_finish_kernel.  flags[stream][k] = "launch k changed something in this image"; launch k of an image returns at once when
// launch k-1 changed nothing (the fill has converged: typical masks need 2-4 launches, the wavefront crosses up to kSsFloodRounds
// tiles per launch from every side), and the finish kernel only works when the last batch launch still changed something - then ONE workgroup
// per image keeps relaxing all tiles until nothing changes (slow, but any mask converges; spiral masks in the tests).
constexpr int kSsFloodBatch = 6, kSsFloodRounds = 10;
constexpr int kSsFloodFlags = kSsFloodBatch + 1;  // per stream; the last one: "the finish kernel had to work" (diagnostics)

__global__ __launch_bounds__(kBlock) void ss_flood_kernel(const uint64_t* mbits, uint64_t* rbits, int rows, int W64, int* flags, int k) {
  int* fl = flags + (size_t)blockIdx.z * kSsFloodFlags;
  if (k > 0 && fl[k - 1] == 0) return;  // converged in an earlier launch (stream order makes fl[k-1] final)
  const int lane = threadIdx.x & (kWave - 1);
  const int tilesY = (rows + 63) / 64;
  const size_t tile = (size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (tile >= (size_t)tilesY * W64) return;  // whole wave
  const size_t base = (size_t)blockIdx.z * rows * W64;
  // kSsFloodRounds relaxations per launch: all tiles of the launch are resident at once (510 per 1080p image), so a tile that
  // re-reads its neighbours' halos (past the L1: COHERENT) sees what they reached a moment ago and the wavefront crosses up to
  // kSsFloodRounds tiles per launch instead of one.  An all-background 1080p mask (the common case: the fill has to come in
  // from the ring to the centre, 9 tiles) needed 9-12 launches with one relaxation each; stale reads only delay, never break, a
  // monotone fill.
  bool grew = false;
  for (int round = 0; round < kSsFloodRounds; ++round)
    grew |= ss_flood_tile<true>(mbits + base, rbits + base, rows, W64, (int)(tile / W64), (int)(tile % W64), lane);
  if (grew && lane == 0) fl[k] = 1;
}

// Round 3: one 1024-lane workgroup per COLUMN STRIP (64 pixels wide, the whole image height) instead of one wave per tile.  The
// tiles of a strip (wave v owns tiles v, v + 16, ...; their rows stay in registers) hand their first / last rows to each other
// through LDS, so the fill runs down or up a whole strip inside the launch at barrier speed; only the columns beside the strip
// come from memory (the neighbouring strips' words, read past the L1).  A relaxation round of the tile kernel above cost a trip
// to memory (its launches ran 40-70 us with ten rounds each, 170 us per frame for 8 x 1080p); here the ring-seeded fill of an
// empty 1080p mask - the common case - is complete after the first pass over the strip.  Same flags, same finish kernel.
// Round 4: NW waves per strip, KT tiles per wave.  <4, 5> (images up to 1280 rows) is the usual form: 240 workgroups of 256 lanes for
// 8 x 1080p find wave slots beside phase B (which runs on its own stream at the same time) more easily than 1024-lane ones - the
// first strip launch 224 instead of 250 us there, 43 us alone either way; what slows it beside phase B is its rounds of memory
// round trips queueing behind phase B's scattered writes (engine_subsense.h).  <16, 4>: images up to 4096 rows (taller ones take
// ss_flood_kernel); BGS_SS_FLOOD_WG1024=1 forces it for every height (tests).
constexpr int kSsFloodKT = 4, kSsFloodKTSmall = 5, kSsFloodNWSmall = 4;
template <int NW, int KT>
__global__ __launch_bounds__(NW * 64) void ss_flood_strip_kernel(const uint64_t* mbits, uint64_t* rbits, int rows, int W64, int* flags, int k) {
  int* fl = flags + (size_t)blockIdx.y * kSsFloodFlags;
  if (k > 0 && fl[k - 1] == 0) return;  // converged in an earlier launch
  __shared__ uint64_t top[NW * KT], bot[NW * KT];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, w = blockIdx.x;
  const int tilesY = (rows + 63) / 64;
  const size_t base = (size_t)blockIdx.y * rows * W64;
  const uint64_t* mb = mbits + base;
  uint64_t* rb = rbits + base;
  auto rd = [&](size_t idx) -> uint64_t { return __hip_atomic_load(reinterpret_cast<unsigned long long*>(rb + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  uint64_t m[KT], r[KT], stored[KT], side_seen[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int t = wave + NW * j, y = t * 64 + lane;
    const bool in = t < tilesY && y < rows;
    m[j] = in ? mb[(size_t)y * W64 + w] : 0;
    stored[j] = in ? rd((size_t)y * W64 + w) : 0;
    r[j] = ss_hfill(stored[j], m[j]);
    side_seen[j] = 0;
  }
  bool grew_any = false;
  for (int round = 0; round < kSsFloodRounds; ++round) {
    // the columns beside the strip as they stand in memory now
    bool news = round == 0;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int t = wave + NW * j, y = t * 64 + lane;
      uint64_t side = 0;
      if (t < tilesY && y < rows) {
        if (w > 0 && (rd((size_t)y * W64 + w - 1) >> 63)) side |= 1ull;
        if (w < W64 - 1 && (rd((size_t)y * W64 + w + 1) & 1ull)) side |= 1ull << 63;
      }
      news |= side != side_seen[j];
      side_seen[j] = side;
      r[j] |= side & m[j];
    }
    if (!__syncthreads_or(news)) break;  // nothing new came in from the sides: whatever the neighbours still do is the next launch's business
    unsigned dirty = (1u << KT) - 1;  // tiles that received pixels since they were last relaxed (wave-uniform)
    for (;;) {  // the strip to its fixed point: tiles to theirs, first / last rows across tile borders through LDS
      bool ch = false;
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int t = wave + NW * j;
        if (t < tilesY && ((dirty >> j) & 1u)) {  // (wave-uniform)
          if (__all(m[j] == ~0ull) && __any(r[j] != 0)) {
            r[j] = ~0ull;  // an empty 64 x 64 tile with a reached pixel anywhere: all of it (the common case, no scan needed)
          } else {
            for (;;) {
              const uint64_t rn = ss_hfill(ss_vfill(r[j], m[j], lane), m[j]);
              const bool c = rn != r[j];
              r[j] = rn;
              if (!__any(c)) break;
            }
          }
          if (lane == 0) top[t] = r[j];
          if (lane == 63) bot[t] = r[j];
        }
      }
      dirty = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int t = wave + NW * j;
        if (t < tilesY) {
          uint64_t v = 0;
          if (lane == 0 && t > 0) v = bot[t - 1];
          if (lane == 63 && t + 1 < tilesY) v = top[t + 1];
          const uint64_t rn = r[j] | (v & m[j]);
          if (__any(rn != r[j])) dirty |= 1u << j, ch = true;
          r[j] = rn;
        }
      }
      if (!__syncthreads_or(ch)) break;  // (also keeps the next trip's LDS writes behind this trip's reads)
    }
    bool grew = false;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int t = wave + NW * j, y = t * 64 + lane;
      if (t < tilesY && y < rows && r[j] != stored[j]) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(rb + (size_t)y * W64 + w), r[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stored[j] = r[j], grew = true;
      }
    }
    grew_any |= grew;
  }
  if (__syncthreads_or(grew_any) && threadIdx.x == 0) fl[k] = 1;
}

__global__ __launch_bounds__(1024) void ss_flood_finish_kernel(const uint64_t* mbits, uint64_t* rbits, int rows, int W64, int* flags, int batch) {
  int* fl = flags + (size_t)blockIdx.x * kSsFloodFlags;
  if (batch > 0 && fl[batch - 1] == 0) return;  // the batch converged: the usual case
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  const int ntiles = ((rows + 63) / 64) * W64;
  const size_t base = (size_t)blockIdx.x * rows * W64;
  if (threadIdx.x == 0) fl[kSsFloodBatch] = 1;
  for (;;) {
    bool ch = false;
    for (int tile = wave; tile < ntiles; tile += nwaves) ch |= ss_flood_tile<true>(mbits + base, rbits + base, rows, W64, tile / W64, tile % W64, lane);
    __threadfence();
    if (!__syncthreads_or(ch)) break;
  }
}

// ---- the post-processing chain on BIT PLANES (:628-636) ------------------------------------------------------------------------
// Every mask of the chain is binary, so between phase A's byte mask and the byte maps the next frame reads, the chain runs on
// bit-packed rows ([images][rows][W64] words, bit i of word w = pixel 64 w + i, bits at x >= cols are 0 in every plane): a lane owns
// one word = 64 pixels, a 3x3 / 7x7 erode or dilate is shifts and ANDs / ORs of 3 words per row, the flood fill already worked
// this way.  ~2 operations per pixel instead of ~40 LDS-bound ones in the byte kernels (morph_box_kernel, still used by the
// stand-alone mask operations, LOBSTER and GMG).

// bytes -> bits (non-zero).  G = 16: a lane packs 16 pixels (one dwordx4), four lanes make a word; needs cols % 16 == 0 and
// 16-byte aligned rows.  G = 1: one wave per word, a ballot (any geometry).
template <int G>
__global__ __launch_bounds__(kBlock) void ss_bits_pack_kernel(const uint8_t* src, uint64_t* dst, int rows, int cols, int W64, size_t nwords) {
  if constexpr (G == 16) {
    const size_t id = (size_t)blockIdx.x * kBlock + threadIdx.x;  // (word, quarter)
    const size_t wid = id >> 2;
    const int q = (int)(id & 3);
    uint32_t nib = 0;
    const bool inw = wid < nwords;
    if (inw) {
      const size_t row = wid / W64;  // image * rows + y
      const int x = (int)(wid % W64) * 64 + q * 16;
      if (x < cols) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + row * cols + x);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t t = ((((d[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d[k]) >> 7) & 0x01010101u;  // bit 0 of each byte: byte != 0
          nib |= ((t * 0x01020408u) >> 24 & 0xfu) << (4 * k);
        }
      }
    }
    uint64_t w = (uint64_t)nib << (16 * q);
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)w, o, kWave), hi = __shfl_xor((uint32_t)(w >> 32), o, kWave);
      w |= ((uint64_t)hi << 32) | lo;
    }
    if (inw && q == 0) dst[wid] = w;
  } else {
    const size_t wid = (size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (wid >= nwords) return;  // whole wave
    const size_t row = wid / W64;
    const int x = (int)(wid % W64) * 64 + (threadIdx.x & (kWave - 1));
    const unsigned long long b = __ballot(x < cols && src[row * cols + x] != 0);
    if ((threadIdx.x & (kWave - 1)) == 0) dst[wid] = b;
  }
}

// (2R+1) x (2R+1) box erode (OP 0) / dilate (OP 1) = R iterations of the 3x3 operation; cells outside the image never take part.
template <int OP, int R>
__global__ __launch_bounds__(kBlock) void ss_bits_box_kernel(const uint64_t* in, uint64_t* out, int rows, int cols, int W64, size_t nwords) {
  const size_t wid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (wid >= nwords) return;
  const int w = (int)(wid % W64);
  const size_t row = wid / W64;
  const int y = (int)(row % rows);
  const uint64_t lastmask = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  constexpr uint64_t ident = OP == 0 ? ~0ull : 0ull;
  auto word = [&](size_t r, int ww) -> uint64_t {  // a row's word with everything outside the image replaced by the identity
    if (ww < 0 || ww >= W64) return ident;
    uint64_t v = in[r * W64 + ww];
    if (OP == 0 && ww == W64 - 1) v |= ~lastmask;
    return v;
  };
  uint64_t acc = ident;
#pragma unroll
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= rows) continue;
    const size_t r = row + dy;
    const uint64_t c = word(r, w), l = word(r, w - 1), rr = word(r, w + 1);
    uint64_t h = c;
#pragma unroll
    for (int d = 1; d <= R; ++d) {
      const uint64_t fromleft = (c << d) | (l >> (64 - d)), fromright = (c >> d) | (rr << (64 - d));  // pixels x-d and x+d
      h = OP == 0 ? (h & fromleft & fromright) : (h | fromleft | fromright);
    }
    acc = OP == 0 ? (acc & h) : (acc | h);
  }
  out[wid] = (w == W64 - 1) ? (acc & lastmask) : acc;
}

// cv::medianBlur(k) of a BINARY mask on bit planes: a pixel is set iff at least (k*k + 1) / 2 of the k x k cells around it are set,
// cells outside the image taking the nearest pixel's value (BORDER_REPLICATE).  One lane = one 64-pixel word: the column sums of the
// k rows as a 4-plane bit-sliced counter for the word and its two neighbours, then the k shifted copies added into an 8-plane
// accumulator, then a bit-sliced compare with the threshold - ~25 instructions per pixel for 13 x 13, no LDS.  Writes the bit plane
// and the byte map (0 / 255) the next frame's phase A reads.  (Round 2 counted in LDS from bytes: morph_box_kernel, 84 us for
// 8 x 1080p at k = 13 - the longest kernel of the post-processing chain; this one takes its place at any odd k <= 13.)
template <int R>
__global__ __launch_bounds__(kBlock) void ss_bits_median_kernel(const uint64_t* in, uint64_t* out_bits, uint8_t* out_bytes, int rows, int cols, int W64, size_t nwords) {
  constexpr int K = 2 * R + 1, T = (K * K + 1) / 2;
  const size_t wid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (wid >= nwords) return;
  const int w = (int)(wid % W64);
  const size_t row = wid / W64, img = row / rows;
  const int y = (int)(row % rows);
  const int mbits = cols - (W64 - 1) * 64;  // valid bits of a row's last word (1..64)
  const uint64_t lastmask = mbits == 64 ? ~0ull : ((1ull << mbits) - 1);
  // column sums (0..K) of the rows y-R..y+R (clamped) for the words w-1, w, w+1, every word extended beyond the image by replication
  uint64_t V[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b) V[j][b] = 0;
#pragma unroll
  for (int dy = -R; dy <= R; ++dy) {
    const uint64_t* r = in + (img * rows + (size_t)min(max(y + dy, 0), rows - 1)) * W64;
    const uint64_t first = r[0], last = r[W64 - 1];
    const uint64_t left_fill = (first & 1ull) ? ~0ull : 0ull, right_fill = ((last >> (mbits - 1)) & 1ull) ? ~0ull : 0ull;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ww = w + j - 1;
      uint64_t x = ww < 0 ? left_fill : ww >= W64 ? right_fill : r[ww];
      if (ww == W64 - 1) x |= right_fill & ~lastmask;
      uint64_t carry = x;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint64_t t = V[j][b] & carry;
        V[j][b] ^= carry, carry = t;
      }
    }
  }
  // window sums: the K horizontally shifted copies of the centre word's column sums
  uint64_t acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = b < 4 ? V[1][b] : 0;
#pragma unroll
  for (int d = 1; d <= R; ++d)
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      uint64_t v[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) v[b] = side == 0 ? (V[1][b] << d) | (V[0][b] >> (64 - d)) : (V[1][b] >> d) | (V[2][b] << (64 - d));  // pixels x - d / x + d
      uint64_t carry = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b < 4) {
          const uint64_t axb = acc[b] ^ v[b];
          const uint64_t c2 = (acc[b] & v[b]) | (carry & axb);
          acc[b] = axb ^ carry, carry = c2;
        } else {
          const uint64_t t = acc[b] & carry;
          acc[b] ^= carry, carry = t;
        }
      }
    }
  // acc >= T, bit-sliced: from the top bit down, "greater so far" | "equal so far"
  uint64_t gt = 0, eq = ~0ull;
#pragma unroll
  for (int b = 7; b >= 0; --b) {
    if ((T >> b) & 1)
      eq &= acc[b];
    else
      gt |= eq & acc[b], eq &= ~acc[b];
  }
  uint64_t res = gt | eq;
  if (w == W64 - 1) res &= lastmask;
  if (out_bits) out_bits[wid] = res;
  uint8_t* ob = out_bytes + (img * rows + y) * (size_t)cols + (size_t)w * 64;
  const int nvalid = w == W64 - 1 ? mbits : 64;
  if ((cols & 15) == 0 && (reinterpret_cast<uintptr_t>(out_bytes) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q * 16 < nvalid) {
        uint32_t dw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t nib = (uint32_t)(res >> (q * 16 + k * 4)) & 0xfu;
          dw[k] = ((nib * 0x00204081u) & 0x01010101u) * 0xffu;  // bit i of the nibble -> byte i = 0 / 255
        }
        *reinterpret_cast<uint4*>(ob + q * 16) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
      }
    }
  } else {
    for (int i = 0; i < nvalid; ++i) ob[i] = ((res >> i) & 1ull) ? 255 : 0;
  }
}

// flood fill operands from the closed mask: mbits = pixels that have the seed's value (the value at (0,0)), rbits = nothing reached yet
__global__ __launch_bounds__(kBlock) void ss_bits_flood_prepare_kernel(const uint64_t* pre, uint64_t* mbits, uint64_t* rbits, int rows, int cols, int W64, size_t nwords) {
  const size_t wid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (wid >= nwords) return;
  const size_t img = wid / ((size_t)rows * W64);
  const bool seed = pre[img * rows * W64] & 1ull;
  const uint64_t lastmask = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  const uint64_t v = pre[wid], valid = ((int)(wid % W64) == W64 - 1) ? lastmask : ~0ull;
  mbits[wid] = (seed ? v : ~v) & valid;
  rbits[wid] = 0;
}

// :631-634  cur = raw | ~floodFill(pre) | erode3(pre), floodFill(pre) = reached ? 255 : pre
__global__ __launch_bounds__(kBlock) void ss_bits_combine_kernel(const uint64_t* raw, const uint64_t* pre, const uint64_t* reached, const uint64_t* eroded, uint64_t* out, int cols, int W64,
                                                                 size_t nwords) {
  const size_t wid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (wid >= nwords) return;
  const uint64_t lastmask = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  const uint64_t valid = ((int)(wid % W64) == W64 - 1) ? lastmask : ~0ull;
  out[wid] = (raw[wid] | ~(reached[wid] | pre[wid]) | eroded[wid]) & valid;
}

// cv::floodFill(img, Point(0,0), 255) as an image: reached pixels become 255, the rest keep their value
__global__ __launch_bounds__(kBlock) void ss_flood_paint_kernel(const uint8_t* src, const uint64_t* rbits, uint8_t* dst, int rows, int cols, int W64) {
  const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= (size_t)rows * cols) return;
  const int y = (int)(p / cols), x = (int)(p % cols);
  dst[p] = ((rbits[(size_t)y * W64 + (x >> 6)] >> (x & 63)) & 1ull) ? 255 : src[p];
}

// :637-642: blink mask clean-up against the dilated final mask (old then new), final-segmentation running means, output
// `dilated` is a bit plane ([images][rows][W64]).
template <int G>  // pixels per lane: 4 (dword per byte map, float4 per mean map; cols % 4 == 0: they share a word of the bit plane) or 1
__global__ __launch_bounds__(kBlock) void ss_finish_kernel(const SsArgs a, const uint64_t* dilated, int W64, size_t count) {
  const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * G;
  if (i >= count) return;
  const size_t N = (size_t)a.rows * a.cols, g = (size_t)a.first * N + i;
  const size_t img = i / N, pp = i % N;
  const int py = (int)(pp / a.cols), px = (int)(pp % a.cols);
  const uint64_t dword = dilated[(img * a.rows + py) * W64 + (px >> 6)] >> (px & 63);
  // cv::addWeighted(f32, 1-f, u8, (1/255)*f, 0, dst, CV_32F): both operands as float, arithmetic in double
  const double aLT = (double)(1.0f - a.fLT), bLT = __dmul_rn(1.0 / 255, (double)a.fLT), aST = (double)(1.0f - a.fST), bST = __dmul_rn(1.0 / 255, (double)a.fST);
  if constexpr (G == 4) {
    const uint32_t b = *reinterpret_cast<const uint32_t*>(a.blinks + g) & *reinterpret_cast<const uint32_t*>(a.lastDilInv + g);
    const uint32_t nib = (uint32_t)dword & 0xfu;
    const uint32_t inv = ~(((nib & 1u) * 0xffu) | ((nib & 2u) * (0xff00u >> 1)) | ((nib & 4u) * (0xff0000u >> 2)) | ((nib & 8u) * (0xff000000u >> 3)));
    *reinterpret_cast<uint32_t*>(a.lastDilInv + g) = inv;
    *reinterpret_cast<uint32_t*>(a.blinks + g) = b & inv;
    const uint32_t m4 = *reinterpret_cast<const uint32_t*>(a.lastFG + g);
    if (a.fg) *reinterpret_cast<uint32_t*>(a.fg + i) = m4;
    float4 lt = *reinterpret_cast<const float4*>(a.FinLT + g), st = *reinterpret_cast<const float4*>(a.FinST + g);
    float* l = &lt.x;
    float* t = &st.x;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const double m = (double)(float)((m4 >> (8 * o)) & 0xffu);
      l[o] = (float)__dadd_rn(__dmul_rn((double)l[o], aLT), __dmul_rn(m, bLT));
      t[o] = (float)__dadd_rn(__dmul_rn((double)t[o], aST), __dmul_rn(m, bST));
    }
    *reinterpret_cast<float4*>(a.FinLT + g) = lt;
    *reinterpret_cast<float4*>(a.FinST + g) = st;
  } else {
    uint8_t b = a.blinks[g] & a.lastDilInv[g];
    const uint8_t inv = (dword & 1ull) ? 0 : 255;
    a.lastDilInv[g] = inv;
    a.blinks[g] = b & inv;
    const uint8_t m = a.lastFG[g];
    if (a.fg) a.fg[i] = m;
    a.FinLT[g] = (float)__dadd_rn(__dmul_rn((double)a.FinLT[g], aLT), __dmul_rn((double)(float)m, bLT));
    a.FinST[g] = (float)__dadd_rn(__dmul_rn((double)a.FinST[g], aST), __dmul_rn((double)(float)m, bST));
  }
}

// ----------------------------------------------------------------------------------------------- frame-level block :656-665
// cv::resize(..., width/8 x height/8, INTER_AREA) (:153, :656) for sizes that are NOT multiples of 8: OpenCV's general path
// (resizeArea_<uchar, float> over computeResizeAreaTab's fractional cell weights; recalled from OpenCV 2.4 imgwarp.cpp, unpinned -
// the tests' CPU restatement states the same function).  One destination index of one axis: a left partial cell, whole cells [s1, s2), a right partial.
struct SsAreaSpan {
  int l, s1, s2, r;
  float al, af, ar;
};
__device__ __forceinline__ SsAreaSpan ss_area_span(int ssize, int dsize, int d) {
  const double scale = 1.0 / ((double)dsize / ssize);  /* as cv::resize computes it: inv_scale = dsize / ssize, scale = 1 / inv_scale (differs from ssize / dsize in the last ulp for some sizes) */
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = min(sx2, ssize - 1);
  sx1 = min(sx1, sx2);
  SsAreaSpan a;
  a.l = a.r = -1, a.al = a.ar = 0.f, a.s1 = sx1, a.s2 = sx2, a.af = (float)(1.0 / cell);
  if (sx1 - fsx1 > 1e-3) a.l = sx1 - 1, a.al = (float)((sx1 - fsx1) / cell);
  if (fsx2 - sx2 > 1e-3) {
    double w = fsx2 - sx2;
    w = w > 1. ? 1. : w;
    w = w > cell ? cell : w;
    a.r = sx2, a.ar = (float)(w / cell);
  }
  return a;
}
template <int C>
__device__ __forceinline__ float ss_area_row(const uint8_t* row, int c, const SsAreaSpan& ax) {
  float buf = 0.f;
  if (ax.l >= 0) buf += row[(size_t)ax.l * C + c] * ax.al;
  for (int sx = ax.s1; sx < ax.s2; ++sx) buf += row[(size_t)sx * C + c] * ax.af;
  if (ax.r >= 0) buf += row[(size_t)ax.r * C + c] * ax.ar;
  return buf;
}
template <int C>
__device__ __forceinline__ float ss_area_value(const uint8_t* img, int rows, int cols, int dsh, int dsw, int y, int x, int c) {
  const SsAreaSpan ax = ss_area_span(cols, dsw, x), ay = ss_area_span(rows, dsh, y);
  float sum = 0.f;
  bool first = true;
  if (ay.l >= 0) sum = ay.al * ss_area_row<C>(img + (size_t)ay.l * cols * C, c, ax), first = false;
  for (int sy = ay.s1; sy < ay.s2; ++sy) {
    const float b = ay.af * ss_area_row<C>(img + (size_t)sy * cols * C, c, ax);
    sum = first ? b : sum + b, first = false;
  }
  if (ay.r >= 0) {
    const float b = ay.ar * ss_area_row<C>(img + (size_t)ay.r * cols * C, c, ax);
    sum = first ? b : sum + b;
  }
  return sum;
}

template <int C>
__global__ __launch_bounds__(kBlock) void ss_downsample_kernel(const SsArgs a) {
  const int stream = a.first + blockIdx.z;
  const int dsw = a.cols / 8, dsh = a.rows / 8;
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  unsigned diff = 0;
  if (idx < dsw * dsh) {
    const int x = idx % dsw, y = idx / dsw;
    const uint8_t* img = a.frame + (size_t)blockIdx.z * a.rows * a.cols * C;
    float d[C];
    const bool whole = a.rows % 8 == 0 && a.cols % 8 == 0;
    uint32_t sums[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sums[c] = 0;
    if (whole) {  // the cell's 8 rows of 8 pixels as dwords (global loads need no alignment here), channel sums with v_dot4_u32_u8 / v_sad_u8
      typedef uint32_t __attribute__((aligned(1))) u32u;
      for (int yy = 0; yy < 8; ++yy) {
        const u32u* rp = reinterpret_cast<const u32u*>(img + ((size_t)(y * 8 + yy) * a.cols + x * 8) * C);
        if constexpr (C == 3) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {  // 4 pixels = B G R B | G R B G | R B G R
            const uint32_t w0 = rp[3 * h], w1 = rp[3 * h + 1], w2 = rp[3 * h + 2];
            sums[0] = __builtin_amdgcn_udot4(w2, 0x00000100u, __builtin_amdgcn_udot4(w1, 0x00010000u, __builtin_amdgcn_udot4(w0, 0x01000001u, sums[0], false), false), false);
            sums[1] = __builtin_amdgcn_udot4(w2, 0x00010000u, __builtin_amdgcn_udot4(w1, 0x01000001u, __builtin_amdgcn_udot4(w0, 0x00000100u, sums[1], false), false), false);
            sums[2] = __builtin_amdgcn_udot4(w2, 0x01000001u, __builtin_amdgcn_udot4(w1, 0x00000100u, __builtin_amdgcn_udot4(w0, 0x00010000u, sums[2], false), false), false);
          }
        } else {
          sums[0] = __builtin_amdgcn_sad_u8(rp[1], 0u, __builtin_amdgcn_sad_u8(rp[0], 0u, sums[0]));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v;
      if (whole) {  // cv::resize INTER_AREA, integer ratio on both axes (resizeAreaFast_)
        v = (float)sat_u8((float)sums[c] * (1.f / 64));
      } else {
        v = (float)sat_u8(ss_area_value<C>(img, a.rows, a.cols, dsh, dsw, y, x, c));
      }
      float* lt = a.dsLT + ((size_t)stream * dsw * dsh + idx) * C + c;
      float* st = a.dsST + ((size_t)stream * dsw * dsh + idx) * C + c;
      const float nlt = v * a.fLT + *lt * (1 - a.fLT), nst = v * a.fST + *st * (1 - a.fST);  // cv::accumulateWeighted
      *lt = nlt, *st = nst;
      d[c] = fabsf(nst - nlt);
    }
    if constexpr (C == 3)
      diff = max((unsigned)d[0], max((unsigned)d[1], (unsigned)d[2]));
    else
      diff = (unsigned)d[0] / 2;  // :664
  }
  // block reduction -> one atomic per block
  __shared__ unsigned part[kBlock / kWave];
  unsigned v = diff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += part[w];
    if (t) atomicAdd(&a.sc[stream].totDiff, t);
  }
}

// LUT auto-adjustment :643-655 (one lane per LUT entry) and the scalar logic :666-699 (lane 0), one workgroup per stream
__global__ __launch_bounds__(256) void ss_frame_level_kernel(const SsArgs a) {
  const int stream = a.first + blockIdx.x;
  SsScalars* sc = a.sc + stream;
  __shared__ float ratio_s, last_s;
  if (threadIdx.x == 0) {
    const size_t relevant = (size_t)(a.rows - 4) * (a.cols - 4);
    ratio_s = (float)sc->nzCount / relevant;
    last_s = sc->lastNZ;
  }
  __syncthreads();
  const float ratio = ratio_s, last = last_s;
  uint8_t* lut = a.lut + (size_t)stream * 256;
  const int t = threadIdx.x;
  if (ratio < 0.1f && last < 0.1f) {
    const double lim = (double)a.lbspOff + ceil((double)((float)t * a.relT / 4));
    if (lut[t] > sat_u8((float)lim)) --lut[t];
  } else if (ratio > 0.5f && last > 0.5f) {
    if (lut[t] < sat_u8((float)a.lbspOff + 255 * a.relT)) ++lut[t];
  }
  if (threadIdx.x != 0) return;
  sc->lastNZ = ratio;
  sc->nzCount = 0;
  sc->doRefresh = 0;
  if (a.lrScaling) {
    const float diffRatio = (float)sc->totDiff / ((a.rows / 8) * (a.cols / 8));
    sc->totDiff = 0;
    const int thr = a.nMinColor / 2;
    if (sc->autoReset) {
      if (sc->framesSinceReset > 1000)
        sc->autoReset = 0;
      else if (diffRatio >= thr && sc->cooldown == 0) {
        sc->framesSinceReset = 0;
        sc->doRefresh = 1;  // refreshModel(0.1f) + UpdateRate = 1 run in ss_refresh_kernel(mode 1), next in the stream
        sc->cooldown = a.nMov / 4;
      } else
        ++sc->framesSinceReset;
    } else if (diffRatio >= thr * 2) {
      sc->framesSinceReset = 0;
      sc->autoReset = 1;
    }
    if (diffRatio >= thr / 2) {
      const int lo = 2 >> (int)(diffRatio / 2), hi = 256 >> (int)(diffRatio / 2);
      sc->capLo = (float)max(lo, 1), sc->capHi = (float)max(hi, 1);
    } else {
      sc->capLo = 2.0f, sc->capHi = 256.0f;
    }
    if (sc->cooldown > 0) --sc->cooldown;
  }
}

// getBackgroundImage :702-718
template <int C>
__global__ __launch_bounds__(kBlock) void ss_background_kernel(const SsArgs a) {
  const int stream = a.first + blockIdx.z;
  const size_t N = (size_t)a.rows * a.cols;
  const size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;  // byte index inside one [N][C] image
  if (e >= N * C) return;
  const size_t p = e / C;
  const int c = (int)(e - p * C);
  // the colour word of a record is its first dword: one 4-byte load per sample
  const uint32_t* base = reinterpret_cast<const uint32_t*>(a.samples);
  float acc = 0;
  for (int k = 0; k < a.nS; ++k) acc += div_rn((float)((base[ss_rec(a, stream, N, p, k) * (SsSample<C>::kBytes / 4)] >> (8 * c)) & 0xffu), (float)a.nS);
  a.bgimg[(size_t)blockIdx.z * N * C + e] = (uint8_t)sat_u8(acc);
}

}  // namespace bgs
