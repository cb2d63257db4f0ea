// engine_subsense.h — host-side orchestration of the SuBSENSE kernels (kernel_subsense.h).  Included by bgs_hip.hip inside
// its anonymous namespace, after bgs_engine / fail() / HIP_TRY / Timed are defined.
// One ss_process() call = SuBSENSEBGS::process (package_bgs/pl/SuBSENSE.cpp:21-45) for streams [first, first+count).
//
// Everything is enqueued on the launch stream and NOTHING waits for the device: the frame-level block
// (BackgroundSubtractorSuBSENSE.cpp:643-699) runs on the device, the flood fill (:630) is a fixed batch of launches that stop
// working on a device-side flag (kernel_subsense.h: ss_flood_kernel / ss_flood_finish_kernel), per-stream constants travel in
// kernel arguments.  Calls on one engine must come from one host thread at a time; ranges of streams in flight on different
// HIP streams must be disjoint (flood flags are per stream; every call takes its own {evA, evB} pair from a ring of eight, the one side
// stream is shared: phase B launches run in the order of their calls).

enum { SS_R, SS_V, SS_T, SS_DLAST0, SS_DLAST1, SS_DMINLT, SS_DMINST, SS_RAWLT, SS_RAWST0, SS_RAWST1, SS_FINLT, SS_FINST, SS_NF32 };
enum { SS_UNSTABLE, SS_BLINKS, SS_LASTFG, SS_LASTRAW, SS_LASTRAWBLINK, SS_LASTDILINV, SS_RAW, SS_NU8 };

// launch the C = 3 or C = 1 instantiation of a SuBSENSE kernel template
#define SS_LAUNCH(KERNEL, grid, block, stream, ...)                                             \
  do {                                                                                          \
    if (e->ch == 3)                                                                             \
      hipLaunchKernelGGL((bgs::KERNEL<3>), grid, block, 0, stream, __VA_ARGS__);                \
    else                                                                                        \
      hipLaunchKernelGGL((bgs::KERNEL<1>), grid, block, 0, stream, __VA_ARGS__);                \
  } while (0)

#define SS_LAUNCH_LDS(KERNEL, grid, block, lds, stream, ...)                                     \
  do {                                                                                          \
    if (e->ch == 3)                                                                             \
      hipLaunchKernelGGL((bgs::KERNEL<3>), grid, block, lds, stream, __VA_ARGS__);              \
    else                                                                                        \
      hipLaunchKernelGGL((bgs::KERNEL<1>), grid, block, lds, stream, __VA_ARGS__);              \
  } while (0)

constexpr int SS_NBITS = 6;  // raw, closed-tmp / eroded, pre (closed), combined, final mask, dilated
struct SsDevice {
  void* samples = nullptr;  // records of colour + descriptor (kernel_subsense.h: SsSample, ss_rec)
  int nSpad = 0, pixelMajor = 0;
  uint8_t *lut = nullptr, *lastColor = nullptr, *curColor = nullptr;  // cur*: LOBSTER scratch (kernel_subsense.h)
  uint16_t *lastDesc = nullptr, *req = nullptr, *curDesc = nullptr;
  float* f32[SS_NF32] = {nullptr};
  uint8_t* u8[SS_NU8] = {nullptr};
  float *dsLT = nullptr, *dsST = nullptr;
  bgs::SsScalars* sc = nullptr;
  bgs::SsScalars* scSnap = nullptr;  // the scalars as phase A found them (ss_feedback_kernel runs beside the frame-level block that rewrites sc)
  uint32_t* ho = nullptr;            // [S][N][2] phase A -> ss_feedback_kernel hand-over (kernel_subsense.h)
  void* lastRec = nullptr;           // [S][N] 16-byte records: the full refresh's packed view of last colour / flag / descriptors (BGR)
  uint32_t* magic = nullptr;         // ss_mod's multipliers
  int* flood_flags = nullptr;  // [S][kSsFloodFlags], see ss_flood_kernel
  hipStream_t side = nullptr;  // phase B runs here, beside the post-processing chain (both only need phase A)
  // One {evA, evB} pair per call in flight (a ring: calls for disjoint stream ranges may be in flight on several HIP streams), and the
  // "phase A token": evTok[k] is recorded behind a call's phase A, and the next call's phase A - whatever HIP stream it is on - waits
  // for it (ss_process).  part[]: the engine's own streams for the parts of one large batch (ss_process).
  static constexpr int kRing = 8, kParts = 4;
  hipEvent_t evA[kRing] = {nullptr}, evB[kRing] = {nullptr}, evTok[kRing] = {nullptr};
  int ring = 0, tok = -1;        // next pair to use; the token of the last call (-1: none yet)
  hipStream_t tokStream = nullptr;
  hipStream_t part[kParts] = {nullptr};  // part 0 runs on the caller's stream
  hipEvent_t evFork = nullptr, evPart[kParts] = {nullptr};
  uint64_t *mbits = nullptr, *rbits = nullptr;  // flood fill: bit-packed mask / reached set, [S][rows][W64]
  uint64_t* bitws = nullptr;                     // SS_NBITS more bit planes of the same shape: the post-processing chain's intermediates
  std::vector<uint8_t> pp;   // per stream: which copy of Dlast / RawST is current
  int use3x3 = 1, lrScaling = 0, medK = 9;
  float capLo0 = 4.f, capHi0 = 512.f;
  void release() {
    void* p[] = {samples, lut, lastColor, curColor, lastDesc, curDesc, req, dsLT, dsST, sc, scSnap, ho, lastRec, magic, flood_flags, mbits, rbits, bitws};
    for (void* q : p)
      if (q) (void)hipFree(q);
    for (auto& q : f32)
      if (q) (void)hipFree(q), q = nullptr;
    for (auto& q : u8)
      if (q) (void)hipFree(q), q = nullptr;
    if (side) (void)hipStreamSynchronize(side), (void)hipStreamDestroy(side), side = nullptr;
    for (auto& q : part)
      if (q) (void)hipStreamSynchronize(q), (void)hipStreamDestroy(q), q = nullptr;
    for (int i = 0; i < kRing; ++i)
      for (hipEvent_t* q : {&evA[i], &evB[i], &evTok[i]})
        if (*q) (void)hipEventDestroy(*q), *q = nullptr;
    for (auto& q : evPart)
      if (q) (void)hipEventDestroy(q), q = nullptr;
    if (evFork) (void)hipEventDestroy(evFork), evFork = nullptr;
    ring = 0, tok = -1, tokStream = nullptr;
    samples = nullptr, lut = lastColor = curColor = nullptr, lastDesc = req = curDesc = nullptr, dsLT = dsST = nullptr, sc = nullptr, scSnap = nullptr, ho = nullptr, lastRec = nullptr, magic = nullptr, flood_flags = nullptr, mbits = rbits = nullptr, bitws = nullptr;
  }
};

int ss_allocate(bgs_engine* e) {
  if (e->rows < 5 || e->cols < 5) return fail(BGS_ERR_UNSUPPORTED, "SuBSENSE needs at least 5x5 pixels (LBSP::validateROI)");
  const bgs_params& p = e->p;
  if (p.subsense_n_samples < 1 || p.subsense_n_samples > bgs::kSsMaxSamples || p.subsense_n_required > p.subsense_n_samples)
    return fail(BGS_ERR_UNSUPPORTED, "SuBSENSE: nBGSamples must be 1..%d and nRequiredBGSamples <= nBGSamples", bgs::kSsMaxSamples);
  SsDevice* d = new SsDevice();
  e->ss = d;
  // geometry-dependent switches of BackgroundSubtractorSuBSENSE::initialize (:121-140), ROI = whole frame (SuBSENSE.cpp:36)
  const int total = e->rows * e->cols, qvga = 320 * 240;
  if (total >= qvga) {
    d->lrScaling = 1;
    d->use3x3 = !(total > qvga * 2);
    int k = (int)std::floor((float)total / qvga + 0.5f) + 9;
    k = std::min(k, 14);
    d->medK = (k % 2) ? k : k - 1;
    d->capLo0 = 2.f, d->capHi0 = 256.f;
  }
  const size_t N = e->n, P = N * e->S, nS = (size_t)p.subsense_n_samples, C = (size_t)e->ch;
  // records: the first batch of kSsBatch samples sample-major, the rest pixel-major in whole double batches (kernel_subsense.h: ss_rec, phase A)
  d->pixelMajor = 1, d->nSpad = bgs::kSsBatch + ((int)std::max<size_t>(nS, bgs::kSsBatch) - bgs::kSsBatch + 2 * bgs::kSsBatch - 1) / (2 * bgs::kSsBatch) * (2 * bgs::kSsBatch);
  DMALLOC(d->samples, P * (size_t)d->nSpad * (C == 3 ? 16 : 4));
  DMALLOC(d->lastColor, P * C + 8);      // + 8: ss_refresh_one reads a pixel's 3 bytes / 3 words with one 4- / 8-byte load
  DMALLOC(d->lastDesc, P * C * 2 + 8);
  DMALLOC(d->req, P * 2 * 2);
  DMALLOC(d->lut, (size_t)e->S * 256);
  DMALLOC(d->sc, (size_t)e->S * sizeof(bgs::SsScalars));
  DMALLOC(d->scSnap, (size_t)e->S * sizeof(bgs::SsScalars));
  DMALLOC(d->ho, P * 2 * sizeof(uint32_t));
  if (C == 3) DMALLOC(d->lastRec, P * 16);
  {
    uint32_t m[bgs::kSsMagicN];
    for (uint32_t k = 0; k < (uint32_t)bgs::kSsMagicN; ++k) m[k] = bgs::ss_magic(k);
    DMALLOC(d->magic, sizeof(m));
    HIP_TRY(hipMemcpyAsync(d->magic, m, sizeof(m), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // m is on this stack frame
  }
  DMALLOC(d->flood_flags, (size_t)e->S * bgs::kSsFloodFlags * sizeof(int));
  const size_t words = (size_t)e->S * e->rows * ((e->cols + 63) / 64);
  DMALLOC(d->mbits, words * 8);
  DMALLOC(d->rbits, words * 8);
  if (e->algo == BGS_SUBSENSE) DMALLOC(d->bitws, words * 8 * SS_NBITS);
  for (auto& q : d->f32) DMALLOC(q, P * sizeof(float));
  for (auto& q : d->u8) DMALLOC(q, P);
  const size_t ds = (size_t)(e->rows / 8) * (e->cols / 8) * C * e->S + 4;
  DMALLOC(d->dsLT, ds * sizeof(float));
  DMALLOC(d->dsST, ds * sizeof(float));
  d->pp.assign(e->S, 0);
  return BGS_OK;
}

void ss_fill_args(const bgs_engine* e, bgs::SsArgs& a, int first, int cur_pp, unsigned frameIndex) {
  const SsDevice* d = e->ss;
  const bgs_params& p = e->p;
  a.samples = d->samples, a.nSpad = d->nSpad, a.pixelMajor = d->pixelMajor, a.lastColor = d->lastColor, a.lastDesc = d->lastDesc, a.req = d->req, a.lut = d->lut, a.sc = d->sc;
  a.scSnap = d->scSnap, a.ho = d->ho, a.magic = d->magic, a.lastRec = (uint4*)d->lastRec;
  a.R = d->f32[SS_R], a.V = d->f32[SS_V], a.T = d->f32[SS_T];
  a.DlastOld = d->f32[cur_pp ? SS_DLAST1 : SS_DLAST0], a.DlastNew = d->f32[cur_pp ? SS_DLAST0 : SS_DLAST1];
  a.RawSTOld = d->f32[cur_pp ? SS_RAWST1 : SS_RAWST0], a.RawSTNew = d->f32[cur_pp ? SS_RAWST0 : SS_RAWST1];
  a.DminLT = d->f32[SS_DMINLT], a.DminST = d->f32[SS_DMINST], a.RawLT = d->f32[SS_RAWLT], a.FinLT = d->f32[SS_FINLT], a.FinST = d->f32[SS_FINST];
  a.unstable = d->u8[SS_UNSTABLE], a.blinks = d->u8[SS_BLINKS], a.lastFG = d->u8[SS_LASTFG], a.lastRaw = d->u8[SS_LASTRAW];
  a.lastRawBlink = d->u8[SS_LASTRAWBLINK], a.lastDilInv = d->u8[SS_LASTDILINV], a.raw = d->u8[SS_RAW];
  a.dsLT = d->dsLT, a.dsST = d->dsST;
  a.rows = e->rows, a.cols = e->cols, a.nS = p.subsense_n_samples, a.nReq = p.subsense_n_required, a.nMinColor = p.subsense_min_color_dist_threshold;
  a.nDescOff = p.subsense_desc_dist_threshold_offset, a.nMov = p.subsense_samples_for_moving_avgs, a.lbspOff = p.lbsp_threshold_offset;
  a.use3x3 = d->use3x3, a.lrScaling = d->lrScaling, a.medK = d->medK, a.relT = p.lbsp_rel_threshold;
  static const bool self_in_a = !(getenv("BGS_SS_SELF_IN_A") && atoi(getenv("BGS_SS_SELF_IN_A")) == 0);
  a.selfInA = (e->algo == BGS_SUBSENSE && self_in_a) ? 1 : 0;  // (LOBSTER's phase A leaves every write to phase B)
  static const int refill = getenv("BGS_SS_REFILL") ? std::max(1, std::min(64, atoi(getenv("BGS_SS_REFILL")))) : bgs::kSsRefill;  // tuning knob
  a.refill = refill;
  static const int ipass_min = getenv("BGS_SS_IPASS_MIN") ? std::max(1, std::min(64, atoi(getenv("BGS_SS_IPASS_MIN")))) : bgs::kSsIpassMin;  // 1 = the round-3 form
  a.ipassMin = ipass_min;
  a.frameIndex = frameIndex, a.first = first;
  const int64_t fi = frameIndex ? frameIndex : 1;
  a.fLT = 1.0f / (float)std::min<int64_t>(fi, p.subsense_samples_for_moving_avgs);
  a.fST = 1.0f / (float)std::min<int64_t>(fi, p.subsense_samples_for_moving_avgs / 4);
}

void ss_morph(const uint8_t* src, uint8_t* dst, int rows, int cols, int count, int op, int ksize, hipStream_t s) {
  bgs::MorphArgs m{src, dst, rows, cols, op, ksize};
  bgs::morph_launch(m, count, s);
}

// cv::saturate_cast<uchar>(offset + t * rel) per entry (BackgroundSubtractorSuBSENSE.cpp:227-228); host-side, once per stream
void ss_initial_lut(const bgs_params& p, int channels, uint8_t lut[256]) {
  for (int t = 0; t < 256; ++t) {
    float v = (float)(size_t)p.lbsp_threshold_offset + (float)(size_t)t * p.lbsp_rel_threshold;
    if (channels == 1) v = v / 3;  // :209-210
    long r = std::lrint((double)v);
    lut[t] = (uint8_t)std::min<long>(std::max<long>(r, 0), 255);
  }
}

// refreshModel (kernel_subsense.h): the full refresh of SuBSENSE's record layout takes the 16 pixels x 16 columns form
// One relaxation launch of the flood fill (kernel_subsense.h): a workgroup per 64-pixel column strip for images up to 4096 rows,
// a wave per 64x64 tile otherwise (BGS_SS_FLOOD_TILES=1 forces the latter: A/B and test knob).
void ss_launch_flood(dim3 tile_grid, int count, int tilesY, hipStream_t s, const uint64_t* mbits, uint64_t* rbits, int rows, int W64, int* fl, int k) {
  static const bool tiles_only = getenv("BGS_SS_FLOOD_TILES") && atoi(getenv("BGS_SS_FLOOD_TILES")) == 1;
  static const bool wide_groups = getenv("BGS_SS_FLOOD_WG1024") && atoi(getenv("BGS_SS_FLOOD_WG1024")) == 1;  // A/B and test knob: round 3's 1024-lane workgroups for every height
  if (tilesY <= bgs::kSsFloodNWSmall * bgs::kSsFloodKTSmall && !tiles_only && !wide_groups)
    hipLaunchKernelGGL((bgs::ss_flood_strip_kernel<bgs::kSsFloodNWSmall, bgs::kSsFloodKTSmall>), dim3(W64, count), dim3(bgs::kSsFloodNWSmall * 64), 0, s, mbits, rbits, rows, W64, fl, k);
  else if (tilesY <= 16 * bgs::kSsFloodKT && !tiles_only)
    hipLaunchKernelGGL((bgs::ss_flood_strip_kernel<16, bgs::kSsFloodKT>), dim3(W64, count), dim3(1024), 0, s, mbits, rbits, rows, W64, fl, k);
  else
    hipLaunchKernelGGL(bgs::ss_flood_kernel, tile_grid, dim3(bgs::kBlock), 0, s, mbits, rbits, rows, W64, fl, k);
}

int ss_launch_refresh(bgs_engine* e, const bgs::SsArgs& a, size_t N, int count, int mode, hipStream_t s) {
  const bool fast = mode == 0 && a.pixelMajor && a.nS > bgs::kSsBatch;
  const dim3 grid(fast ? (unsigned)(((N + 15) / 16 + bgs::kSsRefreshGroups - 1) / bgs::kSsRefreshGroups) : std::min<unsigned>(blocks_for(N), 512u), 1, count), block(bgs::kBlock);
  if (e->ch == 3) {
    if (fast) hipLaunchKernelGGL(bgs::ss_lastrec_pack_kernel, dim3(blocks_for(N), 1, count), block, 0, s, a);
    if (fast) hipLaunchKernelGGL((bgs::ss_refresh_kernel<3, true>), grid, block, 0, s, a, mode);
    else hipLaunchKernelGGL((bgs::ss_refresh_kernel<3, false>), grid, block, 0, s, a, mode);
  } else {
    if (fast) hipLaunchKernelGGL((bgs::ss_refresh_kernel<1, true>), grid, block, 0, s, a, mode);
    else hipLaunchKernelGGL((bgs::ss_refresh_kernel<1, false>), grid, block, 0, s, a, mode);
  }
  return BGS_OK;
}

int ss_init_streams(bgs_engine* e, int first, int count, const uint8_t* d_frames, hipStream_t s) {
  SsDevice* d = e->ss;
  const size_t N = e->n, off = N * first, npix = N * count, C = (size_t)e->ch;
  uint8_t lut[256];
  ss_initial_lut(e->p, e->ch, lut);
  bgs::SsScalars sc0{};
  sc0.autoReset = d->lrScaling, sc0.capLo = d->capLo0, sc0.capHi = d->capHi0;
  bgs::SsLut256 lutv;
  std::memcpy(lutv.v, lut, 256);
  hipLaunchKernelGGL(bgs::ss_init_consts_kernel, dim3(count), dim3(bgs::kBlock), 0, s, d->lut, d->sc, lutv, sc0, first);
  for (int i = first; i < first + count; ++i) d->pp[i] = 0;
  auto fillf = [&](int idx, float v) -> hipError_t {
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    return hipMemsetD32Async((hipDeviceptr_t)(d->f32[idx] + off), (int)bits, npix, s);
  };
  HIP_TRY(fillf(SS_T, d->capLo0));  // m_oUpdateRateFrame = lower cap (:142)
  HIP_TRY(fillf(SS_R, 1.0f));
  HIP_TRY(fillf(SS_V, 10.0f));
  for (int idx : {SS_DLAST0, SS_DLAST1, SS_DMINLT, SS_DMINST, SS_RAWLT, SS_RAWST0, SS_RAWST1, SS_FINLT, SS_FINST}) HIP_TRY(fillf(idx, 0.0f));
  for (int idx : {SS_UNSTABLE, SS_BLINKS, SS_LASTFG, SS_LASTRAW, SS_LASTRAWBLINK, SS_LASTDILINV}) HIP_TRY(hipMemsetAsync(d->u8[idx] + off, 0, npix, s));
  const size_t dsn = (size_t)(e->rows / 8) * (e->cols / 8) * C;
  if (dsn) {
    HIP_TRY(hipMemsetAsync(d->dsLT + dsn * first, 0, dsn * count * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(d->dsST + dsn * first, 0, dsn * count * sizeof(float), s));
  }
  // BGR: the full refresh below writes every record of these streams, zeros included (kernel_subsense.h: ss_refresh_kernel, mode 0) -
  // clearing the 13 GB of an 8 x 1080p model first took 3 ms beside the refresh's 5
  const bool refresh_fills = C == 3 && d->pixelMajor && e->p.subsense_n_samples > bgs::kSsBatch;
  if (!refresh_fills) HIP_TRY(hipMemsetAsync((uint8_t*)d->samples + off * (size_t)d->nSpad * (C == 3 ? 16 : 4), 0, npix * (size_t)d->nSpad * (C == 3 ? 16 : 4), s));
  // first-frame descriptors (:229-243) with the initial LUT, border = 0; LastColor interior = frame
  bgs::LbspArgs la{};
  la.img = d_frames, la.desc = d->lastDesc + off * C, la.rows = e->rows, la.cols = e->cols;
  std::memcpy(la.lut, lut, 256);
  const dim3 lgrid((e->cols + bgs::kLbspTW - 1) / bgs::kLbspTW, (e->rows + bgs::kLbspTH - 1) / bgs::kLbspTH, count);
  if (e->ch == 3)
    hipLaunchKernelGGL((bgs::lbsp_kernel<3>), lgrid, dim3(bgs::kBlock), 0, s, la);
  else
    hipLaunchKernelGGL((bgs::lbsp_kernel<1>), lgrid, dim3(bgs::kBlock), 0, s, la);
  bgs::SsArgs a{};
  ss_fill_args(e, a, first, 0, 0);
  a.frame = d_frames;
  SS_LAUNCH(ss_init_lastcolor_kernel, dim3(blocks_for(N), 1, count), dim3(bgs::kBlock), s, a);
  ss_launch_refresh(e, a, N, count, 0, s);  // refreshModel(1.0f) (:246)
  HIP_TRY(hipGetLastError());
  return BGS_OK;
}

// the side stream phase B runs on, and the ring of events that tie it to the callers' streams
int ss_side_stream(SsDevice* d) {
  if (d->side) return BGS_OK;
  // (round 4: the lowest stream priority for it changed nothing - neither while phase B held every wave slot of the CUs nor with four
  // workgroups per CU: 2.69 / 1.58-1.59 ms per 8 x 1080p step either way)
  HIP_TRY(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
  for (int i = 0; i < SsDevice::kRing; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&d->evA[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d->evB[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d->evTok[i], hipEventDisableTiming));
  }
  return BGS_OK;
}

// the smallest launch of phase A that takes part in the token / is a part of its own: 2^20 pixels (4 096 workgroups; BGS_SS_PART_MIN_PIXELS:
// the tests set 1 so that small frames take the same paths)
size_t ss_part_min_pixels() {
  static const size_t v = getenv("BGS_SS_PART_MIN_PIXELS") ? (size_t)std::max(1ll, atoll(getenv("BGS_SS_PART_MIN_PIXELS"))) : (size_t)1 << 20;
  return v;
}

// streams [first, first + count) on HIP stream s; ss_process (below) decides how a batch is cut into such calls
int ss_process_range(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, hipStream_t s, int64_t t) {
  SsDevice* d = e->ss;
  const size_t N = e->n, off = N * first, npix = N * count;
  // (frames need no particular alignment: the tile loaders read dwords relative to each image's own base, which is unaligned anyway for
  // every second stream of a batch whose rows*cols*channels is odd - global dword loads may be unaligned on this hardware)
  if (t == 0) {  // SuBSENSE.cpp:27-36: construct + initialize on the first frame, then fall through to operator()
    int rc = ss_init_streams(e, first, count, d_frames, s);
    if (rc) return rc;
  }
  const int cur = d->pp[first];
  for (int i = first; i < first + count; ++i)
    if (d->pp[i] != cur) return fail(BGS_ERR_INVALID, "streams %d and %d are not in lock-step", first, i);
  bgs::SsArgs a{};
  ss_fill_args(e, a, first, cur, (unsigned)(t + 1));
  a.frame = d_frames, a.fg = d_fg, a.bgimg = d_bg;
  const dim3 tilesB((e->cols + bgs::kSsTW - 1) / bgs::kSsTW, (e->rows + bgs::kSsBTH - 1) / bgs::kSsBTH, count), block(bgs::kBlock);
  // BGS_SS_FEEDBACK_SPLIT=1: the rules behind the loop as ss_feedback_kernel in front of phase B instead of stage 3 of phase A
  // (identical results; measured slower, kernel_subsense.h - kept as an A/B knob)
  static const bool split = getenv("BGS_SS_FEEDBACK_SPLIT") && atoi(getenv("BGS_SS_FEEDBACK_SPLIT")) == 1;
  static const bool overlap = !(getenv("BGS_SS_OVERLAP") && atoi(getenv("BGS_SS_OVERLAP")) == 0);
  if (overlap) {
    const int rc = ss_side_stream(d);
    if (rc != BGS_OK) return rc;
  }
  const int slot = d->ring;  // this call's {evA, evB, evTok}
  d->ring = (d->ring + 1) % SsDevice::kRing;
  // The phase A token (round 4; OFF by default - measured, no gain).  Phase A is bound by vector issue, everything behind it (phase
  // B's scattered writes, the post-processing chain's many small launches) by DRAM and by latency: a step is A followed by a tail that
  // leaves the vector units idle.  When the cameras are driven as several stream ranges on HIP streams of their own
  // (bgs_process_range_device), one range's tail can run beside another range's phase A - but left to themselves the ranges stay
  // aligned (round 2).  With the token, large launches of phase A take turns: each waits for the phase A of the call
  // before it, on whatever HIP stream that was, so that A0 A1 A0' A1' ... run back to back and every tail beside the other range's
  // phase A.  Measured on 8 x 1080p, fresh-noise frames (profiles/r04_subsense_token_parts.txt): two ranges on two streams WITHOUT the
  // token 2.56 ms young / 1.50-1.57 aged (one batch call: 2.82 / 1.60-1.67), with it 2.67 / 1.55-1.62; four ranges are slower either
  // way (3.1-3.3 / 2.0-2.2).  Phase A beside another range's tail runs 25 % longer (1.2 -> 1.5 ms per 8 frames: it is not as purely
  // issue-bound as its counters suggest), and a tail of ~25 launches does not shrink with its range.  BGS_SS_A_TOKEN=1: on (A/B knob).
  static const bool token_on = (getenv("BGS_SS_A_TOKEN") && atoi(getenv("BGS_SS_A_TOKEN")) == 1) || (getenv("BGS_SS_PARTS") && atoi(getenv("BGS_SS_PARTS")) > 1);  // (the parts of a batch rely on it)
  const bool token = token_on && overlap && npix >= ss_part_min_pixels();
  if (token && d->tok >= 0 && d->tokStream != s) HIP_TRY(hipStreamWaitEvent(s, d->evTok[d->tok], 0));
  {
    Timed tm(e, s, "ss_phase_a_kernel");
    // BGS_SS_QUEUE=1: BGR frames through the per-wave candidate list (kernel_subsense.h "rounds"; identical results; measured slower: the
    // candidates a wave holds per trip are too few to fill its lanes) - A/B knob
    static const bool queue = getenv("BGS_SS_QUEUE") && atoi(getenv("BGS_SS_QUEUE")) == 1;
    const int ath = (e->ch == 3 && queue && !split) ? bgs::kSsQATH : bgs::kSsATH;
    const dim3 tilesA((e->cols + bgs::kSsTW - 1) / bgs::kSsTW, (e->rows + ath - 1) / ath, count);
    if (e->ch == 3) {
      if (split) hipLaunchKernelGGL((bgs::ss_phase_a_kernel<3, true, false>), tilesA, block, 0, s, a);
      else if (queue) hipLaunchKernelGGL((bgs::ss_phase_a_kernel<3, false, true>), tilesA, block, 0, s, a);
      else hipLaunchKernelGGL((bgs::ss_phase_a_kernel<3, false, false>), tilesA, block, 0, s, a);
    } else {
      if (split) hipLaunchKernelGGL((bgs::ss_phase_a_kernel<1, true, false>), tilesA, block, 0, s, a);
      else hipLaunchKernelGGL((bgs::ss_phase_a_kernel<1, false, false>), tilesA, block, 0, s, a);
    }
  }
  if (token) {
    HIP_TRY(hipEventRecord(d->evTok[slot], s));
    d->tok = slot, d->tokStream = s;
  }
  // Phase B (the scattered sample writes) and the post-processing chain both depend on phase A only, and the next frame depends
  // on both: phase B goes to a side stream and rejoins at the end, so its memory-bound scatter overlaps the LDS-bound morphology.
  // (One side stream for every call: phase B launches then run in the order of their calls, which is the order the token gives
  // their phase A launches anyway.)
  // Round 4: the per-pixel rules behind the loop (:498-576) are ss_feedback_kernel; it produces the update requests phase B applies, so
  // it goes in front of phase B on the side stream - both beside the post-processing chain, which only needs phase A's `raw`.
  const dim3 gridF((e->cols + bgs::kBlock - 1) / bgs::kBlock, e->rows, count);
  // WHERE phase B starts.  The timeline of an 8 x 1080p step on the aged model (round 4, profiles/r04_subsense_step_timeline.txt):
  // phase B alone 368 us, the chain alone 288 us, both together 600 - 640 us - hardly better than one after the other.  Beside phase
  // B's ~8 M scattered 16-byte writes (DRAM row activations) every launch of the chain takes four to five times as long
  // and phase B itself 10 - 20 % longer.  Tried: the lowest stream priority for the side stream (no change); 256-lane instead of
  // 1024-lane workgroups for the flood strips (kept; 250 -> 224 us beside phase B, 43 us alone either way); and phase B started
  // BEHIND the flood fill, so that the chain's latency-bound first half runs alone (BGS_SS_B_LATE=1): the second half (median, box
  // filters, byte maps) then takes the slowdown instead - median 21 -> 170 - 250 us - and the step is the same or 2 % longer.  The
  // two share DRAM, not compute; the default stays phase B right behind phase A.
  static const bool b_early = !(getenv("BGS_SS_B_LATE") && atoi(getenv("BGS_SS_B_LATE")) == 1);
  // Phase B beside the chain is held to 4 workgroups per CU by unused dynamic LDS (round 4).  Its workgroups (17 KB of LDS, 4 waves) fit
  // eight to a CU = every wave slot of the CU, and each lives long (scattered 16-byte writes): the chain's small launches on the
  // caller's stream then wait for wave slots - 50-130 us each instead of 5 (profiles/r04_subsense_step_timeline.txt) - and the chain
  // makes no progress until phase B is done.  Phase B is bound by DRAM row activations, not by its resident waves: with four workgroups
  // per CU it takes as long, and the chain gets through beside it: 8 x 1080p step 2.75 -> 2.67 ms young, 1.63 -> 1.56 aged (same box,
  // alternating processes; 3 / 2 per CU the same, 1 per CU slower: profiles/r04_subsense_phase_b_occupancy.txt).  The same limit through
  // the register allocation (__attribute__((amdgpu_waves_per_eu(1, 4))): the kernel descriptor then claims 97 VGPRs, no LDS taken) measured
  // 1-2 % slower than the pad, not faster.  BGS_SS_B_LDS_PAD=bytes
  static const unsigned b_lds_pad = getenv("BGS_SS_B_LDS_PAD") ? (unsigned)std::max(0, std::min(140000, atoi(getenv("BGS_SS_B_LDS_PAD")))) : bgs::kSsBLdsPad;
  auto launch_b = [&]() -> int {
    if (overlap) {
      HIP_TRY(hipEventRecord(d->evA[slot], s));
      HIP_TRY(hipStreamWaitEvent(d->side, d->evA[slot], 0));
      if (split) SS_LAUNCH(ss_feedback_kernel, gridF, block, d->side, a);
      SS_LAUNCH_LDS(ss_phase_b_kernel, tilesB, block, b_lds_pad, d->side, a);
      HIP_TRY(hipEventRecord(d->evB[slot], d->side));
    } else {
      if (split) SS_LAUNCH(ss_feedback_kernel, gridF, block, s, a);
      SS_LAUNCH(ss_phase_b_kernel, tilesB, block, s, a);
    }
    return BGS_OK;
  };
  if (b_early || !overlap) {
    const int rc = launch_b();
    if (rc != BGS_OK) return rc;
  }
  // byte maps of this launch as vectors when the pixel count and the caller's buffers allow it (the engine's own planes are 256-byte aligned)
  const bool v16 = npix % 16 == 0 && (off % 16) == 0, v4 = npix % 4 == 0 && (off % 4) == 0 && e->cols % 4 == 0 && (!d_fg || aligned(d_fg, 4));
  auto launch_blink = [&]() {  // :624-627; only reads phase A's `raw`
    if (v16)
      hipLaunchKernelGGL(bgs::ss_blink_kernel<16>, dim3(blocks_for(npix / 16)), block, 0, s, a, npix);
    else
      hipLaunchKernelGGL(bgs::ss_blink_kernel<1>, dim3(blocks_for(npix)), block, 0, s, a, npix);
  };
  if (b_early || !overlap) launch_blink();
  uint8_t* raw = d->u8[SS_RAW] + off;
  uint8_t* lastFG = d->u8[SS_LASTFG] + off;
  // :628-636 on bit planes (kernel_subsense.h): a lane owns 64 pixels of a row
  const int W64 = (e->cols + 63) / 64, tilesY = (e->rows + 63) / 64;
  const size_t wpi = (size_t)e->rows * W64, nwords = wpi * count, plane = wpi * e->S;  // words per image / of this launch / per plane
  uint64_t* bw = d->bitws + (size_t)first * wpi;
  uint64_t *b_raw = bw, *b_tmp = bw + plane, *b_pre = bw + 2 * plane, *b_cur = bw + 3 * plane, *b_fg = bw + 4 * plane, *b_dil = bw + 5 * plane;
  uint64_t* mbits = d->mbits + (size_t)first * wpi;
  uint64_t* rbits = d->rbits + (size_t)first * wpi;
  const dim3 wgrid(blocks_for(nwords));
  auto pack = [&](const uint8_t* src, uint64_t* dst) {
    if (e->cols % 16 == 0 && aligned(src, 16))
      hipLaunchKernelGGL(bgs::ss_bits_pack_kernel<16>, dim3(blocks_for(nwords * 4)), block, 0, s, src, dst, e->rows, e->cols, W64, nwords);
    else
      hipLaunchKernelGGL(bgs::ss_bits_pack_kernel<1>, dim3(blocks_for(nwords * bgs::kWave)), block, 0, s, src, dst, e->rows, e->cols, W64, nwords);
  };
  pack(raw, b_raw);
  // morphologyEx(MORPH_CLOSE) :628  -> b_pre = PreFlood
  hipLaunchKernelGGL((bgs::ss_bits_box_kernel<1, 1>), wgrid, block, 0, s, (const uint64_t*)b_raw, b_tmp, e->rows, e->cols, W64, nwords);
  hipLaunchKernelGGL((bgs::ss_bits_box_kernel<0, 1>), wgrid, block, 0, s, (const uint64_t*)b_tmp, b_pre, e->rows, e->cols, W64, nwords);
  // floodFill(PreFlood copy, (0,0), 255) :629-630 -> rbits = reached set
  hipLaunchKernelGGL(bgs::ss_bits_flood_prepare_kernel, wgrid, block, 0, s, (const uint64_t*)b_pre, mbits, rbits, e->rows, e->cols, W64, nwords);
  hipLaunchKernelGGL(bgs::ss_flood_seed_kernel, dim3(count), block, 0, s, (const uint64_t*)mbits, rbits, e->rows, e->cols, W64);
  const dim3 fgrid(blocks_for((size_t)tilesY * W64 * bgs::kWave), 1, count);
  int* fl = d->flood_flags + (size_t)first * bgs::kSsFloodFlags;
  HIP_TRY(hipMemsetAsync(fl, 0, (size_t)count * bgs::kSsFloodFlags * sizeof(int), s));
  static const int batch = getenv("BGS_SS_FLOOD_BATCH") ? std::max(0, std::min(bgs::kSsFloodBatch, atoi(getenv("BGS_SS_FLOOD_BATCH")))) : bgs::kSsFloodBatch;  // test knob: 0/1 force the finish kernel to do the work
  // (the finish kernel almost always returns at once; 256 lanes so that it does not wait for a quarter of a CU beside phase B - see ss_flood_strip_kernel)
  static const bool wide_finish = getenv("BGS_SS_FLOOD_WG1024") && atoi(getenv("BGS_SS_FLOOD_WG1024")) == 1;
  for (int k = 0; k < batch; ++k) ss_launch_flood(fgrid, count, tilesY, s, mbits, rbits, e->rows, W64, fl, k);
  hipLaunchKernelGGL(bgs::ss_flood_finish_kernel, dim3(count), dim3(wide_finish ? 1024 : 256), 0, s, (const uint64_t*)mbits, rbits, e->rows, W64, fl, batch);
  if (!(b_early || !overlap)) {
    const int rc = launch_b();
    if (rc != BGS_OK) return rc;
    launch_blink();
  }
  // erode x3 :632 = one 7x7 box -> b_tmp;  :631-634 -> b_cur
  hipLaunchKernelGGL((bgs::ss_bits_box_kernel<0, 3>), wgrid, block, 0, s, (const uint64_t*)b_pre, b_tmp, e->rows, e->cols, W64, nwords);
  hipLaunchKernelGGL(bgs::ss_bits_combine_kernel, wgrid, block, 0, s, (const uint64_t*)b_raw, (const uint64_t*)b_pre, (const uint64_t*)rbits, (const uint64_t*)b_tmp, b_cur, e->cols, W64, nwords);
  // medianBlur :635 of the binary mask -> the byte map phase A reads (lastFG) and its bit plane (b_fg): bit-sliced on the planes
  // (round 3); BGS_SS_MEDIAN_BITS=0: round 2's counts in LDS (morph_box_kernel) + a pack launch
  static const bool median_bits = !(getenv("BGS_SS_MEDIAN_BITS") && atoi(getenv("BGS_SS_MEDIAN_BITS")) == 0);
  if (median_bits && d->medK >= 3 && d->medK <= 13) {
    switch (d->medK / 2) {
#define SS_MEDIAN_CASE(RV) \
  case RV: hipLaunchKernelGGL((bgs::ss_bits_median_kernel<RV>), wgrid, block, 0, s, (const uint64_t*)b_cur, b_fg, lastFG, e->rows, e->cols, W64, nwords); break;
      SS_MEDIAN_CASE(1) SS_MEDIAN_CASE(2) SS_MEDIAN_CASE(3) SS_MEDIAN_CASE(4) SS_MEDIAN_CASE(5) SS_MEDIAN_CASE(6)
#undef SS_MEDIAN_CASE
    }
  } else {
    bgs::MorphArgs m{nullptr, lastFG, e->rows, e->cols, 3, d->medK, b_cur, W64};
    bgs::morph_launch(m, count, s);
    pack(lastFG, b_fg);
  }
  hipLaunchKernelGGL((bgs::ss_bits_box_kernel<1, 3>), wgrid, block, 0, s, (const uint64_t*)b_fg, b_dil, e->rows, e->cols, W64, nwords);  // dilate x3 :636
  if (v4)  // :637-642
    hipLaunchKernelGGL(bgs::ss_finish_kernel<4>, dim3(blocks_for(npix / 4)), block, 0, s, a, (const uint64_t*)b_dil, W64, npix);
  else
    hipLaunchKernelGGL(bgs::ss_finish_kernel<1>, dim3(blocks_for(npix)), block, 0, s, a, (const uint64_t*)b_dil, W64, npix);
  if (d->lrScaling) {
    const int dsn = (e->rows / 8) * (e->cols / 8);
    SS_LAUNCH(ss_downsample_kernel, dim3(blocks_for(dsn), 1, count), block, s, a);
  }
  hipLaunchKernelGGL(bgs::ss_frame_level_kernel, dim3(count), dim3(256), 0, s, a);
  if (overlap) HIP_TRY(hipStreamWaitEvent(s, d->evB[slot], 0));  // the refresh below and the next frame need phase B's writes
  ss_launch_refresh(e, a, N, count, 1, s);  // refreshModel(0.1f) if asked (:680)
  if (d_bg) SS_LAUNCH(ss_background_kernel, dim3(blocks_for(N * e->ch), 1, count), block, s, a);
  HIP_TRY(hipGetLastError());
  for (int i = first; i < first + count; ++i) d->pp[i] = (uint8_t)(cur ^ 1);
  return BGS_OK;
}

// One batch = SuBSENSEBGS::process for streams [first, first + count).  BGS_SS_PARTS=n (default 1: off - measured slower) cuts a large
// batch into up to SsDevice::kParts parts, part 0 on the caller's stream, the others on streams of the engine, each an
// ss_process_range call of its own with the phase A token, so that each part's tail (phase B, post-processing) runs beside the next
// part's phase A and only the last part's tail is left standing alone; the caller's stream continues behind all of it.  8 x 1080p:
// 2 parts 3.01 ms young / 2.24 aged, 4 parts 3.99 / 2.30 against 2.82 / 1.60-1.67 as one launch (profiles/r04_subsense_token_parts.txt;
// same reasons as the token's).  Same kernels on the same per-stream state: results do not depend on the cut (tests: 1, 2, 4 parts in
// child processes).
int ss_process(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, hipStream_t s, int64_t t) {
  SsDevice* d = e->ss;
  static const int parts_env = getenv("BGS_SS_PARTS") ? std::max(1, std::min((int)SsDevice::kParts, atoi(getenv("BGS_SS_PARTS")))) : bgs::kSsParts;
  static const bool overlap = !(getenv("BGS_SS_OVERLAP") && atoi(getenv("BGS_SS_OVERLAP")) == 0);
  const size_t N = e->n, C = (size_t)e->ch;
  const int P = std::min<int>({parts_env, count, (int)std::min<size_t>(N * (size_t)count / ss_part_min_pixels(), (size_t)SsDevice::kParts)});  // a part is a launch that fills the device
  if (t == 0 || !overlap || P < 2) return ss_process_range(e, first, count, d_frames, d_fg, d_bg, s, t);  // (the first frame's full refresh is one launch)
  if (!d->evFork) {
    HIP_TRY(hipEventCreateWithFlags(&d->evFork, hipEventDisableTiming));
    for (int p = 1; p < SsDevice::kParts; ++p) {
      HIP_TRY(hipStreamCreateWithFlags(&d->part[p], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&d->evPart[p], hipEventDisableTiming));
    }
  }
  HIP_TRY(hipEventRecord(d->evFork, s));  // the frames are ready, and every earlier call has been joined into s
  for (int p = 0; p < P; ++p) {
    const int f = first + (int)((int64_t)count * p / P), n = first + (int)((int64_t)count * (p + 1) / P) - f;
    const size_t o = (size_t)(f - first) * N;
    hipStream_t ps = p == 0 ? s : d->part[p];
    if (p) HIP_TRY(hipStreamWaitEvent(ps, d->evFork, 0));
    const int rc = ss_process_range(e, f, n, d_frames + o * C, d_fg ? d_fg + o : nullptr, d_bg ? d_bg + o * C : nullptr, ps, t);
    if (rc != BGS_OK) return rc;
    if (p) HIP_TRY(hipEventRecord(d->evPart[p], ps));
  }
  for (int p = 1; p < P; ++p) HIP_TRY(hipStreamWaitEvent(s, d->evPart[p], 0));
  return BGS_OK;
}

int64_t ss_get_state(bgs_engine* e, int stream, const char* plane, void* dst, size_t cap) {
  SsDevice* d = e->ss;
  if (e->algo == BGS_LOBSTER) {  // the LOBSTER model has no feedback maps
    bool ok = false;
    for (const char* nm : {"lastfg", "lastcolor", "lastdesc", "color", "desc", "lut"}) ok = ok || !strcmp(plane, nm);
    if (!ok) return fail(BGS_ERR_STATE, "unknown state plane '%s' for LOBSTER", plane);
  }
  const size_t N = e->n, off = N * stream, nS = (size_t)e->p.subsense_n_samples, C = (size_t)e->ch;
  const int cur = d->pp[stream];
  struct Ent {
    const char* name;
    const void* p;
    size_t bytes;
  };
  const Ent tab[] = {{"R", d->f32[SS_R] + off, N * 4},
                     {"V", d->f32[SS_V] + off, N * 4},
                     {"T", d->f32[SS_T] + off, N * 4},
                     {"Dlast", d->f32[cur ? SS_DLAST1 : SS_DLAST0] + off, N * 4},
                     {"DminLT", d->f32[SS_DMINLT] + off, N * 4},
                     {"DminST", d->f32[SS_DMINST] + off, N * 4},
                     {"RawLT", d->f32[SS_RAWLT] + off, N * 4},
                     {"RawST", d->f32[cur ? SS_RAWST1 : SS_RAWST0] + off, N * 4},
                     {"FinLT", d->f32[SS_FINLT] + off, N * 4},
                     {"FinST", d->f32[SS_FINST] + off, N * 4},
                     {"unstable", d->u8[SS_UNSTABLE] + off, N},
                     {"blinks", d->u8[SS_BLINKS] + off, N},
                     {"lastfg", d->u8[SS_LASTFG] + off, N},
                     {"lastraw", d->u8[SS_LASTRAW] + off, N},
                     {"lastcolor", d->lastColor + off * C, N * C},
                     {"lastdesc", d->lastDesc + off * C, N * 2 * C},
                     {"lut", d->lut + (size_t)stream * 256, 256}};
  for (const Ent& t : tab)
    if (!strcmp(plane, t.name)) {
      if (cap < t.bytes) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
      if (d2h_staged(dst, t.p, t.bytes) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
      return (int64_t)t.bytes;
    }
  if (!strcmp(plane, "color") || !strcmp(plane, "desc")) {  // canonical export: color u8 [nS][N][C], desc u16 [nS][N][C], whatever the record layout
    const bool wantColor = !strcmp(plane, "color");
    const size_t need = N * nS * C * (wantColor ? 1 : 2), recB = C == 3 ? 16 : 4;
    if (cap < need) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
    const size_t per = d->pixelMajor ? (size_t)d->nSpad : nS;  // records per pixel held on the device
    std::vector<uint8_t> recs(N * per * recB);
    if (d2h_staged(recs.data(), (const uint8_t*)d->samples + off * per * recB, recs.size()) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    for (size_t k = 0; k < nS; ++k)
      for (size_t px = 0; px < N; ++px) {  // export: [nS][N][C], whatever the device order
        const size_t B = bgs::kSsBatch;  // same mapping as ss_rec (kernel_subsense.h)
        const uint8_t* r = recs.data() + (!d->pixelMajor ? k * N + px : k < B ? k * N + px : B * N + px * (per - B) + (k - B)) * recB;
        const size_t o = k * N + px;
        for (size_t c = 0; c < C; ++c) {
          if (wantColor)
            ((uint8_t*)dst)[o * C + c] = r[c];
          else
            std::memcpy((uint8_t*)dst + (o * C + c) * 2, r + (C == 3 ? 4 + 2 * c : 2), 2);
        }
      }
    return (int64_t)need;
  }
  if (!strcmp(plane, "floodflags")) {  // diagnostics: which launches of the last flood fill still changed something, [kSsFloodFlags] int32; the last one = the finish kernel had to work
    if (e->algo != BGS_SUBSENSE) return fail(BGS_ERR_STATE, "floodflags: SuBSENSE only");
    const size_t nb = (size_t)bgs::kSsFloodFlags * sizeof(int);
    if (cap < nb) return fail(BGS_ERR_STATE, "buffer too small for plane floodflags");
    if (d2h_staged(dst, d->flood_flags + (size_t)stream * bgs::kSsFloodFlags, nb) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    return (int64_t)nb;
  }
  if (!strcmp(plane, "magic") && d->magic) {  // ss_mod's table (kernel_subsense.h), for the test that checks it against plain integer division
    if (cap < bgs::kSsMagicN * 4) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
    if (d2h_staged(dst, d->magic, bgs::kSsMagicN * 4) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    return bgs::kSsMagicN * 4;
  }
  if (!strcmp(plane, "scalars")) {
    if (cap < 7 * sizeof(double)) return fail(BGS_ERR_STATE, "buffer too small for plane scalars");
    bgs::SsScalars sc;
    if (d2h_staged(&sc, d->sc + stream, sizeof(sc)) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    double* o = (double*)dst;
    o[0] = (double)e->seen[stream], o[1] = sc.framesSinceReset, o[2] = sc.cooldown, o[3] = sc.capLo, o[4] = sc.capHi, o[5] = sc.autoReset, o[6] = sc.lastNZ;
    return 7 * sizeof(double);
  }
  return fail(BGS_ERR_STATE, "unknown state plane '%s' for SuBSENSE", plane);
}

// ---------------------------------------------------------------------------------------------------------------- LOBSTER
// LOBSTERBGS::process (package_bgs/pl/LOBSTER.cpp:20-45): same sample planes as SuBSENSE, no feedback maps.
int lob_allocate(bgs_engine* e) {
  if (e->rows < 5 || e->cols < 5) return fail(BGS_ERR_UNSUPPORTED, "LOBSTER needs at least 5x5 pixels (LBSP::validateROI)");
  const bgs_params& p = e->p;
  if (p.subsense_n_samples < 1 || p.subsense_n_samples > bgs::kSsMaxSamples || p.subsense_n_required > p.subsense_n_samples)
    return fail(BGS_ERR_UNSUPPORTED, "LOBSTER: nBGSamples must be 1..%d and nRequiredBGSamples <= nBGSamples", bgs::kSsMaxSamples);
  SsDevice* d = new SsDevice();
  e->ss = d;
  const size_t N = e->n, P = N * e->S, nS = (size_t)p.subsense_n_samples, C = (size_t)e->ch;
  d->pixelMajor = 0, d->nSpad = (int)nS;  // LOBSTER: sample-major (kernel_subsense.h)
  DMALLOC(d->samples, P * nS * (C == 3 ? 16 : 4));
  DMALLOC(d->lastColor, P * C + 8);      // + 8: ss_refresh_one reads a pixel's 3 bytes / 3 words with one 4- / 8-byte load
  DMALLOC(d->lastDesc, P * C * 2 + 8);
  DMALLOC(d->curColor, P * C);
  DMALLOC(d->curDesc, P * C * 2);
  DMALLOC(d->req, P * 2 * 2);
  DMALLOC(d->lut, (size_t)e->S * 256);
  DMALLOC(d->u8[SS_LASTFG], P);
  DMALLOC(d->u8[SS_RAW], P);
  d->medK = 9;  // DEFAULT_MEDIAN_BLUR_KERNEL_SIZE, BackgroundSubtractorLBSP.cpp:17
  d->pp.assign(e->S, 0);
  return BGS_OK;
}

void lob_fill_args(const bgs_engine* e, bgs::SsArgs& a, int first, unsigned frameIndex) {
  const SsDevice* d = e->ss;
  const bgs_params& p = e->p;
  a.samples = d->samples, a.nSpad = d->nSpad, a.pixelMajor = d->pixelMajor, a.lastColor = d->lastColor, a.lastDesc = d->lastDesc, a.req = d->req, a.lut = d->lut;
  a.lastFG = d->u8[SS_LASTFG], a.raw = d->u8[SS_RAW];
  a.rows = e->rows, a.cols = e->cols, a.nS = p.subsense_n_samples, a.nReq = p.subsense_n_required;
  a.nMinColor = p.subsense_min_color_dist_threshold, a.nDescOff = p.subsense_desc_dist_threshold_offset;
  a.frameIndex = frameIndex, a.first = first;
  static const int refill = getenv("BGS_LOB_REFILL") ? std::max(1, std::min(64, atoi(getenv("BGS_LOB_REFILL")))) : bgs::kSsRefill;  // tuning knob (lob_phase_a_queue_kernel)
  a.refill = refill;
}

int lob_process(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, hipStream_t s, int64_t t) {
  SsDevice* d = e->ss;
  const size_t N = e->n, off = N * first, npix = N * count, nS = (size_t)e->p.subsense_n_samples, C = (size_t)e->ch;
  const dim3 block(bgs::kBlock);
  if (t == 0) {  // LOBSTER.cpp:27-34: construct + initialize on the first frame, then fall through to operator()
    uint8_t lut[256];
    for (int v = 0; v < 256; ++v) {  // BackgroundSubtractorLOBSTER.cpp:85-86 (gray: the sum / 2), :103-104
      float f = (float)(size_t)v * e->p.lbsp_rel_threshold + (float)(size_t)e->p.lbsp_threshold_offset;
      if (e->ch == 1) f = f / 2;
      const long r = std::lrint((double)f);
      lut[v] = (uint8_t)std::min<long>(std::max<long>(r, 0), 255);
    }
    bgs::SsLut256 lutv;
    std::memcpy(lutv.v, lut, 256);
    hipLaunchKernelGGL(bgs::ss_init_consts_kernel, dim3(count), block, 0, s, d->lut, (bgs::SsScalars*)nullptr, lutv, bgs::SsScalars{}, first);
    HIP_TRY(hipMemsetAsync(d->u8[SS_LASTFG] + off, 0, npix, s));
    HIP_TRY(hipMemsetAsync((uint8_t*)d->samples + off * nS * (C == 3 ? 16 : 4), 0, npix * nS * (C == 3 ? 16 : 4), s));
    bgs::LbspArgs la{};
    la.img = d_frames, la.desc = d->lastDesc + off * C, la.rows = e->rows, la.cols = e->cols;
    std::memcpy(la.lut, lut, 256);
    const dim3 lgrid((e->cols + bgs::kLbspTW - 1) / bgs::kLbspTW, (e->rows + bgs::kLbspTH - 1) / bgs::kLbspTH, count);
    if (e->ch == 3)
      hipLaunchKernelGGL((bgs::lbsp_kernel<3>), lgrid, block, 0, s, la);
    else
      hipLaunchKernelGGL((bgs::lbsp_kernel<1>), lgrid, block, 0, s, la);
    bgs::SsArgs a0{};
    lob_fill_args(e, a0, first, 0);
    a0.frame = d_frames;
    SS_LAUNCH(ss_init_lastcolor_kernel, dim3(blocks_for(N), 1, count), block, s, a0);
    ss_launch_refresh(e, a0, N, count, 0, s);  // refreshModel(1.0f), :120
  }
  bgs::SsArgs a{};
  lob_fill_args(e, a, first, (unsigned)(t + 1));
  a.frame = d_frames, a.fg = d_fg, a.bgimg = d_bg;
  a.lastColor = d->curColor, a.lastDesc = d->curDesc;  // phase A writes / phase B reads what a requesting pixel copies into the model
  const dim3 tiles((e->cols + bgs::kSsTW - 1) / bgs::kSsTW, (e->rows + bgs::kSsTH - 1) / bgs::kSsTH, count);
  const dim3 tilesB((e->cols + bgs::kSsTW - 1) / bgs::kSsTW, (e->rows + bgs::kSsBTH - 1) / bgs::kSsBTH, count);
  {
    // round 4: lanes fed from a queue (kernel_subsense.h); BGS_LOB_QUEUE=0: one pixel per lane in lock step (rounds 1-3; A/B and test knob)
    static const bool queue = !(getenv("BGS_LOB_QUEUE") && atoi(getenv("BGS_LOB_QUEUE")) == 0);
    Timed tm(e, s, "lob_phase_a_kernel");
    if (queue) {
      const dim3 tilesQ((e->cols + bgs::kSsTW - 1) / bgs::kSsTW, (e->rows + bgs::kLobATH - 1) / bgs::kLobATH, count);
      SS_LAUNCH(lob_phase_a_queue_kernel, tilesQ, block, s, a);
    } else {
      SS_LAUNCH(lob_phase_a_kernel, tiles, block, s, a);
    }
  }
  // (Phase B beside the median on the side stream, as in SuBSENSE's step, was tried at the end of round 4: LOBSTER's tail is only 0.15-0.2
  // ms and the two event hops cost more than the overlap gives - 8 x 1080p 2.39 against 2.34 ms on S_surv, 0.64 against 0.59 on smooth input.)
  SS_LAUNCH(ss_phase_b_kernel, tilesB, block, s, a);
  uint8_t* lastFG = d->u8[SS_LASTFG] + off;
  ss_morph(d->u8[SS_RAW] + off, lastFG, e->rows, e->cols, count, 3, d->medK, s);  // cv::medianBlur(oCurrFGMask, m_oLastFGMask, 9) :281
  if (d_fg) HIP_TRY(hipMemcpyAsync(d_fg, lastFG, npix, hipMemcpyDeviceToDevice, s));  // :282
  if (d_bg) SS_LAUNCH(ss_background_kernel, dim3(blocks_for(N * e->ch), 1, count), block, s, a);  // getBackgroundImage :286-303
  HIP_TRY(hipGetLastError());
  return BGS_OK;
}

void ss_free(bgs_engine* e) {
#ifdef BGS_SS_STATS
  {
    unsigned long long h[16] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bgs::g_ss_stats), sizeof h) == hipSuccess)
      fprintf(stderr, "ss_stats trips %llu active_lanes %llu refills %llu | R narrow %llu lanes %llu wide %llu lanes %llu | passes %llu lanes %llu | put off %llu lanes %llu\n", h[0], h[1], h[2], h[3], h[4], h[5],
              h[6], h[7], h[8], h[9], h[10]);
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(bgs::g_ss_stats), z, sizeof z);
  }
#endif
  if (e->ss) {
    e->ss->release();
    delete e->ss;
    e->ss = nullptr;
  }
}
