// bgs_hip.hip — the C ABI of libbgs_hip (include/bgs_hip.h) and the engine behind it.
//
// The engine is the device-side twin of one reference IBGS object per stream: it owns the model state of
// n_streams independent streams as SoA planes in HBM and turns every IBGS::process call
// (package_bgs/IBGS.h:24) into ONE fused kernel launch.  Host entry points stage through pinned memory;
// device entry points take HBM pointers and launch over streams x pixels.  There is no CPU fallback: if HIP
// is unavailable every compute entry point fails with BGS_ERR_HIP.
#include "../../include/bgs_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <atomic>
#include <vector>

#include "kernel_gmg.h"
#include "kernel_ingest.h"
#include "kernel_cc.h"
#include "kernel_dp.h"
#include "kernel_mog1.h"
#include "kernel_mog2.h"
#include "kernel_pointwise.h"
#include "kernel_stencil.h"
#include "kernel_subsense.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) return fail(BGS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Device -> pageable host memory (bgs_get_state, the blob lists): through a page-locked bounce buffer of the library's own, never
// hipMemcpy straight into the caller's memory.  For copies past a size threshold the HIP runtime page-locks the caller's buffer on the
// fly and KEEPS the last eight such mappings, looked up by host ADDRESS (ROCclr: hsaCopyStagedOrPinned -> pinHostMemory / addPinnedMem
// / findPinnedMem).  A caller that frees such a buffer and gets the same address back from the next allocation - numpy does, for every
// array past malloc's mmap threshold - is served the stale mapping, and the copy engine then writes through pages that are gone:
// round 4's GPU suite died twice that way (rocr::core::Runtime::VMFaultHandler -> abort while a test read a model plane back,
// profiles/r04_vmfault_backtrace.txt), in different tests, with no kernel of this library running.  A page-locked destination takes
// the direct path.  One buffer per process, serialised: these are diagnostics and small result lists, not the frame path (bgs_process
// stages through the engine's own page-locked images).
int d2h_staged(void* dst, const void* src, size_t bytes) {
  static std::mutex mu;
  static uint8_t* bounce = nullptr;
  constexpr size_t kBounce = (size_t)4 << 20;
  std::lock_guard<std::mutex> lk(mu);
  if (!bounce && hipHostMalloc((void**)&bounce, kBounce, hipHostMallocPortable) != hipSuccess) {
    bounce = nullptr;
    return fail(BGS_ERR_HIP, "hipHostMalloc of the read-back buffer failed");
  }
  for (size_t o = 0; o < bytes; o += kBounce) {
    const size_t n = std::min(kBounce, bytes - o);
    if (hipMemcpy(bounce, (const uint8_t*)src + o, n, hipMemcpyDeviceToHost) != hipSuccess) return fail(BGS_ERR_HIP, "hipMemcpy (device to host) failed");
    std::memcpy((uint8_t*)dst + o, bounce, n);
  }
  return BGS_OK;
}
inline unsigned blocks_for(size_t groups) { return (unsigned)((groups + bgs::kBlock - 1) / bgs::kBlock); }

}  // namespace

namespace {
struct SsDevice;  // engine_subsense.h
}


// One virtual address range backed by separately created physical chunks (hipMemAddressReserve + hipMemCreate x n + hipMemMap): how
// the big models are placed (model_allocate) and what bgs_calibrate_copy measures beside a plain hipMalloc.
struct VmmRange {
  std::vector<hipMemGenericAllocationHandle_t> handles;
  void* base = nullptr;
  size_t bytes = 0, chunk = 0, mapped = 0;  // mapped: chunks actually mapped (a failure half-way leaves fewer than handles)
};

struct bgs_engine {
  bgs_algo algo;
  bgs_params p;
  int device = 0;
  int S = 1;
  int rows = 0, cols = 0, ch = 0;
  size_t n = 0;  // pixels per stream; 0 until the geometry is known
  std::vector<int64_t> seen, counter;
  std::vector<int64_t> rpos;         // frame history ring: where a stream's NEXT frame goes (ring[rpos % nring]); survives bgs_reset_stream, so that streams fed in the same calls keep sharing launches
  std::vector<uint32_t> last_flags;  // out_flags of each stream's last frame (bgs_stream_flags)

  // frame history ring (FD: 2 slots, WMM/WMV: 3): frame t of stream s lives in ring[t % nring] + s*n*ch
  uint8_t* ring[3] = {nullptr, nullptr, nullptr};
  int nring = 0;
  bool borrow = false;               // device path: history = the caller's previous d_frames, no copies
  bool borrow_in_clip = false;       // set by process_clip: frames of the clip serve as history, for the range of that call
  const uint8_t* borrowed[2] = {nullptr, nullptr};
  // byte state (SFD background, ABL/ASBL background): [S][n*state_ch]
  uint8_t* bgstate = nullptr;
  int state_ch = 0;
  uint8_t* abl_lut = nullptr;   // ABL: 256 x 256 table of background bytes for the current alpha (kernel_pointwise.h)
  double abl_lut_alpha = 0;     // the alpha it was built for (ASBL: two tables of 257 rows, learning then detection phase)
  double asbl_lut_alpha[2] = {0, 0};
  bool abl_lut_valid = false;
  int n_cu = 256;
  uint8_t* bgstate2 = nullptr;  // ASBL: second buffer of the ping-pong pair (the 3x3 median reads neighbours' OLD background)
  std::vector<uint8_t> flip;    // ASBL: which buffer holds the current background, per stream
  float* mog1_state = nullptr;  // MOG1 model (kernel_mog1.h, tiled)
  float* dp_state = nullptr;    // package_bgs/dp models (kernel_dp.h): [S][planes][n]
  SsDevice* ss = nullptr;       // SuBSENSE model (engine_subsense.h)
  int2* gmg_rec = nullptr;        // GMG histograms (kernel_gmg.h): {colour, weight} records [F][P]
  uint8_t* gmg_nfeat = nullptr;
  // MOG2 model (kernel_mog2.h: tiles of ranked weights + fixed-slot records + rank->slot meta words)
  uint8_t* mog2_state = nullptr;
  int xcd_swizzle = 1;             // XCD-aware block order (kernel_mog2.h): 0 off, 1 model kernels (MOG2, MOG1, dp), 2 also the byte-stream kernels
  int mog2_sparse = 3;             // data-dependent traffic (kernel_mog2.h): 0 dense (everything loaded and written), 1 only what changed is written, 2 / 4 a lane also loads only the modes its pixel has, 3 = choose 1 or 4 from the scene
  int mog2_complete = 1;           // sector-complete stores (kernel_mog2.h); BGS_MOG2_COMPLETE=0 for A/B runs
  bool clip_fuse = true;           // MOG2 clip calls keep the model in registers across frames (option 7; results identical either way)
  int mog2_sparse_now = 2;         // what auto mode currently runs
  int mog2_sparse_want = 2;        // what the last poll asked for (a switch needs two polls in a row)
  unsigned mog2_launches = 0;      // auto mode: per-frame launches so far (every 16th one samples)
  unsigned* d_stat = nullptr;      // device: {record slots sampled, modes live, records needed after the summaries}
  // pinned copies of the counters, a ring of kStatSlots posts (one per sampling launch, each with its event): a host that runs far
  // ahead of the device still finds the most recent sample that has COMPLETED when it looks
  static constexpr int kStatSlots = 8;
  unsigned* h_stat = nullptr;      // [kStatSlots][3]
  hipEvent_t stat_ev[kStatSlots] = {nullptr};
  bool stat_posted[kStatSlots] = {false};
  unsigned stat_seq = 0;           // posts so far
  int model_chunk_mb = 256;        // big models are built from physical chunks of this size (model_allocate); 0: one plain hipMalloc
  bool poison = false;             // BGS_DEBUG_POISON: every fresh device buffer is filled with 0xA5 (see dmalloc)
  // a model built from fixed-size physical chunks with the virtual memory API (model_allocate)
  VmmRange vmm;
  size_t model_chunk_min_bytes = (size_t)768 << 20;     // models below this take one plain hipMalloc (BGS_MODEL_CHUNK_MIN_MB: test knob)

  // host staging (bgs_process)
  uint8_t *h_in = nullptr, *h_fg = nullptr, *h_bg = nullptr;
  uint8_t *d_in = nullptr, *d_fg = nullptr, *d_bg = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t band_ev[8] = {nullptr};  // bgs_process: one event per output band
  // bgs_process: caller buffers that came back unchanged call after call are page-locked once and DMA'd directly (host_pin)
  struct HostPin {
    const void* ptr = nullptr;   // what the previous call passed for this role
    size_t bytes = 0;
    bool pinned = false, refused = false;
  };
  std::vector<HostPin> pin;       // [stream][3]: 0 input, 1 mask, 2 background (every camera has its own buffers)
  std::vector<std::pair<const uint8_t*, size_t>> arenas;  // bgs_host_arena: caller memory page-locked as a whole; images inside it are DMA'd in place
  // bgs_submit / bgs_wait: one lane per camera - its own HIP stream, staging and device images - so that the uploads, kernels and
  // downloads of different cameras overlap; created at a stream's first bgs_submit
  struct Lane {
    hipStream_t hs = nullptr;
    hipEvent_t done = nullptr;
    uint8_t *h_in = nullptr, *h_fg = nullptr, *h_bg = nullptr, *d_in = nullptr, *d_fg = nullptr, *d_bg = nullptr;
    bool pending = false, fg_direct = false, bg_direct = false;
    uint8_t *fg = nullptr, *bg = nullptr;  // the caller's output images of the pending submission
    size_t fg_step = 0, bg_step = 0;
    uint32_t flags = 0;
  };
  std::vector<Lane> lanes;
  // host-path diagnostics (bgs_get_state "hostpath"; bench.py host_path): what hipHostRegister cost and how often it ran, how long the
  // CPU spent copying images into / out of pinned staging, bytes that crossed the bus each way
  double diag_reg_ms = 0, diag_stage_in_ms = 0, diag_stage_out_ms = 0;
  int64_t diag_reg_calls = 0, diag_unreg_calls = 0, diag_h2d_bytes = 0, diag_d2h_bytes = 0, diag_frames = 0;
  int host_register = 0;          // BGS_OPT_HOST_REGISTER: roles that may be page-locked in place (bit 0 input, 1 mask, 2 background); 0 = always stage
  bool ingest_on = false;             // bgs_set_ingest: bgs_process takes raw frames
  bgs_ingest ingest{};
  int raw_rows = 0, raw_cols = 0;     // geometry of the raw frames (fixed by the first one)
  uint8_t *h_raw = nullptr, *d_raw = nullptr, *d_ingest_ws = nullptr;  // raw staging for configurations that need device work
  uint8_t* pack_fg = nullptr;         // frames of rows*cols % 64 != 0 pixels: the byte masks a packed-only caller's bit masks are made from
  size_t pack_fg_bytes = 0;
  int last_fg_stream = -1;            // bgs_last_mask_blobs: whose mask d_fg holds (-1: none valid)
  void* cc_work = nullptr;            // its device scratch: workspace | boxes | moments | count
  int cc_cap = 0;                     // boxes the scratch has room for

  // dominant-kernel timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  const char* kernel_name = "";
};

namespace {

void ss_free(bgs_engine* e);  // engine_subsense.h
void vmm_free(VmmRange& v);  // below

void lane_release(bgs_engine::Lane& ln) {  // whatever of a lane came to be (bgs_submit's set-up may have failed half-way)
  if (ln.hs) (void)hipStreamSynchronize(ln.hs);
  void* host[] = {ln.h_in, ln.h_fg, ln.h_bg};
  for (void* h : host)
    if (h) (void)hipHostFree(h);
  void* dv[] = {ln.d_in, ln.d_fg, ln.d_bg};
  for (void* d : dv)
    if (d) (void)hipFree(d);
  if (ln.done) (void)hipEventDestroy(ln.done);
  if (ln.hs) (void)hipStreamDestroy(ln.hs);
  ln = bgs_engine::Lane();
}

void free_all(bgs_engine* e) {
  for (auto& r : e->ring)
    if (r) (void)hipFree(r), r = nullptr;
  if (e->abl_lut) (void)hipFree(e->abl_lut), e->abl_lut = nullptr, e->abl_lut_valid = false;
  if (e->cc_work) (void)hipFree(e->cc_work), e->cc_work = nullptr, e->cc_cap = 0;
  if (e->pack_fg) (void)hipFree(e->pack_fg), e->pack_fg = nullptr, e->pack_fg_bytes = 0;
  if (e->h_raw) (void)hipHostFree(e->h_raw), e->h_raw = nullptr;
  if (e->d_raw) (void)hipFree(e->d_raw), e->d_raw = nullptr;
  if (e->d_ingest_ws) (void)hipFree(e->d_ingest_ws), e->d_ingest_ws = nullptr;
  e->last_fg_stream = -1;
  if (e->vmm.base) {  // a model built by model_allocate from physical chunks: not hipFree's to release
    if ((void*)e->mog2_state == e->vmm.base) e->mog2_state = nullptr;
    if ((void*)e->mog1_state == e->vmm.base) e->mog1_state = nullptr;
    if ((void*)e->dp_state == e->vmm.base) e->dp_state = nullptr;
    vmm_free(e->vmm);
  }
  void* dev[] = {e->dp_state, e->gmg_rec, e->gmg_nfeat, e->bgstate, e->bgstate2, e->mog1_state, e->mog2_state, e->d_in, e->d_fg, e->d_bg};
  for (void* d : dev)
    if (d) (void)hipFree(d);
  e->dp_state = nullptr, e->gmg_rec = nullptr, e->gmg_nfeat = nullptr, e->bgstate = e->bgstate2 = nullptr, e->mog1_state = nullptr, e->mog2_state = nullptr, e->d_in = e->d_fg = e->d_bg = nullptr;
  void* host[] = {e->h_in, e->h_fg, e->h_bg};
  for (void* h : host)
    if (h) (void)hipHostFree(h);
  e->h_in = e->h_fg = e->h_bg = nullptr;
  for (auto& ev : e->events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  e->events.clear();
  ss_free(e);
  if (e->d_stat) (void)hipFree(e->d_stat), e->d_stat = nullptr;
  if (e->h_stat) (void)hipHostFree(e->h_stat), e->h_stat = nullptr;
  for (int i = 0; i < bgs_engine::kStatSlots; ++i) {
    if (e->stat_ev[i]) (void)hipEventDestroy(e->stat_ev[i]), e->stat_ev[i] = nullptr;
    e->stat_posted[i] = false;
  }
  for (auto& ev : e->band_ev)
    if (ev) (void)hipEventDestroy(ev), ev = nullptr;
  for (auto& hp : e->pin) {
    if (hp.pinned) (void)hipHostUnregister(const_cast<void*>(hp.ptr));
    hp = bgs_engine::HostPin();
  }
  for (auto& a : e->arenas) (void)hipHostUnregister(const_cast<uint8_t*>(a.first));
  e->arenas.clear();
  for (auto& ln : e->lanes) lane_release(ln);
}

int check_params(bgs_algo algo, const bgs_params& p) {
  if (algo == BGS_GMG && (p.gmg_max_features < 1 || p.gmg_max_features > 64)) return fail(BGS_ERR_UNSUPPORTED, "GMG maxFeatures must be 1..64, got %d", p.gmg_max_features);
  if (algo == BGS_GMG && p.gmg_smoothing_radius != 0 && (p.gmg_smoothing_radius < 3 || p.gmg_smoothing_radius > 15 || p.gmg_smoothing_radius % 2 == 0))
    return fail(BGS_ERR_UNSUPPORTED, "GMG smoothingRadius (cv::medianBlur kernel) must be 0 or odd 3..15, got %d", p.gmg_smoothing_radius);
  if (algo == BGS_MOG2 && p.mog2_nmixtures != bgs::kMog2K) return fail(BGS_ERR_UNSUPPORTED, "MOG2 kernel is built for K=%d mixtures, got %d", bgs::kMog2K, p.mog2_nmixtures);
  if (algo == BGS_MOG1 && p.mog1_nmixtures != bgs::kMog1K) return fail(BGS_ERR_UNSUPPORTED, "MOG1 kernel is built for K=%d mixtures, got %d", bgs::kMog1K, p.mog1_nmixtures);
  if ((algo == BGS_DP_ZIVKOVIC_AGMM || algo == BGS_DP_GRIMSON_GMM) && (p.dp_gaussians < 1 || p.dp_gaussians > 5))
    return fail(BGS_ERR_UNSUPPORTED, "dp GMM kernels are built for 1..5 gaussians, got %d", p.dp_gaussians);
  if (algo == BGS_DP_ADAPTIVE_MEDIAN && p.dp_sampling_rate == 0) return fail(BGS_ERR_UNSUPPORTED, "AdaptiveMedian samplingRate 0 (frame_num %% 0)");
  return BGS_OK;
}

// Every model / history / staging buffer of an engine comes from here.  Nothing may rely on what a fresh allocation holds:
// each model is initialised at a stream's first frame ON THE LAUNCH STREAM (mog2_clear, mog1_clear_kernel, gmg_clear_kernel,
// dp_gmm_clear_kernel, ss_init_streams, ...).  BGS_DEBUG_POISON=1 makes a violation deterministic instead of timing- and
// allocator-dependent: the buffer is filled with 0xA5 (a NaN-free but wildly wrong float, a mode count of 165) before first use.
int dmalloc(bgs_engine* e, void** p, size_t bytes) {
  HIP_TRY(hipMalloc(p, bytes));
  if (e->poison) HIP_TRY(hipMemsetAsync(*p, 0xA5, bytes, e->stream));  // allocate() drains e->stream before it returns
  return BGS_OK;
}
#define DMALLOC(ptr, bytes)                               \
  do {                                                    \
    int rc__ = dmalloc(e, (void**)&(ptr), (bytes));       \
    if (rc__) return rc__;                                \
  } while (0)

struct Timed {
  bgs_engine* e;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  Timed(bgs_engine* e_, hipStream_t s_, const char* name, bool enable = true) : e(e_), s(s_) {
    if (enable) e->kernel_name = name;
    // bounded: timing is a measurement aid, a caller that leaves it on must not grow the list forever
    if (enable && e->timing && e->events.size() < 16384 && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s);
  }
  ~Timed() {
    if (a && b) {
      (void)hipEventRecord(b, s);
      e->events.emplace_back(a, b);
    }
  }
};

// Automatic choice of how a per-frame launch loads a pixel's model (kernel_mog2.h; results are identical, only speed differs):
//   1 eager   everything at once, no dependent loads: right when most pixels have most modes and need them;
//   2 count   only the modes a pixel has (one dependent round): quiet scenes, one or two modes per pixel;
//   4 filter  summaries first, then only the records they cannot rule out (one dependent round, +4 B per mode for the
//             summaries): pays when at least half of a pixel's records are ruled out (modes far apart).
// About 256 sampled workgroups of every filter-kernel launch count, per pixel, the modes it has and the records that kernel loads
// or would load (when another kernel is current, every 16th launch - every 4th of a stream's first 64 - is a filter launch for
// that purpose).  The host never blocks: the counters come back through a pinned buffer and an event that is queried before
// every launch; it switches at once on clear evidence, else when two samples in a row ask for the same other mode.
void mog2_stat_read(bgs_engine* e) {
  // the newest post whose copy has completed; everything older is dropped with it
  int slot = -1;
  for (unsigned back = 1; back <= (unsigned)bgs_engine::kStatSlots && back <= e->stat_seq; ++back) {
    const int i = (int)((e->stat_seq - back) % bgs_engine::kStatSlots);
    if (!e->stat_posted[i]) break;  // already consumed (and so is everything older)
    if (slot < 0 && hipEventQuery(e->stat_ev[i]) == hipSuccess) slot = i;
    if (slot >= 0) e->stat_posted[i] = false;
  }
  if (slot < 0) return;
  const unsigned* hs = e->h_stat + 3 * slot;
  const unsigned total = hs[0], live = hs[1], need = hs[2];
  if (total < 64 * 5) return;
  const float lf = (float)live / (float)total, nf = (float)need / (float)total;
  const int want = (nf < 0.5f * lf && lf - nf > 0.1f) ? 4 : lf < 0.7f ? 2 : 1;
  const bool clear = (want == 4 && nf < 0.35f * lf) || (want != 4 && e->mog2_sparse_now == 4 && nf > 0.8f * lf);
  static const bool debug = getenv("BGS_DEBUG_STAT") != nullptr;
  if (debug)
    fprintf(stderr, "[bgs] mog2 auto: %u record slots sampled, %.3f live, %.3f needed after the summaries -> mode %d (now %d)\n", total, lf, nf, want, e->mog2_sparse_now);
  if (want != e->mog2_sparse_now && (clear || want == e->mog2_sparse_want)) e->mog2_sparse_now = want;
  e->mog2_sparse_want = want;
}
void mog2_stat_post(bgs_engine* e, hipStream_t s) {
  const int i = (int)(e->stat_seq % bgs_engine::kStatSlots);  // the oldest slot is reused (its event re-recorded) if nobody read it
  (void)hipMemcpyAsync(e->h_stat + 3 * i, e->d_stat, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, s);
  (void)hipMemsetAsync(e->d_stat, 0, 3 * sizeof(unsigned), s);
  (void)hipEventRecord(e->stat_ev[i], s);
  e->stat_posted[i] = true;
  e->stat_seq++;
}

int launch_mog2(bgs_engine* e, bgs::Mog2Args& a, hipStream_t s, bool timed = true) {
  const bgs_params& p = e->p;
  // shadow test only when it can change the delivered mask: not thresholded, or the threshold separates shadow from foreground
  a.shadow = p.mog2_detect_shadows && (!p.enable_threshold || ((p.mog2_shadow_value > p.threshold) != (255 > p.threshold)));
  a.want_bg = a.bgimg != nullptr, a.packed = a.fg_bits != nullptr;
  a.xcd_swizzle = e->xcd_swizzle, a.complete = e->mog2_complete;
  const bool autom = timed && e->mog2_sparse == 3;
  if (autom) mog2_stat_read(e);
  int mode = e->mog2_sparse == 3 ? e->mog2_sparse_now : e->mog2_sparse;
  if (mode >= 4 && (a.shadow || a.want_bg)) mode = 2;  // shadow test and background image read every mode's mean: nothing to rule out
  // auto mode: the filter kernel's sampled workgroups count what each way of loading would read; when another kernel is current,
  // every 16th launch (every 4th of the first 64) goes through the filter kernel anyway so that the choice keeps following the scene
  if (autom && mode != 4) {
    const unsigned n = e->mog2_launches++;
    if ((n & (n < 64 ? 3u : 15u)) == 0) mode = 4;
  }
  a.sparse = mode;
  a.stat = (autom && mode == 4) ? e->d_stat : nullptr;
  if (a.packed && a.npix % 64) return fail(BGS_ERR_UNSUPPORTED, "packed mask needs pixels %% 64 == 0");
  Timed t(e, s, "mog2_update_kernel", timed);
  const dim3 grid(blocks_for(a.npix)), block(bgs::kBlock);  // one pixel per lane (round 2's 1 / 2 / 4 comparison: equal or better everywhere)
  unsigned every = 1;  // sample about 256 workgroups per launch whatever the grid: enough to decide, few enough atomics not to show
  while (grid.x / every > 256) every <<= 1;
  a.stat_mask = every - 1;
  // BGS_MOG2_LDS_PAD=bytes: unused dynamic LDS per workgroup - fewer workgroups fit a CU (160 KB): the occupancy study of DESIGN.md 6.1
  // (round 4: 2 / 3 / 4 / 5 waves per SIMD 2.15 / 1.50 / 1.22 / 1.09 ms; a build of the filter kernel without the shadow and background
  // code - 66 instead of 91 VGPRs, 7 waves - ran no faster than this one's 5: profiles/r04_mog2_occupancy_ab.txt)
  static const unsigned lds_pad = getenv("BGS_MOG2_LDS_PAD") ? (unsigned)std::max(0, std::min(65536, atoi(getenv("BGS_MOG2_LDS_PAD")))) : 0u;
  if (mode >= 4)
    hipLaunchKernelGGL((bgs::mog2_update_kernel<bgs::kMog2Filter>), grid, block, lds_pad, s, a);
  else if (mode >= 2)
    hipLaunchKernelGGL((bgs::mog2_update_kernel<bgs::kMog2Count>), grid, block, lds_pad, s, a);
  else
    hipLaunchKernelGGL((bgs::mog2_update_kernel<bgs::kMog2Eager>), grid, block, lds_pad, s, a);
  if (a.stat) mog2_stat_post(e, s);
  return BGS_OK;
}

void mog1_fill_args(const bgs_engine* e, bgs::Mog1Args& m, double lr) {
  const bgs_params& p = e->p;
  const int C = e->ch;
  const double defaultNoiseSigma = 30 * 0.5;
  m.alpha = (float)lr, m.T = (float)p.mog1_background_ratio, m.vT = (float)p.mog1_var_threshold;
  m.w0 = (float)0.05;
  m.sk0 = C == 3 ? (float)(m.w0 / (defaultNoiseSigma * 2 * std::sqrt(3.))) : (float)(m.w0 / (defaultNoiseSigma * 2));
  m.var0 = (float)(defaultNoiseSigma * defaultNoiseSigma * 4);
  m.minVar = (float)(p.mog1_noise_sigma * p.mog1_noise_sigma);
  m.thr = p.threshold, m.enable_thr = p.enable_threshold, m.packed = m.fg_bits != nullptr, m.xcd_swizzle = e->xcd_swizzle;
}

void mog2_fill_args(const bgs_engine* e, bgs::Mog2Args& m, double lr) {
  const bgs_params& p = e->p;
  m.state = e->mog2_state;
  m.alphaT = (float)lr, m.alpha1 = 1.f - m.alphaT, m.prune = (float)(-lr * (double)p.mog2_ct);
  m.Tb = p.mog2_var_threshold, m.TB = p.mog2_background_ratio, m.Tg = p.mog2_var_threshold_gen;
  m.varInit = p.mog2_var_init, m.varMin = p.mog2_var_min, m.varMax = p.mog2_var_max, m.tau = p.mog2_tau;
  m.thr = p.threshold, m.enable_thr = p.enable_threshold, m.shadow_val = p.mog2_shadow_value;
}

void mog2_clear(bgs_engine* e, const bgs::Mog2Args& m, hipStream_t s) {
  hipLaunchKernelGGL(bgs::mog2_clear_kernel, dim3(blocks_for(m.npix)), dim3(bgs::kBlock), 0, s, m);
}

// One launch over `fuse` (2, 4 or 8) consecutive frames of streams whose model starts at c.m.state_off (kernel_mog2.h, clip launches)
int launch_mog2_clip(bgs_engine* e, bgs::Mog2ClipArgs& c, int fuse, hipStream_t s) {
  const bgs_params& p = e->p;
  bgs::Mog2Args& a = c.m;
  a.shadow = p.mog2_detect_shadows && (!p.enable_threshold || ((p.mog2_shadow_value > p.threshold) != (255 > p.threshold)));
  a.want_bg = a.bgimg != nullptr, a.packed = a.fg_bits != nullptr;
  a.xcd_swizzle = e->xcd_swizzle, a.complete = e->mog2_complete;
  a.sparse = e->mog2_sparse == 0 ? 0 : 1;  // clip launches load every record at once (kernel_mog2.h)
  a.stat = nullptr;
  if (a.packed && a.npix % 64) return fail(BGS_ERR_UNSUPPORTED, "packed mask needs pixels %% 64 == 0");
  Timed t(e, s, "mog2_clip_kernel");
  const dim3 grid(blocks_for(a.npix)), block(bgs::kBlock);
  unsigned every = 1;
  while (grid.x / every > 256) every <<= 1;
  a.stat_mask = every - 1;
#define MOG2_CLIP_CASE(TV) \
  if (fuse == TV) hipLaunchKernelGGL((bgs::mog2_clip_kernel<TV>), grid, block, 0, s, c);
  MOG2_CLIP_CASE(2) MOG2_CLIP_CASE(4) MOG2_CLIP_CASE(8)
#undef MOG2_CLIP_CASE
  return BGS_OK;
}

size_t mog2_state_bytes(const bgs_engine* e) {
  const size_t P = e->n * e->S;
  return (P + bgs::kMog2Tile - 1) / bgs::kMog2Tile * bgs::kMog2TileBytes;
}

// Model allocation for the big, long-lived models (MOG2, MOG1, dp): ONE virtual range backed by separately created physical chunks.
// Measured on MI355X in rounds 1-3 (profiles/r02_placement_probe.txt, profiles/r03_placement.txt): the same kernel on
// the same layout streamed a multi-GB model at one of 2-3 speeds, 8-10 % apart, depending on which physical VRAM one big hipMalloc
// handed out - a physically contiguous run above ~1 GiB, which a LATER large hipMalloc of a long-lived process tends to get, was the
// slow one.  Rounds 1-2 looked for a fast placement by trial (up to 20 candidates of the whole model allocated and timed); round 3
// found that chunks of at most 1 GiB (hipMemCreate / hipMemMap) never land there: 15 of 15 fresh processes at 1.103-1.110 ms per
// launch for chunk sizes 2 MiB .. 1 GiB, 3 of 3 slow with 4 GiB chunks, plain hipMalloc 1.11 1.11 1.12 1.19 1.21 1.21.
// Round 4 checked the claim from the other side (bench.py `calibration`, the round-3 verdict's question): a float4 copy through a chunked
// range is NOT faster than through a fresh plain allocation (6.14-6.21 against 6.19-6.27 TB/s on every box) - the construction does
// not buy bandwidth, it avoids the bad placement - and it does not explain the 6-9 % between BOXES of the pool, which show the same
// copy rate.  So: chunks of 256 MiB (BGS_MODEL_CHUNK_MB; 0 = one plain hipMalloc), no probe, no transient memory.
int vmm_allocate(VmmRange& v, int device, void** out, size_t bytes, size_t chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (!gran) gran = 2u << 20;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* base = nullptr;
  HIP_TRY(hipMemAddressReserve(&base, total, 0, nullptr, 0));
  v.base = base, v.bytes = total, v.chunk = chunk, v.mapped = 0;
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    HIP_TRY(hipMemCreate(&h, chunk, &prop, 0));
    v.handles.push_back(h);
    HIP_TRY(hipMemMap((char*)base + off, chunk, 0, h, 0));
    v.mapped++;  // vmm_free unmaps exactly the (address, size) pairs that were mapped, also after a failure half-way
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  HIP_TRY(hipMemSetAccess(base, total, &acc, 1));
  *out = base;
  return BGS_OK;
}
void vmm_free(VmmRange& v) {
  if (!v.base) return;
  // HIP's virtual memory API is Beta: release in the portable order - every chunk unmapped with the (address, size) it was mapped
  // with, then every handle released, then the range freed - and say so when a step fails (a silent failure here leaks gigabytes)
  auto warn = [](const char* what, hipError_t er) {
    if (er != hipSuccess) {
      fprintf(stderr, "[bgs] %s failed while releasing a chunked range: %s\n", what, hipGetErrorString(er));
      (void)hipGetLastError();
    }
  };
  for (size_t i = 0; i < v.mapped; ++i) warn("hipMemUnmap", hipMemUnmap((char*)v.base + i * v.chunk, v.chunk));
  for (auto& h : v.handles) warn("hipMemRelease", hipMemRelease(h));
  warn("hipMemAddressFree", hipMemAddressFree(v.base, v.bytes));
  v.handles.clear(), v.base = nullptr, v.bytes = 0, v.mapped = 0;
}

// A model of `bytes`: chunked (see above) from 768 MB up - smaller ones sit in the 256 MiB Infinity Cache for a good part and are
// not HBM-bound - else, or when the virtual memory API refuses, one hipMalloc.  An engine has at most one such model.
int model_allocate(bgs_engine* e, void** out, size_t bytes) {
  if (e->model_chunk_mb > 0 && bytes >= e->model_chunk_min_bytes && !e->vmm.base) {
    if (vmm_allocate(e->vmm, e->device, out, bytes, (size_t)e->model_chunk_mb << 20) == BGS_OK) {
      if (e->poison) HIP_TRY(hipMemsetAsync(*out, 0xA5, e->vmm.bytes, e->stream));
      return BGS_OK;
    }
    (void)hipGetLastError();
    vmm_free(e->vmm);  // whatever part of it came to be
    *out = nullptr;
  }
  return dmalloc(e, out, bytes);
}

int mog2_allocate(bgs_engine* e) {
  const size_t bytes = mog2_state_bytes(e);
  HIP_TRY(hipMalloc((void**)&e->d_stat, 3 * sizeof(unsigned)));
  HIP_TRY(hipMemsetAsync(e->d_stat, 0, 3 * sizeof(unsigned), e->stream));  // ordered: allocate() drains e->stream before it returns
  HIP_TRY(hipHostMalloc((void**)&e->h_stat, 3 * bgs_engine::kStatSlots * sizeof(unsigned), hipHostMallocDefault));
  for (int i = 0; i < bgs_engine::kStatSlots; ++i) HIP_TRY(hipEventCreateWithFlags(&e->stat_ev[i], hipEventDisableTiming));
  return model_allocate(e, (void**)&e->mog2_state, bytes);
}

#include "engine_subsense.h"
#include "engine_dp.h"

// (Re)build ABL's lookup table for the current alpha on e->stream.  Called when the geometry is set and when bgs_set_params
// changes alpha; both drain the device first / the stream after, so no launch on any stream sees a half-written table.
int abl_build_lut(bgs_engine* e) {
  if (!e->abl_lut) HIP_TRY(hipMalloc((void**)&e->abl_lut, 256 * 256));
  hipLaunchKernelGGL(bgs::abl_lut_kernel, dim3(256), dim3(bgs::kBlock), 0, e->stream, e->abl_lut, e->p.alpha, 1 - e->p.alpha);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->abl_lut_alpha = e->p.alpha, e->abl_lut_valid = true;
  return BGS_OK;
}

// ASBL's two tables (learning / detection phase), same rules
int asbl_build_lut(bgs_engine* e) {
  const size_t one = (size_t)bgs::kAsblLutRows * 256;
  if (!e->abl_lut) HIP_TRY(hipMalloc((void**)&e->abl_lut, 2 * one));
  const bgs_params& p = e->p;
  for (int learn = 1; learn >= 0; --learn)
    hipLaunchKernelGGL(bgs::asbl_lut_kernel, dim3(bgs::kAsblLutRows), dim3(bgs::kBlock), 0, e->stream, e->abl_lut + (learn ? 0 : one), learn, p.alpha_learn, 1 - p.alpha_learn,
                       p.alpha_detection, 1 - p.alpha_detection);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->asbl_lut_alpha[0] = p.alpha_learn, e->asbl_lut_alpha[1] = p.alpha_detection, e->abl_lut_valid = true;
  return BGS_OK;
}

int allocate(bgs_engine* e, int rows, int cols, int ch) {
  if (rows <= 0 || cols <= 0) return fail(BGS_ERR_INVALID, "bad geometry %dx%d", rows, cols);
  if (ch != 1 && ch != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3, got %d", ch);
  if (e->algo == BGS_MOG2 && ch != 3)
    return fail(BGS_ERR_UNSUPPORTED, "MixtureOfGaussianV2BGS needs 3 channels: getBackgroundImage asserts nchannels == 3 (MixtureOfGaussianV2BGS.cpp:59)");
  HIP_TRY(hipSetDevice(e->device));
  e->rows = rows, e->cols = cols, e->ch = ch, e->n = (size_t)rows * cols;
  const size_t P = e->n * e->S, fb = P * ch;
  if (!e->stream) HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  switch (e->algo) {
    case BGS_FRAME_DIFF: e->nring = 2; break;
    case BGS_WMM:
    case BGS_WMV: e->nring = 3; break;
    case BGS_STATIC_FRAME_DIFF:
    case BGS_ABL: e->state_ch = ch; break;
    case BGS_ASBL: e->state_ch = 1; break;
    case BGS_SIGMA_DELTA:
      if (ch != 3) return fail(BGS_ERR_UNSUPPORTED, "SigmaDeltaBGS is 3-channel only (sdLaMa091AllocInit_8u_C3R, SigmaDeltaBGS.cpp:35)");
      e->state_ch = 3;
      break;
    case BGS_GMG: e->state_ch = 1; break;  // bgstate = the unsmoothed mask
    case BGS_MOG1:
    case BGS_MOG2:
    case BGS_SUBSENSE:
    case BGS_LOBSTER: break;
    case BGS_DP_ZIVKOVIC_AGMM:
    case BGS_DP_GRIMSON_GMM: e->state_ch = 1; break;  // bgstate = modes per pixel
    case BGS_DP_WREN_GA:
    case BGS_DP_MEAN: break;
    case BGS_DP_ADAPTIVE_MEDIAN: e->state_ch = 3; break;  // bgstate = the median image
    default: return fail(BGS_ERR_UNSUPPORTED, "algorithm %d is not implemented in this build", (int)e->algo);
  }
  for (int i = 0; i < e->nring; ++i) DMALLOC(e->ring[i], fb);
  if (e->state_ch) DMALLOC(e->bgstate, P * e->state_ch);
  if (e->algo == BGS_ABL) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cu = prop.multiProcessorCount;
    int rc = abl_build_lut(e);
    if (rc) return rc;
  }
  if (e->algo == BGS_ASBL) {
    DMALLOC(e->bgstate2, P);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cu = prop.multiProcessorCount;
    int rc = asbl_build_lut(e);
    if (rc) return rc;
  }
  if (e->algo == BGS_SIGMA_DELTA) DMALLOC(e->bgstate2, P * 3);  // Vt
  if (e->algo == BGS_GMG) {
    const size_t F = (size_t)e->p.gmg_max_features;
    DMALLOC(e->gmg_rec, P * F * sizeof(int2));
    DMALLOC(e->gmg_nfeat, P);
  }
  if (e->algo == BGS_MOG1) {
    const size_t tile_floats = ch == 3 ? bgs::mog1_tile_floats<3>() : bgs::mog1_tile_floats<1>();
    const size_t tiles = (P + bgs::kMog1Tile - 1) / bgs::kMog1Tile;
    const size_t bytes = tiles * tile_floats * sizeof(float);
    int rc = model_allocate(e, (void**)&e->mog1_state, bytes);
    if (rc) return rc;
  }
  if (e->algo == BGS_MOG2) {
    int rc = mog2_allocate(e);
    if (rc) return rc;
  }
  if (e->algo == BGS_SUBSENSE) {
    int rc = ss_allocate(e);
    if (rc) return rc;
  }
  if (is_dp(e->algo)) {
    int rc = dp_allocate(e);
    if (rc) return rc;
  }
  if (e->algo == BGS_LOBSTER) {
    int rc = lob_allocate(e);
    if (rc) return rc;
  }
  // Whatever allocation enqueued on e->stream (statistics counters, poison fills) is complete before the caller's first
  // launch - which may come on ANOTHER stream (device path) that nothing else orders against this one.
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BGS_OK;
}

// OpenCV's capture loop hands IBGS::process the SAME frame buffer every frame (VideoCapture.cpp:158-218: cvQueryFrame's image,
// wrapped by cv::Mat img_input(frame)), and a caller that keeps its mask / background images allocated does the same on the way
// out.  A buffer that comes back with the same address and size as in the previous call is page-locked (hipHostRegister, once)
// and from then on the DMA engine reads / writes it directly: no staging copy by the CPU.  A different buffer in that role drops
// the registration; bgs_destroy drops them all.  Only contiguous images qualify (row step = row bytes).
// OPT-IN per role (BGS_OPT_HOST_REGISTER): the engine cannot see a buffer being freed and another one mapped at the same address,
// so the caller promises that a buffer it passes in an enabled role stays allocated until it passes a different one or destroys
// the engine.
bool host_pin(bgs_engine* e, int stream, int role, const void* ptr, size_t bytes, hipStream_t user = nullptr) {
  if (!ptr || !bytes) return false;
  for (const auto& a : e->arenas)  // inside an arena the caller registered as a whole: nothing to do per image
    if ((const uint8_t*)ptr >= a.first && (const uint8_t*)ptr + bytes <= a.first + a.second) return true;
  if (!((e->host_register >> role) & 1)) return false;
  bgs_engine::HostPin& hp = e->pin[(size_t)stream * 3 + role];
  if (hp.ptr == ptr && hp.bytes == bytes) {
    if (hp.pinned) return true;
    if (hp.refused) return false;
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t er = hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault);
    e->diag_reg_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    e->diag_reg_calls++;
    if (er == hipSuccess) return hp.pinned = true;
    (void)hipGetLastError();  // not registrable (e.g. already registered by the caller, read-only mapping): keep staging
    hp.refused = true;
    return false;
  }
  if (hp.pinned) {
    (void)hipStreamSynchronize(user ? user : e->stream);
    (void)hipHostUnregister(const_cast<void*>(hp.ptr));
    e->diag_unreg_calls++;
  }
  hp = bgs_engine::HostPin();
  hp.ptr = ptr, hp.bytes = bytes;  // a candidate: registered when it comes back
  return false;
}

int ensure_staging(bgs_engine* e) {
  if (e->h_in) return BGS_OK;
  const size_t fb = e->n * e->ch;
  HIP_TRY(hipHostMalloc((void**)&e->h_in, fb, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&e->h_fg, e->n, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&e->h_bg, fb, hipHostMallocDefault));
  DMALLOC(e->d_in, fb);
  DMALLOC(e->d_fg, e->n);
  DMALLOC(e->d_bg, fb);
  return BGS_OK;
}

#define LAUNCH_FRAME_KERNEL(KERNEL, name)                                                             \
  do {                                                                                                \
    Timed t__(e, s, name);                                                                            \
    if (C == 3) {                                                                                     \
      if (G == 16)                                                                                    \
        hipLaunchKernelGGL((bgs::KERNEL<16, 3>), dim3(blocks_for(a.npix / 16)), dim3(bgs::kBlock), 0, s, a); \
      else if (G == 4)                                                                                \
        hipLaunchKernelGGL((bgs::KERNEL<4, 3>), dim3(blocks_for(a.npix / 4)), dim3(bgs::kBlock), 0, s, a);   \
      else                                                                                            \
        hipLaunchKernelGGL((bgs::KERNEL<1, 3>), dim3(blocks_for(a.npix)), dim3(bgs::kBlock), 0, s, a);       \
    } else {                                                                                          \
      if (G == 16)                                                                                    \
        hipLaunchKernelGGL((bgs::KERNEL<16, 1>), dim3(blocks_for(a.npix / 16)), dim3(bgs::kBlock), 0, s, a); \
      else if (G == 4)                                                                                \
        hipLaunchKernelGGL((bgs::KERNEL<4, 1>), dim3(blocks_for(a.npix / 4)), dim3(bgs::kBlock), 0, s, a);   \
      else                                                                                            \
        hipLaunchKernelGGL((bgs::KERNEL<1, 1>), dim3(blocks_for(a.npix)), dim3(bgs::kBlock), 0, s, a);       \
    }                                                                                                 \
  } while (0)

// widest pixel group every pointer and the pixel count allow
// (`cap`: measured optimum of the kernel: wmm / wmv run ~10 % faster with 4 pixels per lane than with 16, abl the other way round)
int pick_group(const bgs::FrameArgs& a, int C, int cap = 16) {
  const void* ptrs[] = {a.cur, a.p1, a.p2, a.state_out, a.fg, a.bg};
  int G = 16;
  if (a.npix % 16) G = (a.npix % 4) ? 1 : 4;
  if (G > cap) G = cap;
  for (const void* p : ptrs) {
    if (!p) continue;
    if (G == 16 && !aligned(p, 16)) G = 4;
    if (G == 4 && !aligned(p, 4)) G = 1;
  }
  if (const char* env = getenv("BGS_FRAME_GROUP")) {
    const int want = atoi(env);
    if ((want == 1 || want == 4 || want == 16) && want <= G) G = want;
  }
  (void)C;
  return G;
}

// What one launch can cover: streams whose next frame needs the same kernel arguments share a RUN.  Cameras come and go
// independently (the reference creates and deletes one IBGS object per stream whenever it likes: FrameProcessor.cpp:35-155,
// :342-482; ustc_src/ustc_bgs.cpp:75-77), so the streams of a batch may have seen different numbers of frames; what a launch
// depends on is far less than the age - e.g. for MOG2 only "first frame?" and the learning rate, which with the wrapper's fixed
// alpha is the same from a stream's second frame on.  Streams in lock-step (the benchmark, any batch fed by whole-batch calls) are
// one run = one launch, exactly as before.
uint64_t launch_key(const bgs_engine* e, int i) {
  const bgs_params& p = e->p;
  const int64_t t = e->seen[i];
  auto lr_key = [&](double alpha, int64_t cap, int64_t mult) -> uint64_t {  // MOG1 / MOG2: needToInitialize + the learning rate of frame t
    if (t == 0 || alpha >= 1) return 1;
    if (alpha >= 0) return 2;
    return 3 + (uint64_t)std::min<int64_t>(mult * (t + 1), cap);
  };
  switch (e->algo) {
    case BGS_FRAME_DIFF:
    case BGS_WMM:
    case BGS_WMV: return (uint64_t)(e->rpos[i] % e->nring) | (uint64_t)std::min<int64_t>(t, e->nring - 1) << 8;  // ring slot + warm-up level
    case BGS_STATIC_FRAME_DIFF:
    case BGS_SIGMA_DELTA:
    case BGS_DP_ZIVKOVIC_AGMM:
    case BGS_DP_GRIMSON_GMM:
    case BGS_DP_WREN_GA:
    case BGS_DP_MEAN: return t == 0;
    case BGS_ABL: return (uint64_t)(t == 0) | (uint64_t)(((p.limit > 0 && p.limit < e->counter[i]) || p.limit == -1) ? 2 : 0) | (uint64_t)(p.limit > 0 ? std::min<int64_t>(e->counter[i], (int64_t)p.limit + 1) : 0) << 2;
    case BGS_GMG: return (uint64_t)(t == 0) | (uint64_t)(t >= p.gmg_init_frames) << 1 | (uint64_t)(t == (int64_t)p.gmg_init_frames - 1) << 2;
    case BGS_ASBL: return (uint64_t)(t == 0) | (uint64_t)e->flip[i] << 1 | (uint64_t)((p.learning_frames > 0 && e->counter[i] <= p.learning_frames) ? 4 : 0);
    case BGS_DP_ADAPTIVE_MEDIAN: return (uint64_t)(t == 0) | (uint64_t)((t % p.dp_sampling_rate) == 1) << 1;
    case BGS_MOG1: return lr_key(p.alpha, p.mog1_history, 1);
    case BGS_MOG2: return lr_key(p.alpha, p.mog2_history, 2);
    default: return (uint64_t)t | (uint64_t)(e->ss ? e->ss->pp[i] & 1 : 0) << 62;  // SuBSENSE / LOBSTER: the frame index itself goes into the kernels (counter-based random draws); + which half of the ping-pong maps is current
  }
}

int process_run(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits, hipStream_t s, uint32_t* out_flags);

// engine-owned byte masks for packed-only callers of frames whose pixel count is not a multiple of 64 (see process_range); grown on
// demand, ordered on the call's stream like every other engine buffer
int pack_scratch(bgs_engine* e, size_t bytes, hipStream_t s) {
  if (e->pack_fg_bytes >= bytes) return BGS_OK;
  if (e->pack_fg) {
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(e->pack_fg), e->pack_fg = nullptr, e->pack_fg_bytes = 0;
  }
  HIP_TRY(hipMalloc((void**)&e->pack_fg, bytes));
  e->pack_fg_bytes = bytes;
  return BGS_OK;
}
void pack_ragged(bgs_engine* e, const uint8_t* fg, uint64_t* bits, size_t images, hipStream_t s) {
  const size_t W = (e->n + 63) / 64;
  hipLaunchKernelGGL(bgs::mask_pack_ragged_kernel, dim3(blocks_for(images * W * bgs::kWave)), dim3(bgs::kBlock), 0, s, fg, bits, e->n, W, images);
}

// One frame for streams [first, first+count), device pointers, asynchronous on s: one launch per run of streams (see launch_key).
int process_range(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits, hipStream_t s,
                  uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  if (!e->n) return fail(BGS_ERR_INVALID, "geometry not set: call bgs_set_geometry or bgs_process first");
  if (first < 0 || count <= 0 || first + count > e->S) return fail(BGS_ERR_INVALID, "stream range [%d,%d) outside 0..%d", first, first + count, e->S);
  if (!d_frames) return fail(BGS_ERR_INVALID, "d_frames is NULL");
  // Packed masks: stream k of the call owns words [k W, (k + 1) W), W = ceil(rows*cols / 64).  When rows*cols is a multiple of 64
  // (1080p, 4K, 720p, VGA, the reference's 320x176 video ...) the kernels write the words themselves from wave ballots; otherwise a
  // wave's 64 pixels straddle two words (and two streams), so the kernels write byte masks - the caller's d_fg, or the engine's own
  // buffer when it passed none - and mask_pack_ragged_kernel makes the words from them, tail bits zero.
  const size_t W = (e->n + 63) / 64;
  const bool ragged = d_bits && (e->n % 64) != 0;
  const bool via_bytes = e->algo == BGS_GMG || e->algo == BGS_ASBL;  // their packed mask is always made from the finished byte mask (median after the pixel loop)
  if (d_bits && !d_fg && (ragged || via_bytes)) {
    int rc = pack_scratch(e, (size_t)count * e->n, s);
    if (rc) return rc;
    d_fg = e->pack_fg;
  }
  uint32_t all = ~0u;
  const size_t C = (size_t)e->ch, bgC = e->algo == BGS_ASBL ? 1 : C;
  for (int a = first; a < first + count;) {
    int b = a + 1;
    const uint64_t key = launch_key(e, a);
    while (b < first + count && launch_key(e, b) == key) ++b;
    if (e->borrow && (a != first || b != first + count) && e->nring && !e->borrow_in_clip)
      return fail(BGS_ERR_INVALID, "borrowed frame history needs streams in lock-step (streams %d and %d are not)", a, b);
    const size_t o = (size_t)(a - first) * e->n;
    uint32_t fl = 0;
    int rc = process_run(e, a, b - a, d_frames + o * C, d_fg ? d_fg + o : nullptr, d_bg ? d_bg + o * bgC : nullptr, (d_bits && !ragged) ? d_bits + (size_t)(a - first) * W : nullptr, s, &fl);
    if (rc) return rc;
    if (ragged && (fl & BGS_FG_VALID)) pack_ragged(e, d_fg + o, d_bits + (size_t)(a - first) * W, (size_t)(b - a), s);
    all &= fl;
    a = b;
  }
  if (out_flags) *out_flags = all;  // what holds for every stream of the range; per stream: bgs_stream_flags
  return BGS_OK;
}

// One frame for a run of streams that share every kernel argument.
int process_run(bgs_engine* e, int first, int count, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits, hipStream_t s,
                uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  const int64_t t = e->seen[first];
  HIP_TRY(hipSetDevice(e->device));
  const bgs_params& p = e->p;
  const int C = e->ch;
  const size_t npix = e->n * count, off = e->n * first, fb = npix * C;
  if (d_bits && npix % 64) return fail(BGS_ERR_INVALID, "internal: a ragged packed mask reached process_run");  // process_range packs those itself
  uint32_t flags = 0;

  bgs::FrameArgs a{};
  a.cur = d_frames, a.fg = d_fg, a.bg = d_bg, a.fg_bits = d_bits, a.npix = npix;
  a.thr = p.threshold, a.enable_thr = p.enable_threshold, a.enable_weight = p.enable_weight;
  // The XCD-aware block order pays where a workgroup's working set is a multi-plane tile (MOG2, MOG1, dp); the byte-stream
  // kernels run 2-5 % faster in plain block order (tools/ab_pointwise.py), so they only use it at level 2 (for A/B runs).
  a.xcd_swizzle = e->xcd_swizzle >= 2;

  const bool whole = (first == 0 && count == e->S);
  if (e->borrow && !whole && e->nring && !e->borrow_in_clip) return fail(BGS_ERR_INVALID, "borrowed frame history needs whole-batch calls");

  switch (e->algo) {
    case BGS_FRAME_DIFF:
    case BGS_WMM:
    case BGS_WMV: {
      const int R = e->nring, warm = R - 1;
      const int64_t rp = e->rpos[first];  // the same for every stream of the run (launch_key)
      const uint8_t *cur = d_frames, *h1 = nullptr, *h2 = nullptr;
      if (e->borrow) {
        h1 = e->borrowed[0], h2 = e->borrowed[1];
      } else {
        uint8_t* slot = e->ring[rp % R] + off * C;
        if (cur != slot) HIP_TRY(hipMemcpyAsync(slot, cur, fb, hipMemcpyDeviceToDevice, s));  // keep a private copy as history
        cur = slot;
        if (t >= 1) h1 = e->ring[(rp + R - 1) % R] + off * C;
        if (t >= 2 && R == 3) h2 = e->ring[(rp + R - 2) % R] + off * C;
      }
      if (t >= warm) {
        a.cur = cur, a.p1 = h1, a.p2 = h2;
        const int G = pick_group(a, C, e->algo == BGS_FRAME_DIFF ? 16 : 4);
        if (e->algo == BGS_FRAME_DIFF)
          LAUNCH_FRAME_KERNEL(framediff_kernel, "framediff_kernel");
        else if (e->algo == BGS_WMM)
          LAUNCH_FRAME_KERNEL(wmm_kernel, "wmm_kernel");
        else
          LAUNCH_FRAME_KERNEL(wmv_kernel, "wmv_kernel");
        flags = BGS_FG_VALID | (e->algo == BGS_WMM ? BGS_BG_VALID : 0u);
      }
      if (e->borrow) e->borrowed[1] = e->borrowed[0], e->borrowed[0] = d_frames;
      break;
    }
    case BGS_STATIC_FRAME_DIFF:
    case BGS_ABL: {
      uint8_t* st = e->bgstate + off * C;
      if (t == 0) HIP_TRY(hipMemcpyAsync(st, d_frames, fb, hipMemcpyDeviceToDevice, s));  // img_input.copyTo(img_background)
      a.p1 = st;
      if (e->algo == BGS_STATIC_FRAME_DIFF) {
        a.bg = nullptr;
        const int G = pick_group(a, C);
        LAUNCH_FRAME_KERNEL(framediff_kernel, "framediff_kernel");
        if (d_bg) HIP_TRY(hipMemcpyAsync(d_bg, st, fb, hipMemcpyDeviceToDevice, s));
      } else {
        a.state_out = st;
        a.alpha = p.alpha, a.beta = 1 - p.alpha;
        const int64_t cnt = e->counter[first];
        a.update = ((p.limit > 0 && p.limit < cnt) || p.limit == -1) ? 1 : 0;
        const int G = pick_group(a, C, 4);  // 4 pixels per lane: 42 VGPRs -> two 1024-lane workgroups per CU (16: 128 VGPRs, one); measured 0.126 vs 0.134 ms
        {
          Timed t__(e, s, "abl_kernel");
          const size_t per_tile = (size_t)bgs::kAblBlock * G, ntiles = (npix + per_tile - 1) / per_tile;
          // persistent: exactly as many workgroups as are resident at once (1 or 2 per CU, by registers), each walking its share of the tiles
          const dim3 block(bgs::kAblBlock);
#define ABL_CASE(GV, CV, UV)                                                                                                               \
  if (G == GV && C == CV && (a.update != 0) == UV) {                                                                                       \
    /* resident workgroups per CU of this instantiation: a property of the code object (the library is gfx950-only), cached per   \
       process; atomic because engines may be driven from several host threads */                                                  \
    static std::atomic<int> per_cu_cache{0};                                                                                               \
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);                                                                             \
    if (!per_cu) {                                                                                                                         \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bgs::abl_kernel<GV, CV, UV>, bgs::kAblBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1; \
      per_cu_cache.store(per_cu, std::memory_order_relaxed);                                                                               \
    }                                                                                                                                      \
    const dim3 grid((unsigned)std::min<size_t>(ntiles, (size_t)per_cu * e->n_cu));                                                        \
    hipLaunchKernelGGL((bgs::abl_kernel<GV, CV, UV>), grid, block, 0, s, a, (const uint8_t*)e->abl_lut);                                   \
  }
          ABL_CASE(16, 3, true) ABL_CASE(4, 3, true) ABL_CASE(1, 3, true) ABL_CASE(16, 1, true) ABL_CASE(4, 1, true) ABL_CASE(1, 1, true)
          ABL_CASE(16, 3, false) ABL_CASE(4, 3, false) ABL_CASE(1, 3, false) ABL_CASE(16, 1, false) ABL_CASE(4, 1, false) ABL_CASE(1, 1, false)
#undef ABL_CASE
        }
        if (a.update && p.limit > 0 && p.limit < cnt)
          for (int i = first; i < first + count; ++i) e->counter[i]++;
      }
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_GMG: {
      if (d_bits && !d_fg) return fail(BGS_ERR_UNSUPPORTED, "GMG: the packed mask is made from the byte mask, pass d_fg too");
      const size_t P = e->n * e->S;
      if (t == 0) hipLaunchKernelGGL(bgs::gmg_clear_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, e->gmg_nfeat + off, npix);  // initialize(): nfeatures = 0
      bgs::GmgArgs g{};
      g.frame = d_frames, g.raw = e->bgstate + off, g.rec = e->gmg_rec, g.nfeat = e->gmg_nfeat;
      g.plane = P, g.state_off = off, g.npix = npix, g.F = p.gmg_max_features, g.C = C, g.levels = p.gmg_quantization_levels;
      g.typical = t >= p.gmg_init_frames, g.update = p.gmg_update_background_model != 0, g.normalize_now = t == (int64_t)p.gmg_init_frames - 1;
      // a decayed weight goes back as the whole 8-byte record (512 contiguous bytes per wave) rather than as a 4-byte store into it
      // (every other dword of the line: partial sectors): 0.294 -> 0.281 ms per 8 x 1080p, same box, alternating; BGS_GMG_FULL_STORE=0: the 4-byte stores
      static const bool gmg_full = !(getenv("BGS_GMG_FULL_STORE") && atoi(getenv("BGS_GMG_FULL_STORE")) == 0);
      g.fullStore = gmg_full ? 1 : 0;
      g.lr = p.gmg_learning_rate, g.prior = p.gmg_background_prior, g.thr = p.gmg_decision_threshold;
      {
        Timed tm(e, s, "gmg_kernel");
        hipLaunchKernelGGL(bgs::gmg_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, g);
      }
      if (d_fg) {
        if (p.gmg_smoothing_radius > 0) {  // cv::medianBlur(fgmask, smoothingRadius) of a {0,255} mask
          bgs::MorphArgs m{e->bgstate + off, d_fg, e->rows, e->cols, 3, p.gmg_smoothing_radius};
          bgs::morph_launch(m, (int)count, s);
        } else {
          HIP_TRY(hipMemcpyAsync(d_fg, e->bgstate + off, npix, hipMemcpyDeviceToDevice, s));
        }
        if (d_bits) hipLaunchKernelGGL(bgs::mask_pack_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, (const uint8_t*)d_fg, d_bits, npix);
      }
      flags = BGS_FG_VALID;  // no getBackgroundImage for GMG (GMG.cpp:59): img_bgmodel ends up empty
      break;
    }
    case BGS_SUBSENSE: {
      int rc = ss_process(e, first, count, d_frames, d_fg, d_bg, s, t);
      if (rc) return rc;
      if (d_bits)  // the mask is also model state (m_oLastFGMask): pack it from there
        hipLaunchKernelGGL(bgs::mask_pack_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, (const uint8_t*)(e->ss->u8[SS_LASTFG] + off), d_bits, npix);
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_LOBSTER: {
      int rc = lob_process(e, first, count, d_frames, d_fg, d_bg, s, t);
      if (rc) return rc;
      if (d_bits)
        hipLaunchKernelGGL(bgs::mask_pack_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, (const uint8_t*)(e->ss->u8[SS_LASTFG] + off), d_bits, npix);
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_SIGMA_DELTA: {
      uint8_t *mt = e->bgstate + off * 3, *vt = e->bgstate2 + off * 3;
      if (t == 0) {  // SigmaDeltaBGS.cpp:33-39: allocate + initialise, return without output
        HIP_TRY(hipMemcpyAsync(mt, d_frames, fb, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(bgs::sigmadelta_init_vt_kernel, dim3(blocks_for(fb)), dim3(bgs::kBlock), 0, s, vt, fb, e->cols, (int)(uint8_t)p.sd_min_var);
        break;
      }
      bgs::SigmaDeltaArgs q{};
      q.cur = d_frames, q.mt = mt, q.vt = vt, q.fg = d_fg, q.fg_bits = d_bits, q.npix = npix;
      q.N = (uint32_t)p.sd_amp_factor, q.vmin = (uint8_t)p.sd_min_var, q.vmax = (uint8_t)p.sd_max_var, q.xcd_swizzle = e->xcd_swizzle >= 2;
      int G = 16;
      if (npix % 16 || !aligned(d_frames, 16) || !aligned(mt, 16) || !aligned(vt, 16) || (d_fg && !aligned(d_fg, 16))) G = (npix % 4 || !aligned(d_frames, 4) || !aligned(mt, 4) || (d_fg && !aligned(d_fg, 4))) ? 1 : 4;
      {
        Timed tm(e, s, "sigmadelta_kernel");
        if (G == 16) hipLaunchKernelGGL((bgs::sigmadelta_kernel<16>), dim3(blocks_for(npix / 16)), dim3(bgs::kBlock), 0, s, q);
        if (G == 4) hipLaunchKernelGGL((bgs::sigmadelta_kernel<4>), dim3(blocks_for(npix / 4)), dim3(bgs::kBlock), 0, s, q);
        if (G == 1) hipLaunchKernelGGL((bgs::sigmadelta_kernel<1>), dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, q);
      }
      flags = BGS_FG_VALID;
      break;
    }
    case BGS_DP_ZIVKOVIC_AGMM:
    case BGS_DP_GRIMSON_GMM:
    case BGS_DP_WREN_GA:
    case BGS_DP_MEAN:
    case BGS_DP_ADAPTIVE_MEDIAN: {
      int rc = dp_process(e, first, count, t, d_frames, d_fg, d_bits, s, &flags);
      if (rc) return rc;
      break;
    }
    case BGS_ASBL: {
      const int cur = e->flip[first];
      for (int i = first; i < first + count; ++i)
        if (e->flip[i] != cur) return fail(BGS_ERR_INVALID, "streams %d and %d are not in lock-step", first, i);
      uint8_t* bufs[2] = {e->bgstate, e->bgstate2};
      if (t == 0) {  // img_input(gray).copyTo(img_background): a frame of threshold -1 ... simplest exact way is a tiny gray kernel
        bgs::FrameArgs g{};
        g.cur = d_frames, g.fg = bufs[cur] + off, g.npix = npix, g.enable_thr = 0;
        const int G = 1;
        bgs::FrameArgs a = g;  // LAUNCH_FRAME_KERNEL reads `a`
        LAUNCH_FRAME_KERNEL(gray_kernel, "gray_kernel");
      }
      bgs::AsblArgs q{};
      q.frame = d_frames, q.bg_in = bufs[cur] + off, q.bg_out = bufs[cur ^ 1] + off, q.fg = d_fg, q.bg_img = d_bg;
      q.rows = e->rows, q.cols = e->cols, q.thr = p.threshold;
      const int64_t cnt = e->counter[first];
      q.learn = (p.learning_frames > 0 && cnt <= p.learning_frames) ? 1 : 0;
      q.aL = p.alpha_learn, q.bL = 1 - p.alpha_learn, q.aD = p.alpha_detection, q.bD = 1 - p.alpha_detection;
      {
        Timed tm(e, s, "asbl_kernel");
        static const bool table = !(getenv("BGS_ASBL_TABLE") && atoi(getenv("BGS_ASBL_TABLE")) == 0);
        const uintptr_t ptrs = (uintptr_t)q.frame | (uintptr_t)q.bg_in | (uintptr_t)q.bg_out | (uintptr_t)q.fg | (uintptr_t)q.bg_img;
        // the table kernel moves whole dwords: rows of 4n pixels, aligned images (also the stream offset inside them: n % 4 == 0 then)
        if (table && e->cols % 4 == 0 && e->cols >= 4 && ptrs % 4 == 0) {
          // one wave per strip of 256 columns x R rows; 2 resident workgroups of 16 waves per CU (67 KB of LDS each): R such that the
          // strips of this launch fill them once, at least 8 rows (each strip re-reads the rows above and below it)
          const size_t waves = (size_t)2 * e->n_cu * (bgs::kAsbl2Block / bgs::kWave), nsx = (e->cols + bgs::kAsblSW - 1) / bgs::kAsblSW;
          const size_t blocks_y = std::max<size_t>(1, waves / ((size_t)count * nsx));
          const int R = (int)std::max<size_t>(8, (e->rows + blocks_y - 1) / blocks_y);
          const size_t nstrips = (size_t)count * nsx * ((e->rows + R - 1) / R), per_wg = bgs::kAsbl2Block / bgs::kWave;
          if (nstrips >= (1u << 31)) return fail(BGS_ERR_UNSUPPORTED, "AdaptiveSelectiveBackgroundLearning: launch too large");
          const dim3 grid((unsigned)std::min<size_t>((nstrips + per_wg - 1) / per_wg, (size_t)2 * e->n_cu));
          const uint8_t* lut = e->abl_lut + (q.learn ? 0 : (size_t)bgs::kAsblLutRows * 256);
          if (C == 3)
            hipLaunchKernelGGL((bgs::asbl_stream_kernel<3>), grid, dim3(bgs::kAsbl2Block), 0, s, q, lut, count, R);
          else
            hipLaunchKernelGGL((bgs::asbl_stream_kernel<1>), grid, dim3(bgs::kAsbl2Block), 0, s, q, lut, count, R);
        } else {
          const dim3 grid((e->cols + bgs::kAsblTW - 1) / bgs::kAsblTW, (e->rows + bgs::kAsblTH - 1) / bgs::kAsblTH, count);
          if (C == 3)
            hipLaunchKernelGGL((bgs::asbl_kernel<3>), grid, dim3(bgs::kBlock), 0, s, q);
          else
            hipLaunchKernelGGL((bgs::asbl_kernel<1>), grid, dim3(bgs::kBlock), 0, s, q);
        }
      }
      if (d_bits) {
        if (!d_fg) return fail(BGS_ERR_UNSUPPORTED, "AdaptiveSelectiveBackgroundLearning: the packed mask is made from the byte mask, pass d_fg too");
        hipLaunchKernelGGL(bgs::mask_pack_kernel, dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, (const uint8_t*)d_fg, d_bits, npix);
      }
      for (int i = first; i < first + count; ++i) {
        e->flip[i] = (uint8_t)(cur ^ 1);
        if (q.learn) e->counter[i]++;
      }
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    case BGS_MOG1: {
      double lr = p.alpha;
      int64_t nframes = t;
      bgs::Mog1Args m{};
      m.state = e->mog1_state, m.state_off = off, m.npix = npix;
      if (nframes == 0 || lr >= 1) {  // needToInitialize: bgmodel = zeros
        if (C == 3)
          hipLaunchKernelGGL((bgs::mog1_clear_kernel<3>), dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, m);
        else
          hipLaunchKernelGGL((bgs::mog1_clear_kernel<1>), dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, m);
        nframes = 0;
      }
      ++nframes;
      lr = (lr >= 0 && nframes > 1) ? lr : 1. / (double)std::min<int64_t>(nframes, p.mog1_history);
      m.frame = d_frames, m.fg = d_fg, m.fg_bits = d_bits;
      mog1_fill_args(e, m, lr);
      {
        Timed tm(e, s, "mog1_update_kernel");
        const dim3 grid(blocks_for(npix)), block(bgs::kBlock);
        if (C == 3) hipLaunchKernelGGL((bgs::mog1_update_kernel<3>), grid, block, 0, s, m);
        if (C == 1) hipLaunchKernelGGL((bgs::mog1_update_kernel<1>), grid, block, 0, s, m);
      }
      if (nframes == 1)  // re-initialisation restarts the count (the streams of a run may otherwise have different ages: launch_key)
        for (int i = first; i < first + count; ++i) e->seen[i] = 0;
      flags = BGS_FG_VALID;  // BackgroundSubtractorMOG has no getBackgroundImage (MixtureOfGaussianV1BGS.cpp:53)
      break;
    }
    case BGS_MOG2: {
      double lr = p.alpha;
      int64_t nframes = t;
      bgs::Mog2Args m{};
      m.state_off = off, m.npix = npix;
      mog2_fill_args(e, m, 0.0);
      if (nframes == 0 || lr >= 1) {  // needToInitialize: bgmodel = zeros, modesUsed = 0
        mog2_clear(e, m, s);
        nframes = 0;
      }
      ++nframes;
      const int64_t n2 = 2 * nframes;
      lr = (lr >= 0 && nframes > 1) ? lr : 1. / (double)std::min<int64_t>(n2, p.mog2_history);
      mog2_fill_args(e, m, lr);
      m.frame = d_frames, m.fg = d_fg, m.bgimg = d_bg, m.fg_bits = d_bits;
      int rc = launch_mog2(e, m, s);
      if (rc) return rc;
      if (nframes == 1)  // re-initialisation restarts the count (the streams of a run may otherwise have different ages: launch_key)
        for (int i = first; i < first + count; ++i) e->seen[i] = 0;
      flags = BGS_FG_VALID | BGS_BG_VALID;
      break;
    }
    default: return fail(BGS_ERR_UNSUPPORTED, "algorithm %d is not implemented in this build", (int)e->algo);
  }
  HIP_TRY(hipGetLastError());
  for (int i = first; i < first + count; ++i) e->seen[i]++, e->rpos[i]++, e->last_flags[i] = flags;
  if (out_flags) *out_flags = flags;
  return BGS_OK;
}


// bgs_process_clip_device: `nframes` consecutive frames of streams [first, first+count).  Every algorithm: frame by frame
// through process_range (the same launches as nframes range calls).  MOG2: runs of 8 / 4 / 2 frames go through ONE launch
// that keeps the model in registers (kernel_mog2.h); what is left over takes the single-frame kernel.
int process_clip_run(bgs_engine* e, int first, int count, int slab_count, int nframes, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits,
                     hipStream_t s, uint32_t* out_flags);

// The streams of a clip call may have different ages too: one pass per run of streams with the same age and launch arguments
// (launch_key); the frames of a run sit inside the caller's [nframes][count] slab, so a run keeps the slab's strides.
int process_clip(bgs_engine* e, int first, int count, int nframes, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits, hipStream_t s,
                 uint32_t* out_flags) {
  if (nframes < 1) return fail(BGS_ERR_INVALID, "nframes must be >= 1");
  if (!e->n) return fail(BGS_ERR_INVALID, "geometry not set: call bgs_set_geometry or bgs_process first");
  if (first < 0 || count <= 0 || first + count > e->S) return fail(BGS_ERR_INVALID, "stream range [%d,%d) outside 0..%d", first, first + count, e->S);
  if (!d_frames) return fail(BGS_ERR_INVALID, "d_frames is NULL");
  const size_t W = (e->n + 63) / 64;
  const bool ragged = d_bits && (e->n % 64) != 0;  // as in process_range: byte masks first, then mask_pack_ragged_kernel
  if (d_bits && !d_fg && (ragged || e->algo == BGS_GMG || e->algo == BGS_ASBL)) {
    int rc = pack_scratch(e, (size_t)nframes * count * e->n, s);
    if (rc) return rc;
    d_fg = e->pack_fg;
  }
  const size_t C = (size_t)e->ch, bgC = e->algo == BGS_ASBL ? 1 : C;
  std::vector<uint32_t> fl((size_t)nframes), all((size_t)nframes, ~0u);
  for (int a = first; a < first + count;) {
    int b = a + 1;
    while (b < first + count && e->seen[b] == e->seen[a] && launch_key(e, b) == launch_key(e, a)) ++b;
    const size_t o = (size_t)(a - first) * e->n;
    int rc = process_clip_run(e, a, b - a, count, nframes, d_frames + o * C, d_fg ? d_fg + o : nullptr, d_bg ? d_bg + o * bgC : nullptr, (d_bits && !ragged) ? d_bits + (size_t)(a - first) * W : nullptr, s, fl.data());
    if (rc) return rc;
    if (ragged)
      for (int t = 0; t < nframes; ++t)
        if (fl[t] & BGS_FG_VALID) pack_ragged(e, d_fg + (size_t)t * count * e->n + o, d_bits + ((size_t)t * count + (size_t)(a - first)) * W, (size_t)(b - a), s);
    for (int t = 0; t < nframes; ++t) all[t] &= fl[t];
    a = b;
  }
  if (out_flags)
    for (int t = 0; t < nframes; ++t) out_flags[t] = all[t];
  return BGS_OK;
}

// `count` streams of one age starting at `first`, inside a slab of `slab_count` streams per frame
int process_clip_run(bgs_engine* e, int first, int count, int slab_count, int nframes, const uint8_t* d_frames, uint8_t* d_fg, uint8_t* d_bg, uint64_t* d_bits,
                     hipStream_t s, uint32_t* out_flags) {
  const bgs_params& p = e->p;
  const size_t npix = e->n * count, C = (size_t)e->ch;
  const size_t slab = e->n * slab_count;  // pixels from one frame of the clip to the next
  const size_t words = slab / 64;
  // lr >= 1 re-initialises the model on every frame (needToInitialize): nothing to keep in registers
  const bool dp_gmm = e->algo == BGS_DP_ZIVKOVIC_AGMM || e->algo == BGS_DP_GRIMSON_GMM;
  const bool fuse_ok = e->clip_fuse && (dp_gmm || ((e->algo == BGS_MOG2 || e->algo == BGS_MOG1) && p.alpha < 1));
  // FrameDifference / WeightedMoving*: frame t needs frames t-1 (t-2).  A per-frame call copies its frame into the engine's ring; inside
  // a clip the earlier frames of the clip ARE that history, so only the last one (two) are copied into the ring, once, at the end.
  const bool ring_clip = e->nring > 0 && !e->borrow && nframes >= 2;
  const uint8_t* saved_borrowed[2] = {e->borrowed[0], e->borrowed[1]};
  const int64_t ring_t0 = e->seen[first], ring_p0 = e->rpos[first];
  if (ring_clip) {
    const int R = e->nring;
    const size_t offb = e->n * (size_t)first * C;
    e->borrow = true, e->borrow_in_clip = true;
    e->borrowed[0] = ring_t0 >= 1 ? e->ring[(ring_p0 + R - 1) % R] + offb : nullptr;
    e->borrowed[1] = (ring_t0 >= 2 && R == 3) ? e->ring[(ring_p0 + R - 2) % R] + offb : nullptr;
  }
  auto end_ring_clip = [&](int done) -> int {  // `done` frames of the clip went through: leave the ring as per-frame calls would have
    if (!ring_clip) return BGS_OK;
    e->borrow = false, e->borrow_in_clip = false;
    e->borrowed[0] = saved_borrowed[0], e->borrowed[1] = saved_borrowed[1];
    const int R = e->nring;
    const size_t offb = e->n * (size_t)first * C;
    for (int k = 1; k < R; ++k) {
      const int j = done - k;  // clip frame j is the k-th last one
      if (j < 0) break;        // older ones are in the ring already
      HIP_TRY(hipMemcpyAsync(e->ring[(ring_p0 + j) % R] + offb, d_frames + (size_t)j * slab * C, npix * C, hipMemcpyDeviceToDevice, s));
    }
    return BGS_OK;
  };
  int t = 0;
  while (t < nframes) {
    const int left = nframes - t;
    const int fuse = !fuse_ok ? 1 : left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
    const uint8_t* fr = d_frames + (size_t)t * slab * C;
    uint8_t* fg = d_fg ? d_fg + (size_t)t * slab : nullptr;
    uint8_t* bg = d_bg ? d_bg + (size_t)t * slab * (e->algo == BGS_ASBL ? 1 : C) : nullptr;
    uint64_t* bits = d_bits ? d_bits + (size_t)t * words : nullptr;
    if (fuse == 1) {
      int rc = process_range(e, first, count, fr, fg, bg, bits, s, out_flags ? out_flags + t : nullptr);
      if (rc) {
        (void)end_ring_clip(t);
        return rc;
      }
    } else {
      const int64_t seen = e->seen[first];
      for (int i = first; i < first + count; ++i)
        if (e->seen[i] != seen) return fail(BGS_ERR_INVALID, "streams %d and %d are not in lock-step (%lld vs %lld frames)", first, i, (long long)seen, (long long)e->seen[i]);
      HIP_TRY(hipSetDevice(e->device));
      if (dp_gmm) {  // package_bgs/dp GMMs: the same kernel with a frame loop (kernel_dp.h)
        uint32_t fl = 0;
        int rc = dp_process(e, first, count, seen, fr, fg, bits, s, &fl, fuse, slab);
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
        for (int i = first; i < first + count; ++i) e->seen[i] += fuse, e->rpos[i] += fuse, e->last_flags[i] = fl;
        if (out_flags)
          for (int j = 0; j < fuse; ++j) out_flags[t + j] = fl;
        t += fuse;
        continue;
      }
      if (e->algo == BGS_MOG1) {
        bgs::Mog1ClipArgs c{};
        c.m.state = e->mog1_state, c.m.state_off = e->n * first, c.m.npix = npix;
        if (seen == 0) {  // needToInitialize on a stream's first frame
          if (C == 3)
            hipLaunchKernelGGL((bgs::mog1_clear_kernel<3>), dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, c.m);
          else
            hipLaunchKernelGGL((bgs::mog1_clear_kernel<1>), dim3(blocks_for(npix)), dim3(bgs::kBlock), 0, s, c.m);
        }
        c.m.frame = fr, c.m.fg = fg, c.m.fg_bits = bits;
        mog1_fill_args(e, c.m, 0.0);
        for (int j = 0; j < fuse; ++j) {
          const int64_t nf = seen + j + 1;
          c.alpha[j] = (float)((p.alpha >= 0 && nf > 1) ? p.alpha : 1. / (double)std::min<int64_t>(nf, p.mog1_history));
        }
        c.frame_stride = slab * C, c.fg_stride = slab, c.bits_stride = words;
        {
          Timed tm(e, s, "mog1_clip_kernel");
          const dim3 grid(blocks_for(npix)), block(bgs::kBlock);
#define MOG1_CLIP_CASE(CV, TV) \
  if (C == CV && fuse == TV) hipLaunchKernelGGL((bgs::mog1_clip_kernel<CV, TV>), grid, block, 0, s, c);
          MOG1_CLIP_CASE(3, 2) MOG1_CLIP_CASE(3, 4) MOG1_CLIP_CASE(3, 8) MOG1_CLIP_CASE(1, 2) MOG1_CLIP_CASE(1, 4) MOG1_CLIP_CASE(1, 8)
#undef MOG1_CLIP_CASE
        }
        HIP_TRY(hipGetLastError());
        for (int i = first; i < first + count; ++i) e->seen[i] += fuse, e->rpos[i] += fuse, e->last_flags[i] = BGS_FG_VALID;
        if (out_flags)
          for (int j = 0; j < fuse; ++j) out_flags[t + j] = BGS_FG_VALID;
        t += fuse;
        continue;
      }
      bgs::Mog2ClipArgs c{};
      c.m.state_off = e->n * first, c.m.npix = npix;
      mog2_fill_args(e, c.m, 0.0);
      if (seen == 0) mog2_clear(e, c.m, s);  // needToInitialize on a stream's first frame
      for (int j = 0; j < fuse; ++j) {       // the learning rate of each frame, as the single-frame path computes it
        const int64_t nf = seen + j + 1;
        const double lr = (p.alpha >= 0 && nf > 1) ? p.alpha : 1. / (double)std::min<int64_t>(2 * nf, p.mog2_history);
        c.alphaT[j] = (float)lr, c.alpha1[j] = 1.f - c.alphaT[j], c.prune[j] = (float)(-lr * (double)p.mog2_ct);
      }
      c.m.frame = fr, c.m.fg = fg, c.m.bgimg = bg, c.m.fg_bits = bits;
      c.frame_stride = slab * 3, c.fg_stride = slab, c.bg_stride = slab * 3, c.bits_stride = words;
      int rc = launch_mog2_clip(e, c, fuse, s);
      if (rc) return rc;
      HIP_TRY(hipGetLastError());
      for (int i = first; i < first + count; ++i) e->seen[i] += fuse, e->rpos[i] += fuse, e->last_flags[i] = BGS_FG_VALID | BGS_BG_VALID;
      if (out_flags)
        for (int j = 0; j < fuse; ++j) out_flags[t + j] = BGS_FG_VALID | BGS_BG_VALID;
    }
    t += fuse;
  }
  return end_ring_clip(nframes);
}

}  // namespace

// N3 (bgs_ingest_*, bgs_set_ingest): geometry of the frame preparation
namespace {
struct IngestPlan {
  int rw, rh, rows, cols, x0, y0, mode;
  double scale_x, scale_y;
};
int ingest_plan(const bgs_ingest* c, int src_rows, int src_cols, IngestPlan* pl) {
  if (!c || c->struct_size != sizeof(bgs_ingest)) return fail(BGS_ERR_INVALID, "bgs_ingest.struct_size mismatch");
  if (src_rows <= 0 || src_cols <= 0 || c->resize_percent <= 0) return fail(BGS_ERR_INVALID, "bad ingest geometry");
  pl->rw = (int)(((int64_t)src_cols * c->resize_percent) / 100), pl->rh = (int)(((int64_t)src_rows * c->resize_percent) / 100);  // VideoCapture.cpp:142
  if (pl->rw < 1 || pl->rh < 1) return fail(BGS_ERR_INVALID, "resize to %d %% leaves no pixels", c->resize_percent);
  const bool roi = c->roi_x1 > c->roi_x0 && c->roi_y1 > c->roi_y0;
  pl->x0 = roi ? c->roi_x0 : 0, pl->y0 = roi ? c->roi_y0 : 0;
  pl->cols = roi ? c->roi_x1 - c->roi_x0 : pl->rw, pl->rows = roi ? c->roi_y1 - c->roi_y0 : pl->rh;
  if (pl->x0 < 0 || pl->y0 < 0 || pl->x0 + pl->cols > pl->rw || pl->y0 + pl->rows > pl->rh)
    return fail(BGS_ERR_INVALID, "ROI (%d,%d)-(%d,%d) outside the %dx%d frame (cvSetImageROI would fail)", c->roi_x0, c->roi_y0, c->roi_x1, c->roi_y1, pl->rw, pl->rh);
  pl->scale_x = 1. / ((double)pl->rw / src_cols), pl->scale_y = 1. / ((double)pl->rh / src_rows);
  const int isx = (int)std::lrint(pl->scale_x), isy = (int)std::lrint(pl->scale_y);
  const bool area_fast = std::fabs(pl->scale_x - isx) < 2.220446049250313e-16 && std::fabs(pl->scale_y - isy) < 2.220446049250313e-16;
  pl->mode = (pl->rw == src_cols && pl->rh == src_rows) ? 0 : (area_fast && isx == 2 && isy == 2) ? 2 : 1;
  return BGS_OK;
}
// cv::getGaussianKernel(7, 1.5, CV_32F) -> the integer kernel of the 8-bit fixed-point filter (R3, kernel_ingest.h)
void gaussian7_kernel(int ik[7]) {
  float cf[7];
  const double sigma = 1.5, scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < 7; ++i) {
    const double x = i - 3.0;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; ++i) {
    cf[i] = (float)(cf[i] * sum);
    ik[i] = (int)std::lrintf(cf[i] * 256.f);
  }
}
}  // namespace

extern "C" {

int bgs_abi_version(void) { return BGS_ABI_VERSION; }

const char* bgs_last_error(void) { return g_err.c_str(); }

int bgs_default_params(bgs_algo algo, bgs_params* p) {
  if (!p) return fail(BGS_ERR_INVALID, "params is NULL");
  if ((int)algo < 0 || algo >= BGS_ALGO_COUNT) return fail(BGS_ERR_INVALID, "unknown algorithm %d", (int)algo);
  std::memset(p, 0, sizeof(*p));
  p->struct_size = (uint32_t)sizeof(*p);
  p->enable_threshold = 1;
  p->threshold = (algo == BGS_ASBL) ? 25 : 15;
  p->enable_weight = 1;
  p->alpha = 0.05;
  p->limit = -1;
  p->learning_frames = 90;
  p->alpha_learn = 0.05;
  p->alpha_detection = 0.05;
  p->mog2_history = 500;
  p->mog2_nmixtures = 5;
  p->mog2_var_threshold = 16.f;
  p->mog2_background_ratio = 0.9f;
  p->mog2_var_threshold_gen = 9.f;
  p->mog2_var_init = 15.f;
  p->mog2_var_min = 4.f;
  p->mog2_var_max = 75.f;
  p->mog2_ct = 0.05f;
  p->mog2_tau = 0.5f;
  p->mog2_detect_shadows = 1;
  p->mog2_shadow_value = 127;
  p->mog1_history = 200;
  p->mog1_nmixtures = 5;
  p->mog1_background_ratio = 0.7;
  p->mog1_var_threshold = 2.5 * 2.5;
  p->mog1_noise_sigma = 30 * 0.5;
  p->lbsp_rel_threshold = 0.333f;
  p->lbsp_threshold_offset = 0;
  p->subsense_min_color_dist_threshold = 30;
  p->subsense_n_samples = 50;
  p->subsense_n_required = 2;
  p->subsense_samples_for_moving_avgs = 100;
  p->subsense_desc_dist_threshold_offset = 3;
  p->gmg_max_features = 64;
  p->gmg_init_frames = 20;
  p->gmg_quantization_levels = 16;
  p->gmg_smoothing_radius = 7;
  p->gmg_update_background_model = 1;
  p->gmg_learning_rate = 0.025;
  p->gmg_background_prior = 0.8;
  p->gmg_decision_threshold = 0.7;
  p->sd_amp_factor = 1;
  p->sd_min_var = 15;
  p->sd_max_var = 255;
  // package_bgs/dp wrappers, DP*BGS.cpp:19
  p->dp_gaussians = 3;
  p->dp_sampling_rate = 7;
  switch (algo) {
    case BGS_DP_ZIVKOVIC_AGMM: p->dp_threshold = 25.0f, p->dp_alpha = 0.001f; break;
    case BGS_DP_GRIMSON_GMM: p->dp_threshold = 9.0f, p->dp_alpha = 0.01f; break;
    case BGS_DP_WREN_GA: p->dp_threshold = 12.25f, p->dp_alpha = 0.005f, p->learning_frames = 30; break;
    case BGS_DP_MEAN: p->dp_threshold = 2700.0f, p->dp_alpha = 1e-6f, p->learning_frames = 30; break;
    case BGS_DP_ADAPTIVE_MEDIAN: p->dp_threshold = 40.0f, p->learning_frames = 30; break;
    case BGS_LOBSTER:  // BackgroundSubtractorLOBSTER.h:6-16
      p->lbsp_rel_threshold = 0.365f, p->subsense_desc_dist_threshold_offset = 4, p->subsense_min_color_dist_threshold = 30;
      p->subsense_n_samples = 35, p->subsense_n_required = 2;
      break;
    default: break;
  }
  return BGS_OK;
}

int bgs_create(bgs_algo algo, const bgs_params* params, int hip_device, int n_streams, bgs_engine** out) {
  if (!out) return fail(BGS_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if ((int)algo < 0 || algo >= BGS_ALGO_COUNT) return fail(BGS_ERR_INVALID, "unknown algorithm %d", (int)algo);
  if (n_streams < 1) return fail(BGS_ERR_INVALID, "n_streams must be >= 1");
  if (params && params->struct_size != sizeof(bgs_params)) return fail(BGS_ERR_INVALID, "bgs_params.struct_size %u != %zu (ABI mismatch)", params->struct_size, sizeof(bgs_params));
  bgs_engine* e = new (std::nothrow) bgs_engine();
  if (!e) return fail(BGS_ERR_NOMEM, "out of host memory");
  e->algo = algo;
  if (params)
    e->p = *params;
  else
    bgs_default_params(algo, &e->p);
  int rc = check_params(algo, e->p);
  if (rc) {
    delete e;
    return rc;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    delete e;
    return fail(BGS_ERR_HIP, "no HIP device visible: libbgs_hip has no CPU path");
  }
  if (hip_device < 0 || hip_device >= ndev) {
    delete e;
    return fail(BGS_ERR_INVALID, "hip_device %d outside 0..%d", hip_device, ndev - 1);
  }
  e->device = hip_device;
  e->S = n_streams;
  e->seen.assign(n_streams, 0);
  e->rpos.assign(n_streams, 0);
  e->pin.assign((size_t)n_streams * 3, bgs_engine::HostPin());
  e->lanes.assign(n_streams, bgs_engine::Lane());
  e->last_flags.assign(n_streams, 0);
  e->counter.assign(n_streams, 0);
  e->flip.assign(n_streams, 0);
  if (const char* env = getenv("BGS_MOG2_COMPLETE")) e->mog2_complete = atoi(env) != 0;
  if (const char* env = getenv("BGS_XCD_SWIZZLE")) e->xcd_swizzle = atoi(env);
  if (const char* env = getenv("BGS_MOG2_SPARSE")) e->mog2_sparse = atoi(env);
  if (const char* env = getenv("BGS_MODEL_CHUNK_MB")) e->model_chunk_mb = atoi(env);
  if (const char* env = getenv("BGS_MODEL_CHUNK_MIN_MB")) e->model_chunk_min_bytes = (size_t)std::max(atoi(env), 0) << 20;
  if (const char* env = getenv("BGS_CLIP_FUSE")) e->clip_fuse = atoi(env) != 0;
  if (const char* env = getenv("BGS_DEBUG_POISON")) e->poison = atoi(env) != 0;
  if (const char* env = getenv("BGS_HOST_REGISTER")) e->host_register = atoi(env) & 7;
  *out = e;
  return BGS_OK;
}

int bgs_set_params(bgs_engine* e, const bgs_params* params) {
  if (!e || !params) return fail(BGS_ERR_INVALID, "NULL argument");
  if (params->struct_size != sizeof(bgs_params)) return fail(BGS_ERR_INVALID, "bgs_params.struct_size mismatch");
  int rc = check_params(e->algo, *params);
  if (rc) return rc;
  const bgs_params old = e->p;
  e->p = *params;
  if (e->n) {
    // Parameters the reference hands to its model object once, when it is built on the first frame, stay as they were:
    // later values are ignored there too (SuBSENSE.cpp:27-36, LOBSTER.cpp:27-34, DP*BGS.cpp `if(firstTime)`), and here they
    // also size the device buffers.
    bgs_params& p = e->p;
    if (e->algo == BGS_SUBSENSE || e->algo == BGS_LOBSTER) {
      p.lbsp_rel_threshold = old.lbsp_rel_threshold, p.lbsp_threshold_offset = old.lbsp_threshold_offset;
      p.subsense_min_color_dist_threshold = old.subsense_min_color_dist_threshold, p.subsense_n_samples = old.subsense_n_samples;
      p.subsense_n_required = old.subsense_n_required, p.subsense_samples_for_moving_avgs = old.subsense_samples_for_moving_avgs;
      p.subsense_desc_dist_threshold_offset = old.subsense_desc_dist_threshold_offset;
    }
    if (is_dp(e->algo)) {
      p.dp_threshold = old.dp_threshold, p.dp_alpha = old.dp_alpha, p.dp_gaussians = old.dp_gaussians;
      p.dp_sampling_rate = old.dp_sampling_rate, p.learning_frames = old.learning_frames;
    }
    if (e->algo == BGS_GMG) p.gmg_max_features = old.gmg_max_features;  // sizes the histogram planes
    if (e->algo == BGS_ASBL && (!e->abl_lut_valid || p.alpha_learn != e->asbl_lut_alpha[0] || p.alpha_detection != e->asbl_lut_alpha[1])) {
      if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(BGS_ERR_HIP, "device sync failed");
      rc = asbl_build_lut(e);
      if (rc) return rc;
    }
    if (e->algo == BGS_ABL && (!e->abl_lut_valid || p.alpha != e->abl_lut_alpha)) {
      // a launch still in flight on some stream may be reading the table: let the device drain before it is rewritten
      if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(BGS_ERR_HIP, "device sync failed");
      rc = abl_build_lut(e);
      if (rc) return rc;
    }
  }
  return BGS_OK;
}

int bgs_set_geometry(bgs_engine* e, int rows, int cols, int channels) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (e->n) {
    if (rows == e->rows && cols == e->cols && channels == e->ch) return BGS_OK;
    return fail(BGS_ERR_GEOMETRY, "engine is %dx%dx%d, asked for %dx%dx%d", e->rows, e->cols, e->ch, rows, cols, channels);
  }
  int rc = allocate(e, rows, cols, channels);
  if (rc) {
    free_all(e);
    e->n = 0;
  }
  return rc;
}

// option 1: borrow the caller's frame buffers as history (device path of FD/WMM/WMV): the buffers passed to the
// previous one (FD) or two (WMM/WMV) bgs_process_batch_device calls must stay valid and unchanged.
int bgs_set_option(bgs_engine* e, int option, int64_t value) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  switch (option) {
    case 1: e->borrow = value != 0; return BGS_OK;
    case 2:  // round-2 A/B knobs (pixels per lane, planar layout): accepted and ignored since the slot layout of round 3
    case 3: return BGS_OK;
    case 4: e->xcd_swizzle = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return BGS_OK;
    case 6: e->mog2_sparse = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 4); return BGS_OK;
    case 7: e->clip_fuse = value != 0; return BGS_OK;
    case 8:
      if (hipSetDevice(e->device) == hipSuccess && e->stream) (void)hipStreamSynchronize(e->stream);
      (void)hipDeviceSynchronize();
      for (size_t i = 0; i < e->pin.size(); ++i)
        if (!((value >> (i % 3)) & 1) && e->pin[i].pinned) (void)hipHostUnregister(const_cast<void*>(e->pin[i].ptr)), e->pin[i] = bgs_engine::HostPin();
      e->host_register = (int)(value & 7);
      return BGS_OK;
    case 5: return BGS_OK;  // BGS_OPT_PLACEMENT_PROBE of rounds 1-2: accepted and ignored (placement is deterministic since round 3: model_allocate)
    case 9:
      if (e->n) return fail(BGS_ERR_INVALID, "the model is allocated when the geometry is set");
      e->model_chunk_mb = (int)std::max<int64_t>(value, 0);
      return BGS_OK;
    case 10:
      if (e->n) return fail(BGS_ERR_INVALID, "the model is allocated when the geometry is set");
      e->model_chunk_min_bytes = (size_t)std::max<int64_t>(value, 0) << 20;
      return BGS_OK;
    default: return fail(BGS_ERR_INVALID, "unknown option %d", option);
  }
}

// A bgs_submit still in flight runs on its lane's own non-blocking HIP stream: a device-path call over the same camera would race
// with it on that camera's model, ring slot and frame count.  The device entry points refuse instead (bgs_wait first).
static int range_busy(const bgs_engine* e, int first, int count) {
  for (int i = std::max(first, 0); i < first + count && i < e->S; ++i)
    if (e->lanes[i].pending) return fail(BGS_ERR_STATE, "stream %d has a bgs_submit in flight: bgs_wait before a device-path call that covers it", i);
  return BGS_OK;
}

int bgs_process_range_device(bgs_engine* e, int first, int count, const void* d_frames, void* d_fg, void* d_bg, void* d_fg_bits, void* hip_stream,
                             uint32_t* out_flags) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (int rc = range_busy(e, first, count)) return rc;
  return process_range(e, first, count, (const uint8_t*)d_frames, (uint8_t*)d_fg, (uint8_t*)d_bg, (uint64_t*)d_fg_bits, (hipStream_t)hip_stream, out_flags);
}

int bgs_process_clip_device(bgs_engine* e, int first, int count, int nframes, const void* d_frames, void* d_fg, void* d_bg, void* d_fg_bits,
                            void* hip_stream, uint32_t* out_flags) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (int rc = range_busy(e, first, count)) return rc;
  return process_clip(e, first, count, nframes, (const uint8_t*)d_frames, (uint8_t*)d_fg, (uint8_t*)d_bg, (uint64_t*)d_fg_bits, (hipStream_t)hip_stream, out_flags);
}

int bgs_process_batch_device(bgs_engine* e, const void* d_frames, void* d_fg, void* d_bg, void* d_fg_bits, void* hip_stream, uint32_t* out_flags) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (int rc = range_busy(e, 0, e->S)) return rc;
  return process_range(e, 0, e->S, (const uint8_t*)d_frames, (uint8_t*)d_fg, (uint8_t*)d_bg, (uint64_t*)d_fg_bits, (hipStream_t)hip_stream, out_flags);
}

static int lane_wait(bgs_engine* e, int stream, uint32_t* out_flags);  // bgs_submit / bgs_wait, below

int bgs_process(bgs_engine* e, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step, uint8_t* bg,
                size_t bg_step, uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  if (!in || rows <= 0 || cols <= 0) return BGS_OK;  // if(img_input.empty()) return;
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  if (in_step < (size_t)cols * channels) return fail(BGS_ERR_INVALID, "in_step %zu < cols*channels", in_step);
  // N3: with bgs_set_ingest the caller hands over the RAW captured frame; everything below runs in the prepared geometry
  IngestPlan pl{};
  const bool ingest = e->ingest_on;
  const int raw_rows = rows, raw_cols = cols;
  if (ingest) {
    int rc0 = ingest_plan(&e->ingest, rows, cols, &pl);
    if (rc0) return rc0;
    if (e->ingest.equalize_hist && channels != 1) return fail(BGS_ERR_UNSUPPORTED, "equalizeHist needs a 1-channel frame (cv::equalizeHist asserts CV_8UC1, PreProcessor.cpp:64)");
    if (e->raw_rows && (e->raw_rows != rows || e->raw_cols != cols)) return fail(BGS_ERR_GEOMETRY, "raw frames were %dx%d, now %dx%d", e->raw_rows, e->raw_cols, rows, cols);
    e->raw_rows = rows, e->raw_cols = cols;
    rows = pl.rows, cols = pl.cols;
  }
  const bool ingest_on_device = ingest && (pl.mode != 0 || e->ingest.equalize_hist || e->ingest.gaussian_blur);
  int rc = bgs_set_geometry(e, rows, cols, channels);
  if (rc) return rc;
  if (e->lanes[stream].pending) {  // a bgs_submit of this stream is still in flight: its result is collected (and dropped) first
    rc = lane_wait(e, stream, nullptr);
    if (rc) return rc;
  }
  rc = ensure_staging(e);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  const size_t rb = (size_t)cols * channels, fb = e->n * channels;
  // history-keeping algorithms receive the upload straight in their ring slot (zero-copy history)
  uint8_t* dst = e->d_in;
  if (e->nring) dst = e->ring[e->rpos[stream] % e->nring] + (size_t)stream * fb;
  if (ingest_on_device) {
    // resize / equalizeHist / GaussianBlur: the raw frame goes up as it is, the preparation runs between the upload and the model kernel
    const size_t raw_rb = (size_t)raw_cols * channels, raw_bytes = raw_rb * raw_rows;
    if (!e->h_raw) {
      HIP_TRY(hipHostMalloc((void**)&e->h_raw, raw_bytes, hipHostMallocDefault));
      HIP_TRY(hipMalloc((void**)&e->d_raw, raw_bytes));
      const size_t ws = bgs_ingest_workspace(&e->ingest, 1, raw_rows, raw_cols, channels);
      if (ws) HIP_TRY(hipMalloc((void**)&e->d_ingest_ws, ws));
    }
    const int bands = raw_rows >= 64 ? 8 : 1;
    for (int b = 0; b < bands; ++b) {
      const int y0 = (int)((int64_t)raw_rows * b / bands), y1 = (int)((int64_t)raw_rows * (b + 1) / bands);
      for (int y = y0; y < y1; ++y) std::memcpy(e->h_raw + (size_t)y * raw_rb, in + (size_t)y * in_step, raw_rb);
      HIP_TRY(hipMemcpyAsync(e->d_raw + (size_t)y0 * raw_rb, e->h_raw + (size_t)y0 * raw_rb, (size_t)(y1 - y0) * raw_rb, hipMemcpyHostToDevice, e->stream));
    }
    rc = bgs_ingest_device(e->device, &e->ingest, e->d_raw, 1, raw_rows, raw_cols, channels, raw_rb, dst, e->d_ingest_ws, e->stream);
    if (rc) return rc;
  } else {
    // staging is pipelined: while the DMA engine moves band k, the CPU copies band k+1 of the caller's (pageable, possibly
    // strided) image into the pinned buffer.  Flip (rows reversed) and ROI (a window of the raw frame) cost nothing extra: they
    // only change which source row and column each staged row starts at.
    e->diag_frames++, e->diag_h2d_bytes += (int64_t)fb;
    if (!ingest && in_step == rb && host_pin(e, stream, 0, in, fb)) {
      HIP_TRY(hipMemcpyAsync(dst, in, fb, hipMemcpyHostToDevice, e->stream));  // straight from the caller's page-locked frame buffer
    } else {
      const auto st0 = std::chrono::steady_clock::now();
      const int bands = rows >= 64 ? 8 : 1;
      for (int b = 0; b < bands; ++b) {
        const int y0 = (int)((int64_t)rows * b / bands), y1 = (int)((int64_t)rows * (b + 1) / bands);
        for (int y = y0; y < y1; ++y) {
          size_t sy = (size_t)y, sx = 0;
          if (ingest) sy = (size_t)(e->ingest.flip ? pl.rh - 1 - (y + pl.y0) : y + pl.y0), sx = (size_t)pl.x0 * channels;
          std::memcpy(e->h_in + (size_t)y * rb, in + sy * in_step + sx, rb);
        }
        HIP_TRY(hipMemcpyAsync(dst + (size_t)y0 * rb, e->h_in + (size_t)y0 * rb, (size_t)(y1 - y0) * rb, hipMemcpyHostToDevice, e->stream));
      }
      e->diag_stage_in_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st0).count();
    }
  }
  uint32_t flags = 0;
  const bool saved_borrow = e->borrow;
  e->borrow = false;
  e->last_fg_stream = -1;
  rc = process_range(e, stream, 1, dst, e->d_fg, bg ? e->d_bg : nullptr, nullptr, e->stream, &flags);  // the mask stays on the device for bgs_last_mask_blobs
  e->borrow = saved_borrow;
  if (rc) return rc;
  if (flags & BGS_FG_VALID) e->last_fg_stream = stream;
  const int bg_ch = e->algo == BGS_ASBL ? 1 : channels;
  // the way back is pipelined the same way when there is a background image to return (6 MB at 1080p): band k is copied out to
  // the caller's image while band k+1 is still on the bus; a mask alone (2 MB) is not worth the events
  bool out_fg = fg && (flags & BGS_FG_VALID), out_bg = bg && (flags & BGS_BG_VALID);
  // outputs the caller keeps allocated (same buffer as last call, contiguous rows) are written by the DMA engine directly
  const bool fg_direct = out_fg && fg_step == (size_t)cols && host_pin(e, stream, 1, fg, e->n);
  const bool bg_direct = out_bg && bg_step == (size_t)cols * bg_ch && host_pin(e, stream, 2, bg, e->n * bg_ch);
  e->diag_d2h_bytes += (int64_t)((out_fg ? e->n : 0) + (out_bg ? e->n * bg_ch : 0));
  if (fg_direct) HIP_TRY(hipMemcpyAsync(fg, e->d_fg, e->n, hipMemcpyDeviceToHost, e->stream));
  if (bg_direct) HIP_TRY(hipMemcpyAsync(bg, e->d_bg, e->n * bg_ch, hipMemcpyDeviceToHost, e->stream));
  if (fg_direct) out_fg = false;
  if (bg_direct) out_bg = false;
  if (!out_fg && !out_bg) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (out_flags) *out_flags = flags;
    return BGS_OK;
  }
  const int obands = out_bg ? (rows >= 64 ? 8 : 1) : 1;
  for (int b = 0; b < obands; ++b) {
    const int y0 = (int)((int64_t)rows * b / obands), y1 = (int)((int64_t)rows * (b + 1) / obands);
    if (out_fg) HIP_TRY(hipMemcpyAsync(e->h_fg + (size_t)y0 * cols, e->d_fg + (size_t)y0 * cols, (size_t)(y1 - y0) * cols, hipMemcpyDeviceToHost, e->stream));
    if (out_bg)
      HIP_TRY(hipMemcpyAsync(e->h_bg + (size_t)y0 * cols * bg_ch, e->d_bg + (size_t)y0 * cols * bg_ch, (size_t)(y1 - y0) * cols * bg_ch, hipMemcpyDeviceToHost, e->stream));
    if (!e->band_ev[b]) HIP_TRY(hipEventCreateWithFlags(&e->band_ev[b], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->band_ev[b], e->stream));
  }
  for (int b = 0; b < obands; ++b) {
    const int y0 = (int)((int64_t)rows * b / obands), y1 = (int)((int64_t)rows * (b + 1) / obands);
    HIP_TRY(hipEventSynchronize(e->band_ev[b]));
    if (out_fg)
      for (int y = y0; y < y1; ++y) std::memcpy(fg + (size_t)y * fg_step, e->h_fg + (size_t)y * cols, (size_t)cols);
    if (out_bg)
      for (int y = y0; y < y1; ++y) std::memcpy(bg + (size_t)y * bg_step, e->h_bg + (size_t)y * cols * bg_ch, (size_t)cols * bg_ch);
  }
  if (out_flags) *out_flags = flags;
  return BGS_OK;
}

// ---- several cameras on the host path: bgs_submit queues one frame of one stream and returns, bgs_wait collects it ----------
// Each camera has its own lane (HIP stream, pinned staging, device images), so while camera A's frame is on the bus camera B's
// kernel runs and camera C's mask comes back: the copy engines and the CUs work side by side instead of taking turns, which is all
// a synchronous bgs_process per camera can do.  One submission per stream in flight; bgs_process on a stream first collects it.
static int lane_wait(bgs_engine* e, int stream, uint32_t* out_flags) {
  bgs_engine::Lane& ln = e->lanes[stream];
  if (out_flags) *out_flags = 0;
  if (!ln.pending) return BGS_OK;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipEventSynchronize(ln.done));
  ln.pending = false;
  const int bg_ch = e->algo == BGS_ASBL ? 1 : e->ch;
  const bool copy_fg = ln.fg && !ln.fg_direct && (ln.flags & BGS_FG_VALID), copy_bg = ln.bg && !ln.bg_direct && (ln.flags & BGS_BG_VALID);
  const auto st0 = std::chrono::steady_clock::now();
  if (copy_fg)
    for (int y = 0; y < e->rows; ++y) std::memcpy(ln.fg + (size_t)y * ln.fg_step, ln.h_fg + (size_t)y * e->cols, (size_t)e->cols);
  if (copy_bg)
    for (int y = 0; y < e->rows; ++y) std::memcpy(ln.bg + (size_t)y * ln.bg_step, ln.h_bg + (size_t)y * e->cols * bg_ch, (size_t)e->cols * bg_ch);
  if (copy_fg || copy_bg) e->diag_stage_out_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st0).count();
  if (out_flags) *out_flags = ln.flags;
  return BGS_OK;
}

int bgs_submit(bgs_engine* e, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step, uint8_t* bg,
               size_t bg_step) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  if (e->ingest_on) return fail(BGS_ERR_UNSUPPORTED, "bgs_submit takes prepared frames (bgs_set_ingest is served by bgs_process)");
  if (e->lanes[stream].pending) return fail(BGS_ERR_STATE, "stream %d has a submission in flight: bgs_wait first", stream);
  if (!in || rows <= 0 || cols <= 0) return BGS_OK;  // if(img_input.empty()) return;
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  if (in_step < (size_t)cols * channels) return fail(BGS_ERR_INVALID, "in_step %zu < cols*channels", in_step);
  int rc = bgs_set_geometry(e, rows, cols, channels);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  bgs_engine::Lane& ln = e->lanes[stream];
  const size_t rb = (size_t)cols * channels, fb = e->n * channels;
  const int bg_ch = e->algo == BGS_ASBL ? 1 : channels;
  if (!ln.hs) {
    // all or nothing: a lane whose stream exists but whose buffers do not would send the next submission through null pointers
    hipError_t er = hipStreamCreateWithFlags(&ln.hs, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
    if (er == hipSuccess) er = hipHostMalloc((void**)&ln.h_in, fb, hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostMalloc((void**)&ln.h_fg, e->n, hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostMalloc((void**)&ln.h_bg, e->n * bg_ch, hipHostMallocDefault);
    if (er == hipSuccess) er = hipMalloc((void**)&ln.d_in, fb);
    if (er == hipSuccess) er = hipMalloc((void**)&ln.d_fg, e->n);
    if (er == hipSuccess) er = hipMalloc((void**)&ln.d_bg, e->n * bg_ch);
    if (er != hipSuccess) {
      (void)hipGetLastError();
      lane_release(ln);
      return fail(BGS_ERR_HIP, "bgs_submit: setting up the lane of stream %d failed: %s", stream, hipGetErrorString(er));
    }
  }
  uint8_t* dst = ln.d_in;
  if (e->nring) dst = e->ring[e->rpos[stream] % e->nring] + (size_t)stream * fb;  // history classes: straight into their ring slot
  e->diag_frames++, e->diag_h2d_bytes += (int64_t)fb;
  if (in_step == rb && host_pin(e, stream, 0, in, fb, ln.hs)) {
    HIP_TRY(hipMemcpyAsync(dst, in, fb, hipMemcpyHostToDevice, ln.hs));
  } else {
    const auto st0 = std::chrono::steady_clock::now();
    const int bands = rows >= 64 ? 8 : 1;  // the CPU stages band k+1 while the DMA engine moves band k
    for (int b = 0; b < bands; ++b) {
      const int y0 = (int)((int64_t)rows * b / bands), y1 = (int)((int64_t)rows * (b + 1) / bands);
      for (int y = y0; y < y1; ++y) std::memcpy(ln.h_in + (size_t)y * rb, in + (size_t)y * in_step, rb);
      HIP_TRY(hipMemcpyAsync(dst + (size_t)y0 * rb, ln.h_in + (size_t)y0 * rb, (size_t)(y1 - y0) * rb, hipMemcpyHostToDevice, ln.hs));
    }
    e->diag_stage_in_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st0).count();
  }
  uint32_t flags = 0;
  const bool saved_borrow = e->borrow;
  e->borrow = false;
  if (e->last_fg_stream == stream) e->last_fg_stream = -1;  // bgs_last_mask_blobs serves the synchronous call's mask
  rc = process_range(e, stream, 1, dst, ln.d_fg, bg ? ln.d_bg : nullptr, nullptr, ln.hs, &flags);
  e->borrow = saved_borrow;
  if (rc) return rc;
  ln.flags = flags, ln.fg = fg, ln.bg = bg, ln.fg_step = fg_step, ln.bg_step = bg_step;
  ln.fg_direct = fg && (flags & BGS_FG_VALID) && fg_step == (size_t)cols && host_pin(e, stream, 1, fg, e->n, ln.hs);
  ln.bg_direct = bg && (flags & BGS_BG_VALID) && bg_step == (size_t)cols * bg_ch && host_pin(e, stream, 2, bg, e->n * bg_ch, ln.hs);
  e->diag_d2h_bytes += (int64_t)(((fg && (flags & BGS_FG_VALID)) ? e->n : 0) + ((bg && (flags & BGS_BG_VALID)) ? e->n * bg_ch : 0));
  if (fg && (flags & BGS_FG_VALID)) HIP_TRY(hipMemcpyAsync(ln.fg_direct ? fg : ln.h_fg, ln.d_fg, e->n, hipMemcpyDeviceToHost, ln.hs));
  if (bg && (flags & BGS_BG_VALID)) HIP_TRY(hipMemcpyAsync(ln.bg_direct ? bg : ln.h_bg, ln.d_bg, e->n * bg_ch, hipMemcpyDeviceToHost, ln.hs));
  HIP_TRY(hipEventRecord(ln.done, ln.hs));
  ln.pending = true;
  return BGS_OK;
}

int bgs_host_arena(bgs_engine* e, void* ptr, size_t bytes, int on) {
  if (!e || !ptr || (on && !bytes)) return fail(BGS_ERR_INVALID, "bgs_host_arena: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  for (size_t i = 0; i < e->arenas.size(); ++i)
    if (e->arenas[i].first == (const uint8_t*)ptr) {
      if (on) return e->arenas[i].second == bytes ? BGS_OK : fail(BGS_ERR_INVALID, "bgs_host_arena: this address is registered with another size");
      (void)hipDeviceSynchronize();  // nothing may still be reading or writing it
      HIP_TRY(hipHostUnregister(ptr));
      e->diag_unreg_calls++;
      e->arenas.erase(e->arenas.begin() + (long)i);
      return BGS_OK;
    }
  if (!on) return fail(BGS_ERR_INVALID, "bgs_host_arena: no such arena");
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t er = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  e->diag_reg_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  e->diag_reg_calls++;
  if (er != hipSuccess) {
    (void)hipGetLastError();
    return fail(BGS_ERR_HIP, "hipHostRegister of a %zu-byte arena failed: %s", bytes, hipGetErrorString(er));
  }
  e->arenas.emplace_back((const uint8_t*)ptr, bytes);
  return BGS_OK;
}

int bgs_wait(bgs_engine* e, int stream, uint32_t* out_flags) {
  if (out_flags) *out_flags = 0;
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  return lane_wait(e, stream, out_flags);
}

int64_t bgs_get_state(bgs_engine* e, int stream, const char* plane, void* dst, size_t cap) {
  if (!e || !plane || !dst) return fail(BGS_ERR_INVALID, "NULL argument");
  if (!e->n) return fail(BGS_ERR_STATE, "no model yet");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(BGS_ERR_HIP, "device sync failed");
  const size_t n = e->n, P = n * e->S, off = n * stream;
  const int C = e->ch;
  auto copy_bytes = [&](const uint8_t* src, size_t nb) -> int64_t {
    if (cap < nb) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
    if (d2h_staged(dst, src, nb) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    return (int64_t)nb;
  };
  if (!strcmp(plane, "hostpath")) {  // diagnostics of bgs_process / bgs_submit since creation: 15 doubles (see bench.py host_path)
    double rec[15] = {0};
    for (size_t i = 0; i < e->pin.size(); ++i) rec[i % 3] += e->pin[i].pinned, rec[3 + i % 3] += e->pin[i].refused;  // per role: input, mask, background
    rec[6] = (double)e->diag_reg_calls, rec[7] = e->diag_reg_ms, rec[8] = (double)e->diag_unreg_calls, rec[9] = (double)e->diag_frames;
    rec[10] = (double)e->diag_h2d_bytes, rec[11] = (double)e->diag_d2h_bytes, rec[12] = e->diag_stage_in_ms, rec[13] = e->diag_stage_out_ms, rec[14] = (double)e->arenas.size();
    if (cap < sizeof(rec)) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
    memcpy(dst, rec, sizeof(rec));
    return (int64_t)sizeof(rec);
  }
  if (!strcmp(plane, "placement")) {  // diagnostics: [0] chunk size in MiB of the chunked model (0: one plain allocation), [1] number of chunks
    float rec[2] = {e->vmm.base ? (float)(e->vmm.chunk >> 20) : 0.f, (float)e->vmm.handles.size()};
    if (cap < sizeof(rec)) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
    memcpy(dst, rec, sizeof(rec));
    return (int64_t)sizeof(rec);
  }
  if (e->algo == BGS_MOG2) {
    // canonical export: "w" [K][n], "var" [K][n], "mu" [K][3][n] floats, "nmodes" [n] bytes — whatever the device layout
    int p0 = -1, np = 0;
    if (!strcmp(plane, "w")) p0 = 0, np = 5;
    if (!strcmp(plane, "var")) p0 = 5, np = 5;
    if (!strcmp(plane, "mu")) p0 = 10, np = 15;
    if (!strcmp(plane, "summary")) p0 = 100, np = 5;  // uint32 [K][n] by rank: the 16-bit word q0 | q1 << 5 | q2 << 10 | class << 15 (kernel_mog2.h; for the invariant test)
    const bool nm = !strcmp(plane, "nmodes") || !strcmp(plane, "summary_valid");  // bytes [n]; summary_valid: bit 15 of the meta word
    const bool want_valid = !strcmp(plane, "summary_valid");
    if (p0 >= 0 || nm) {
      // device layout (kernel_mog2.h): weights by rank, {var, mean} records in fixed slots, meta = rank -> slot.  Exported in the
      // reference's array order (rank); entries past a pixel's mode count are zero, as in the reference's zero-initialised bgmodel.
      const size_t need = nm ? n : (size_t)np * n * 4;
      if (cap < need) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
      const size_t T = bgs::kMog2Tile, TB = bgs::kMog2TileBytes;
      const size_t t0 = off / T, t1 = (off + n + T - 1) / T;
      std::vector<uint8_t> tiles((t1 - t0) * TB);
      if (d2h_staged(tiles.data(), e->mog2_state + t0 * TB, tiles.size()) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
      for (size_t i = 0; i < n; ++i) {
        const size_t sp = off + i, in = sp % T;
        const uint8_t* tb = tiles.data() + (sp / T - t0) * TB;
        const float* w = reinterpret_cast<const float*>(tb) + in;
        const float* rec = reinterpret_cast<const float*>(tb + bgs::kMog2RecOff) + in * 4;
        const uint16_t* sum = reinterpret_cast<const uint16_t*>(tb + bgs::kMog2SumOff) + in;
        const unsigned meta = reinterpret_cast<const uint16_t*>(tb + bgs::kMog2MetaOff)[in];
        if (nm) {
          ((uint8_t*)dst)[i] = want_valid ? (uint8_t)((meta >> 15) & 1u) : (uint8_t)bgs::mog2_meta_count(meta);
          continue;
        }
        for (int r = 0; r < bgs::kMog2K; ++r) {
          const unsigned f = (meta >> (3 * r)) & 7u;
          const float* rc = f ? rec + (size_t)(f - 1) * T * 4 : nullptr;
          if (p0 == 100) ((uint32_t*)dst)[(size_t)r * n + i] = f ? sum[(size_t)(f - 1) * T] : 0u;
          if (p0 == 0) ((float*)dst)[(size_t)r * n + i] = f ? w[(size_t)r * T] : 0.f;
          if (p0 == 5) ((float*)dst)[(size_t)r * n + i] = f ? rc[0] : 0.f;
          if (p0 == 10)
            for (int c = 0; c < 3; ++c) ((float*)dst)[((size_t)r * 3 + c) * n + i] = f ? rc[1 + c] : 0.f;
        }
      }
      return (int64_t)need;
    }
  }
  if ((e->algo == BGS_SUBSENSE || e->algo == BGS_LOBSTER) && e->ss) return ss_get_state(e, stream, plane, dst, cap);
  if (is_dp(e->algo)) {  // planes are stored canonically: [stream][plane][n]
    const int planes = dp_planes_of(e);
    const char* fname = (e->algo == BGS_DP_WREN_GA) ? "gauss" : (e->algo == BGS_DP_MEAN) ? "mean" : "modes";
    if (planes && !strcmp(plane, fname)) return dp_export_planes(e, stream, planes, dst, cap);
    if (e->state_ch == 1 && !strcmp(plane, "nmodes")) return copy_bytes(e->bgstate + off, n);
    if (e->state_ch == 3 && !strcmp(plane, "median")) return copy_bytes(e->bgstate + off * 3, n * 3);
    return fail(BGS_ERR_STATE, "unknown state plane '%s' for algorithm %d", plane, (int)e->algo);
  }
  if (e->algo == BGS_GMG && e->gmg_rec) {  // canonical: colors int32 [F][n], weights f32 [F][n] (entries past the count exported as 0), nfeatures int32 [n]
    const size_t F = (size_t)e->p.gmg_max_features;
    std::vector<uint8_t> nf(n);
    if (d2h_staged(nf.data(), e->gmg_nfeat + off, n) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
    if (!strcmp(plane, "nfeatures")) {
      if (cap < n * 4) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
      for (size_t i = 0; i < n; ++i) ((int32_t*)dst)[i] = nf[i];
      return (int64_t)(n * 4);
    }
    if (!strcmp(plane, "colors") || !strcmp(plane, "weights")) {
      if (cap < n * F * 4) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
      const int which = !strcmp(plane, "colors") ? 0 : 1;  // the device holds {colour, weight} records (kernel_gmg.h)
      std::vector<uint32_t> recs(n * 2);
      for (size_t f = 0; f < F; ++f) {
        if (d2h_staged(recs.data(), e->gmg_rec + f * P + off, n * 8) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
        for (size_t i = 0; i < n; ++i) ((uint32_t*)dst)[f * n + i] = f >= nf[i] ? 0u : recs[2 * i + which];
      }
      return (int64_t)(n * F * 4);
    }
  }
  if (e->algo == BGS_MOG1) {  // exported in the reference's order: [rank][channel][pixel] (kernel_mog1.h keeps records by slot)
    const int K = bgs::kMog1K;
    int kind = -1, nf = 0;
    if (!strcmp(plane, "sortkey")) kind = 0, nf = 1;
    if (!strcmp(plane, "w")) kind = 1, nf = 1;
    if (!strcmp(plane, "mu")) kind = 2, nf = C;
    if (!strcmp(plane, "var")) kind = 3, nf = C;
    if (kind >= 0) {
      const size_t need = (size_t)K * nf * n * 4;
      if (cap < need) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
      const size_t T = bgs::kMog1Tile, TF = C == 3 ? bgs::mog1_tile_floats<3>() : bgs::mog1_tile_floats<1>(), t0 = off / T, t1 = (off + n + T - 1) / T;
      std::vector<float> tiles((t1 - t0) * TF);
      if (d2h_staged(tiles.data(), e->mog1_state + t0 * TF, tiles.size() * 4) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
      for (size_t i = 0; i < n; ++i) {
        const size_t sp = off + i, l = sp % T;
        const float* tb = tiles.data() + (sp / T - t0) * TF;
        const unsigned meta = reinterpret_cast<const uint16_t*>(tb + 2 * K * T + K * T * 2 * C)[l];
        for (int k = 0; k < K; ++k) {
          const int slot = (int)((meta >> (3 * k)) & 7u) - 1;  // -1: this rank never held a mode (all zeros in the reference)
          for (int c = 0; c < nf; ++c) {
            float v;
            if (kind <= 1)
              v = tb[(size_t)(kind * K + k) * T + l];
            else
              v = slot < 0 ? 0.f : tb[2 * K * T + (size_t)slot * T * 2 * C + l * 2 * C + (kind == 3 ? C : 0) + c];
            ((float*)dst)[((size_t)k * nf + c) * n + i] = v;
          }
        }
      }
      return (int64_t)need;
    }
  }
  if (e->algo == BGS_SIGMA_DELTA && e->seen[stream] >= 1 && (!strcmp(plane, "mt") || !strcmp(plane, "vt")))
    return copy_bytes((!strcmp(plane, "mt") ? e->bgstate : e->bgstate2) + off * 3, n * 3);
  if (!strcmp(plane, "bg") && e->algo == BGS_ASBL) return copy_bytes((e->flip[stream] ? e->bgstate2 : e->bgstate) + off, n);
  if (!strcmp(plane, "bg") && e->bgstate) return copy_bytes(e->bgstate + off * e->state_ch, n * e->state_ch);
  const int64_t t = e->seen[stream];
  const int64_t rp = e->rpos[stream];
  if (!strcmp(plane, "prev1") && e->nring && t >= 1) return copy_bytes(e->ring[(rp + e->nring - 1) % e->nring] + off * C, n * C);
  if (!strcmp(plane, "prev2") && e->nring == 3 && t >= 2) return copy_bytes(e->ring[(rp + e->nring - 2) % e->nring] + off * C, n * C);
  return fail(BGS_ERR_STATE, "unknown state plane '%s' for algorithm %d", plane, (int)e->algo);
}

int64_t bgs_frames_seen(const bgs_engine* e, int stream) {
  if (!e || stream < 0 || stream >= e->S) return BGS_ERR_INVALID;
  return e->seen[stream];
}

// One camera starts over: what deleting its IBGS object and creating a new one does in the reference (FrameProcessor.cpp:342-482 then
// :35-155; ustc_src/ustc_bgs.cpp:75-77).  Nothing is launched here: the stream's frame count goes back to 0, so its next frame - on
// whatever HIP stream that call uses, in order with everything before it - re-initialises its model exactly like a first frame
// (mog2_clear, ss_init_streams, the warm-up of the history classes ...).  The other streams of the batch are not touched and keep
// sharing launches with each other; the reset stream re-joins them as soon as its launch arguments equal theirs again (launch_key).
int bgs_reset_stream(bgs_engine* e, int stream) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  e->seen[stream] = 0, e->counter[stream] = 0, e->last_flags[stream] = 0;
  if (e->last_fg_stream == stream) e->last_fg_stream = -1;
  return BGS_OK;
}

int bgs_stream_flags(const bgs_engine* e, int stream, uint32_t* out_flags) {
  if (!e || !out_flags) return fail(BGS_ERR_INVALID, "NULL argument");
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  *out_flags = e->last_flags[stream];
  return BGS_OK;
}

int bgs_enable_kernel_timing(bgs_engine* e, int on) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  e->timing = on != 0;
  for (auto& ev : e->events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  e->events.clear();
  return BGS_OK;
}

int bgs_kernel_timing(bgs_engine* e, double* avg_ms, int64_t* launches, const char** kernel_name) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  HIP_TRY(hipSetDevice(e->device));
  double total = 0;
  for (auto& ev : e->events) {
    HIP_TRY(hipEventSynchronize(ev.second));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
    total += ms;
  }
  if (avg_ms) *avg_ms = e->events.empty() ? 0.0 : total / (double)e->events.size();
  if (launches) *launches = (int64_t)e->events.size();
  if (kernel_name) *kernel_name = e->kernel_name;
  return BGS_OK;
}

int64_t bgs_kernel_timing_series(bgs_engine* e, float* ms, int64_t cap) {
  if (!e || !ms || cap < 0) return fail(BGS_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(e->device));
  int64_t n = 0;
  for (auto& ev : e->events) {
    if (n >= cap) break;
    HIP_TRY(hipEventSynchronize(ev.second));
    HIP_TRY(hipEventElapsedTime(&ms[n], ev.first, ev.second));
    ++n;
  }
  return n;
}

void bgs_destroy(bgs_engine* e) {
  if (!e) return;
  if (e->n || e->stream) {
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
  }
  free_all(e);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// ---- measurement aids (bench.py `calibration`): what THIS box's memory system and bus deliver, measured by the library that is being
// benchmarked, in the process that benchmarks it --------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace

int bgs_calibrate_copy(int hip_device, size_t bytes, int chunk_mb, int iters, double* gbps) {
  if (!gbps || bytes < (1u << 20) || iters < 1 || chunk_mb < 0) return fail(BGS_ERR_INVALID, "bgs_calibrate_copy: bad argument");
  *gbps = 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(BGS_ERR_HIP, "no HIP device visible: libbgs_hip has no CPU path");
  HIP_TRY(hipSetDevice(hip_device));
  const size_t half = bytes / 2 / 4096 * 4096, n = half / 16;
  VmmRange v;
  void* base = nullptr;
  if (chunk_mb > 0) {
    if (vmm_allocate(v, hip_device, &base, 2 * half, (size_t)chunk_mb << 20) != BGS_OK) {
      vmm_free(v);
      return BGS_ERR_HIP;  // text set by vmm_allocate
    }
  } else {
    HIP_TRY(hipMalloc(&base, 2 * half));
  }
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t er = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (er == hipSuccess) er = hipEventCreate(&a);
  if (er == hipSuccess) er = hipEventCreate(&b);
  if (er == hipSuccess) er = hipMemsetAsync(base, 1, 2 * half, s);
  float ms = 0;
  if (er == hipSuccess) {
    const dim3 grid((unsigned)((n + 255) / 256));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(calib_copy_kernel, grid, dim3(256), 0, s, (const float4*)base, (float4*)((char*)base + half), n);
    er = hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(calib_copy_kernel, grid, dim3(256), 0, s, (const float4*)base, (float4*)((char*)base + half), n);
    if (er == hipSuccess) er = hipEventRecord(b, s);
    if (er == hipSuccess) er = hipEventSynchronize(b);
    if (er == hipSuccess) er = hipEventElapsedTime(&ms, a, b);
    if (er == hipSuccess) er = hipGetLastError();
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (s) (void)hipStreamSynchronize(s), (void)hipStreamDestroy(s);
  if (chunk_mb > 0)
    vmm_free(v);
  else
    (void)hipFree(base);
  if (er != hipSuccess) return fail(BGS_ERR_HIP, "bgs_calibrate_copy: %s", hipGetErrorString(er));
  *gbps = ms > 0 ? 2.0 * (double)half * iters / (ms * 1e-3) / 1e9 : 0.0;  // half read + half written per launch
  return BGS_OK;
}

int bgs_calibrate_pcie(int hip_device, size_t bytes, int registered, int iters, double* h2d_gbps, double* d2h_gbps, double* register_ms) {
  if (bytes < 4096 || iters < 1) return fail(BGS_ERR_INVALID, "bgs_calibrate_pcie: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(BGS_ERR_HIP, "no HIP device visible: libbgs_hip has no CPU path");
  HIP_TRY(hipSetDevice(hip_device));
  void *h = nullptr, *d = nullptr;
  double reg = 0;
  if (registered) {  // the caller-buffer case of BGS_OPT_HOST_REGISTER: ordinary pageable memory, page-locked in place
    if (posix_memalign(&h, 4096, bytes)) return fail(BGS_ERR_NOMEM, "out of host memory");
    std::memset(h, 1, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t er = hipHostRegister(h, bytes, hipHostRegisterDefault);
    reg = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (er != hipSuccess) {
      free(h);
      return fail(BGS_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString(er));
    }
  } else {
    HIP_TRY(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    std::memset(h, 1, bytes);
  }
  hipStream_t s = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipError_t er = hipMalloc(&d, bytes);
  if (er == hipSuccess) er = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (auto& e1 : ev)
    if (er == hipSuccess) er = hipEventCreate(&e1);
  float up = 0, down = 0;
  if (er == hipSuccess) {
    (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);  // warm
    (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    (void)hipEventRecord(ev[0], s);
    for (int i = 0; i < iters; ++i) (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
    (void)hipEventRecord(ev[1], s);
    for (int i = 0; i < iters; ++i) (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    er = hipEventRecord(ev[2], s);
    if (er == hipSuccess) er = hipEventSynchronize(ev[2]);
    if (er == hipSuccess) er = hipEventElapsedTime(&up, ev[0], ev[1]);
    if (er == hipSuccess) er = hipEventElapsedTime(&down, ev[1], ev[2]);
  }
  for (auto& e1 : ev)
    if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamSynchronize(s), (void)hipStreamDestroy(s);
  if (d) (void)hipFree(d);
  if (registered)
    (void)hipHostUnregister(h), free(h);
  else
    (void)hipHostFree(h);
  if (er != hipSuccess) return fail(BGS_ERR_HIP, "bgs_calibrate_pcie: %s", hipGetErrorString(er));
  if (h2d_gbps) *h2d_gbps = up > 0 ? (double)bytes * iters / (up * 1e-3) / 1e9 : 0.0;
  if (d2h_gbps) *d2h_gbps = down > 0 ? (double)bytes * iters / (down * 1e-3) / 1e9 : 0.0;
  if (register_ms) *register_ms = reg;
  return BGS_OK;
}

int bgs_lbsp_describe_device(int hip_device, const void* d_img, int rows, int cols, int channels, const uint8_t* t_lut, void* d_desc, void* hip_stream) {
  return bgs_lbsp_describe_batch_device(hip_device, d_img, 1, rows, cols, channels, t_lut, d_desc, hip_stream);
}

int bgs_lbsp_describe_batch_device(int hip_device, const void* d_img, int images, int rows, int cols, int channels, const uint8_t* t_lut, void* d_desc,
                                   void* hip_stream) {
  if (!d_img || !d_desc || !t_lut || rows <= 0 || cols <= 0 || images <= 0 || images > 65535) return fail(BGS_ERR_INVALID, "bad argument");
  if (images > 1 && ((size_t)rows * cols * channels) % 4) return fail(BGS_ERR_INVALID, "batched images must each be a multiple of 4 bytes (4-byte aligned rows of dwords)");
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  if (!aligned(d_img, 4)) return fail(BGS_ERR_INVALID, "image must be 4-byte aligned");
  HIP_TRY(hipSetDevice(hip_device));
  bgs::LbspArgs a{};
  a.img = (const uint8_t*)d_img, a.desc = (uint16_t*)d_desc, a.rows = rows, a.cols = cols;
  std::memcpy(a.lut, t_lut, 256);
  const dim3 grid((cols + bgs::kLbspTW - 1) / bgs::kLbspTW, (rows + bgs::kLbspTH - 1) / bgs::kLbspTH, images);
  if (channels == 3)
    hipLaunchKernelGGL((bgs::lbsp_kernel<3>), grid, dim3(bgs::kBlock), 0, (hipStream_t)hip_stream, a);
  else
    hipLaunchKernelGGL((bgs::lbsp_kernel<1>), grid, dim3(bgs::kBlock), 0, (hipStream_t)hip_stream, a);
  HIP_TRY(hipGetLastError());
  return BGS_OK;
}

int bgs_mask_morph_device(int hip_device, const void* d_src, void* d_dst, int rows, int cols, int op, int ksize, int iterations, void* hip_stream) {
  if (!d_src || !d_dst || rows <= 0 || cols <= 0 || op < 0 || op > 4 || iterations < 1) return fail(BGS_ERR_INVALID, "bad argument");
  if ((op == 2 || op == 3) && (ksize < 3 || ksize > 2 * bgs::kMorphMaxR + 1 || ksize % 2 == 0)) return fail(BGS_ERR_INVALID, "median ksize must be odd, 3..15");
  if (d_src == d_dst) return fail(BGS_ERR_INVALID, "in-place morphology is not supported");
  HIP_TRY(hipSetDevice(hip_device));
  hipStream_t s = (hipStream_t)hip_stream;
  if (op == 4) {  // cv::floodFill(img, Point(0,0), 255)
    const int W64 = (cols + 63) / 64, tilesY = (rows + 63) / 64;
    uint64_t *mb = nullptr, *rb = nullptr;
    int* fl = nullptr;
    HIP_TRY(hipMalloc((void**)&mb, (size_t)rows * W64 * 8));
    HIP_TRY(hipMalloc((void**)&rb, (size_t)rows * W64 * 8));
    HIP_TRY(hipMalloc((void**)&fl, bgs::kSsFloodFlags * sizeof(int)));
    (void)hipMemsetAsync(fl, 0, bgs::kSsFloodFlags * sizeof(int), s);
    hipLaunchKernelGGL(bgs::ss_flood_pack_kernel, dim3(blocks_for((size_t)rows * W64 * bgs::kWave)), dim3(bgs::kBlock), 0, s, (const uint8_t*)d_src, mb, rb, rows, cols, W64);
    hipLaunchKernelGGL(bgs::ss_flood_seed_kernel, dim3(1), dim3(bgs::kBlock), 0, s, (const uint64_t*)mb, rb, rows, cols, W64);
    // the same host-free sequence as SuBSENSE's post-processing (engine_subsense.h): a fixed batch, then the finish kernel
    for (int k = 0; k < bgs::kSsFloodBatch; ++k) ss_launch_flood(dim3(blocks_for((size_t)tilesY * W64 * bgs::kWave)), 1, tilesY, s, mb, rb, rows, W64, fl, k);
    hipLaunchKernelGGL(bgs::ss_flood_finish_kernel, dim3(1), dim3(1024), 0, s, (const uint64_t*)mb, rb, rows, W64, fl, bgs::kSsFloodBatch);
    hipLaunchKernelGGL(bgs::ss_flood_paint_kernel, dim3(blocks_for((size_t)rows * cols)), dim3(bgs::kBlock), 0, s, (const uint8_t*)d_src, (const uint64_t*)rb, (uint8_t*)d_dst, rows, cols, W64);
    hipError_t er = hipStreamSynchronize(s);
    (void)hipFree(mb), (void)hipFree(rb), (void)hipFree(fl);
    if (er != hipSuccess) return fail(BGS_ERR_HIP, "flood fill failed: %s", hipGetErrorString(er));
    return BGS_OK;
  }
  const size_t n = (size_t)rows * cols;
  // n iterations of the 3x3 erode/dilate are one (2n+1)^2 box (kernel_stencil.h); a launch covers up to kMorphMaxR of them
  int passes = iterations, per_pass = 1;
  if (op <= 1) per_pass = bgs::kMorphMaxR, passes = (iterations + per_pass - 1) / per_pass;
  uint8_t* tmp = nullptr;
  if (passes > 1) HIP_TRY(hipMalloc((void**)&tmp, n));
  const uint8_t* in = (const uint8_t*)d_src;
  int left = iterations;
  for (int i = 0; i < passes; ++i) {
    // alternate so that the last pass lands in d_dst
    uint8_t* out = ((passes - 1 - i) % 2 == 0) ? (uint8_t*)d_dst : tmp;
    const int it = op <= 1 ? std::min(left, per_pass) : 1;
    bgs::MorphArgs a{in, out, rows, cols, op, op <= 1 ? 2 * it + 1 : ksize};
    bgs::morph_launch(a, 1, s);
    left -= it;
    in = out;
  }
  hipError_t er = hipGetLastError();
  if (tmp) {
    (void)hipStreamSynchronize(s);
    (void)hipFree(tmp);
  }
  if (er != hipSuccess) return fail(BGS_ERR_HIP, "morph kernel launch failed: %s", hipGetErrorString(er));
  return BGS_OK;
}

size_t bgs_mask_components_workspace(int rows, int cols) { return bgs_mask_components_batch_workspace(1, rows, cols); }

size_t bgs_mask_components_batch_workspace(int images, int rows, int cols) {
  if (images <= 0 || rows <= 0 || cols <= 0) return 0;
  const size_t n = (size_t)images * rows * cols, nb = (n + bgs::kCcPerBlock - 1) / bgs::kCcPerBlock;
  return (2 * n + nb + 16) * sizeof(int32_t);  // labels (when the caller passes none) + ids + per-block root counts
}

// images stacked back to back are labelled as one tall image whose links never cross an image boundary
static int cc_run(int hip_device, const void* d_mask, int images, int rows, int cols, int connectivity, int32_t* d_labels, bgs_box* d_boxes, int max_boxes,
                  int32_t* d_count, int32_t* d_offsets, void* d_work, void* hip_stream, bgs_moments* d_moments = nullptr) {
  static_assert(sizeof(bgs_moments) == sizeof(bgs::CcMoments), "bgs_moments layout");
  static_assert(sizeof(bgs_box) == sizeof(bgs::CcBox), "bgs_box layout");
  if (!d_mask || !d_boxes || (!d_count && !d_offsets) || images <= 0 || rows <= 0 || cols <= 0 || max_boxes < 0 || (connectivity != 4 && connectivity != 8))
    return fail(BGS_ERR_INVALID, "bgs_mask_components: bad argument");
  const size_t imgN = (size_t)rows * cols, n = imgN * images;
  if (n >= (size_t)0x7fffffff) return fail(BGS_ERR_UNSUPPORTED, "bgs_mask_components: too many pixels for 32-bit labels");
  HIP_TRY(hipSetDevice(hip_device));
  hipStream_t s = (hipStream_t)hip_stream;
  void* own = nullptr;
  if (!d_work) {
    HIP_TRY(hipMalloc(&own, bgs_mask_components_batch_workspace(images, rows, cols)));
    d_work = own;
  }
  const int nb = (int)((n + bgs::kCcPerBlock - 1) / bgs::kCcPerBlock);
  int* id = (int*)d_work;
  int* L = d_labels ? (int*)d_labels : id + n;
  int* blockCount = id + 2 * n;
  int* total = d_count ? (int*)d_count : blockCount + nb;  // scratch slot when only offsets are wanted
  const int conn8 = connectivity == 8, allRows = rows * images;
  const dim3 grid(blocks_for(n)), block(bgs::kBlock);
  if (d_offsets) HIP_TRY(hipMemsetAsync(d_offsets, 0, ((size_t)images + 1) * sizeof(int32_t), s));
  if (d_moments && max_boxes > 0) HIP_TRY(hipMemsetAsync(d_moments, 0, (size_t)max_boxes * sizeof(bgs_moments), s));
  hipLaunchKernelGGL(bgs::cc_init_kernel, grid, block, 0, s, (const uint8_t*)d_mask, L, allRows, cols, rows, conn8);
  hipLaunchKernelGGL(bgs::cc_merge_kernel, grid, block, 0, s, L, allRows, cols, rows, conn8);
  hipLaunchKernelGGL(bgs::cc_compress_kernel, grid, block, 0, s, L, n);
  hipLaunchKernelGGL(bgs::cc_count_kernel, dim3(nb), block, 0, s, (const int*)L, n, blockCount);
  hipLaunchKernelGGL(bgs::cc_scan_kernel, dim3(1), block, 0, s, blockCount, nb, total);
  hipLaunchKernelGGL(bgs::cc_scatter_kernel, dim3(nb), block, 0, s, (const int*)L, n, (const int*)blockCount, id, (bgs::CcBox*)d_boxes, max_boxes, (int*)d_offsets, imgN);
  if (max_boxes > 0) {
    hipLaunchKernelGGL(bgs::cc_boxes_kernel, dim3(blocks_for((n + bgs::kCcBoxPer - 1) / bgs::kCcBoxPer)), block, 0, s, (const int*)L, (const int*)id, allRows, cols, rows, (bgs::CcBox*)d_boxes, max_boxes, (bgs::CcMoments*)d_moments);
    hipLaunchKernelGGL(bgs::cc_finish_kernel, dim3(blocks_for((size_t)max_boxes)), block, 0, s, (bgs::CcBox*)d_boxes, (const int*)total, max_boxes, (int)imgN);
  }
  if (d_offsets) hipLaunchKernelGGL(bgs::cc_offsets_kernel, dim3(1), dim3(1), 0, s, (int*)d_offsets, images);
  if (d_labels && images > 1) hipLaunchKernelGGL(bgs::cc_localize_kernel, grid, block, 0, s, L, n, imgN);
  hipError_t er = hipGetLastError();
  if (own) {
    const hipError_t e2 = hipStreamSynchronize(s);
    if (er == hipSuccess) er = e2;
    (void)hipFree(own);
  }
  if (er != hipSuccess) return fail(BGS_ERR_HIP, "connected components failed: %s", hipGetErrorString(er));
  return BGS_OK;
}

int bgs_mask_components_device(int hip_device, const void* d_mask, int rows, int cols, int connectivity, int32_t* d_labels, bgs_box* d_boxes, int max_boxes,
                               int32_t* d_count, void* d_work, void* hip_stream) {
  if (!d_count) return fail(BGS_ERR_INVALID, "bgs_mask_components_device: d_count is NULL");
  return cc_run(hip_device, d_mask, 1, rows, cols, connectivity, d_labels, d_boxes, max_boxes, d_count, nullptr, d_work, hip_stream);
}

int bgs_mask_components_batch_device(int hip_device, const void* d_masks, int images, int rows, int cols, int connectivity, int32_t* d_labels, bgs_box* d_boxes,
                                     int max_boxes, int32_t* d_offsets, void* d_work, void* hip_stream) {
  if (!d_offsets) return fail(BGS_ERR_INVALID, "bgs_mask_components_batch_device: d_offsets is NULL");
  return cc_run(hip_device, d_masks, images, rows, cols, connectivity, d_labels, d_boxes, max_boxes, nullptr, d_offsets, d_work, hip_stream);
}

int bgs_mask_blobs_batch_device(int hip_device, const void* d_masks, int images, int rows, int cols, int connectivity, bgs_box* d_boxes, bgs_moments* d_moments,
                                int max_boxes, int32_t* d_offsets, void* d_work, void* hip_stream) {
  if (!d_offsets) return fail(BGS_ERR_INVALID, "bgs_mask_blobs_batch_device: d_offsets is NULL");
  return cc_run(hip_device, d_masks, images, rows, cols, connectivity, nullptr, d_boxes, max_boxes, nullptr, d_offsets, d_work, hip_stream, d_moments);
}

int bgs_last_mask_blobs(bgs_engine* e, int stream, int connectivity, int min_w, int min_h, bgs_box* boxes, bgs_moments* moments, int max_boxes, int32_t* count) {
  if (!e || !count || max_boxes < 0 || (max_boxes > 0 && !boxes)) return fail(BGS_ERR_INVALID, "bgs_last_mask_blobs: bad argument");
  *count = 0;
  if (stream < 0 || stream >= e->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, e->S - 1);
  if (!e->n || e->last_fg_stream != stream)
    return fail(BGS_ERR_STATE, "bgs_last_mask_blobs: the last bgs_process call left no valid mask of stream %d on the device", stream);
  HIP_TRY(hipSetDevice(e->device));
  // device scratch, grown on demand: [workspace][boxes cap][moments cap][count].  The component pass keeps the first `cap` components in
  // raster order and COUNTS all of them; the size filter below runs on the host, so every component has to be kept: when a mask
  // holds more than the scratch has room for (an unfiltered FrameDifference mask can carry thousands of one-pixel speckles before the
  // first real blob), the scratch is regrown to the true total and the pass repeated - otherwise late large blobs would be lost.
  const size_t ws = (bgs_mask_components_workspace(e->rows, e->cols) + 15) & ~(size_t)15;
  int cap = std::max(std::max(max_boxes, 1024), e->cc_cap);
  int32_t n = 0;
  bgs_box* d_boxes = nullptr;
  bgs_moments* d_mom = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    if (!e->cc_work || e->cc_cap < cap) {
      if (e->cc_work) (void)hipFree(e->cc_work), e->cc_work = nullptr, e->cc_cap = 0;
  if (e->pack_fg) (void)hipFree(e->pack_fg), e->pack_fg = nullptr, e->pack_fg_bytes = 0;
      HIP_TRY(hipMalloc(&e->cc_work, ws + (size_t)cap * (sizeof(bgs_box) + sizeof(bgs_moments)) + 16));
      e->cc_cap = cap;
    }
    d_boxes = (bgs_box*)((char*)e->cc_work + ws);
    d_mom = (bgs_moments*)(d_boxes + e->cc_cap);
    int32_t* d_count = (int32_t*)(d_mom + e->cc_cap);
    int rc = cc_run(e->device, e->d_fg, 1, e->rows, e->cols, connectivity, nullptr, d_boxes, e->cc_cap, d_count, nullptr, e->cc_work, e->stream, d_mom);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(&n, d_count, sizeof(n), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (n <= e->cc_cap) break;
    cap = n + n / 8 + 64;  // the true total is known now: one more pass with room for all of them
  }
  n = std::min(n, e->cc_cap);  // (cannot bind after the second pass: the mask has not changed)
  std::vector<bgs_box> hb((size_t)n);
  std::vector<bgs_moments> hm((size_t)n);
  if (n) {
    { const int rc__ = d2h_staged(hb.data(), d_boxes, (size_t)n * sizeof(bgs_box)); if (rc__ != BGS_OK) return rc__; }
    { const int rc__ = d2h_staged(hm.data(), d_mom, (size_t)n * sizeof(bgs_moments)); if (rc__ != BGS_OK) return rc__; }
  }
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (hb[i].w < min_w || hb[i].h < min_h) continue;
    if (kept < max_boxes) {
      boxes[kept] = hb[i];
      if (moments) moments[kept] = hm[i];
    }
    ++kept;
  }
  *count = kept;
  return BGS_OK;
}

// ------------------------------------------------------------------------------------------------------------- N3: ingest

int bgs_set_ingest(bgs_engine* e, const bgs_ingest* c) {
  if (!e) return fail(BGS_ERR_INVALID, "engine is NULL");
  if (e->n) return fail(BGS_ERR_INVALID, "bgs_set_ingest must come before the first frame (it decides the engine's geometry)");
  if (!c) {
    e->ingest_on = false;
    return BGS_OK;
  }
  if (c->struct_size != sizeof(bgs_ingest)) return fail(BGS_ERR_INVALID, "bgs_ingest.struct_size mismatch");
  if (c->resize_percent <= 0) return fail(BGS_ERR_INVALID, "resize_percent must be positive");
  e->ingest = *c, e->ingest_on = true;
  return BGS_OK;
}

int bgs_ingest_default(bgs_ingest* c) {
  if (!c) return fail(BGS_ERR_INVALID, "cfg is NULL");
  std::memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(*c);
  c->resize_percent = 100;  // config/VideoCapture.xml, config/PreProcessor.xml: everything else off
  return BGS_OK;
}

int bgs_ingest_size(const bgs_ingest* c, int src_rows, int src_cols, int* rows, int* cols) {
  IngestPlan pl;
  int rc = ingest_plan(c, src_rows, src_cols, &pl);
  if (rc) return rc;
  if (rows) *rows = pl.rows;
  if (cols) *cols = pl.cols;
  return BGS_OK;
}

size_t bgs_ingest_workspace(const bgs_ingest* c, int images, int src_rows, int src_cols, int channels) {
  IngestPlan pl;
  if (images <= 0 || ingest_plan(c, src_rows, src_cols, &pl)) return 0;
  size_t need = 0;
  if (c->equalize_hist) need += (size_t)images * 256 * (sizeof(unsigned) + 1);
  if (c->gaussian_blur) need += (size_t)images * pl.rows * pl.cols * channels;  // the frame before the blur
  return need ? need + 64 : 0;
}

int bgs_ingest_device(int hip_device, const bgs_ingest* c, const void* d_src, int images, int src_rows, int src_cols, int channels, size_t src_step, void* d_dst,
                      void* d_work, void* hip_stream) {
  IngestPlan pl;
  int rc = ingest_plan(c, src_rows, src_cols, &pl);
  if (rc) return rc;
  if (!d_src || !d_dst || images <= 0 || images > 65535) return fail(BGS_ERR_INVALID, "bgs_ingest_device: bad argument");
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  if (src_step < (size_t)src_cols * channels) return fail(BGS_ERR_INVALID, "src_step %zu < src_cols*channels", src_step);
  if (c->equalize_hist && channels != 1) return fail(BGS_ERR_UNSUPPORTED, "equalizeHist needs a 1-channel frame (cv::equalizeHist asserts CV_8UC1, PreProcessor.cpp:64)");
  const size_t ws = bgs_ingest_workspace(c, images, src_rows, src_cols, channels);
  if (ws && !d_work) return fail(BGS_ERR_INVALID, "bgs_ingest_device: this configuration needs %zu bytes of d_work", ws);
  HIP_TRY(hipSetDevice(hip_device));
  hipStream_t s = (hipStream_t)hip_stream;
  const size_t n = (size_t)pl.rows * pl.cols;
  // workspace: [hist][lut][pad to 16][pre-blur frames]
  unsigned* hist = (unsigned*)d_work;
  uint8_t* lut = (uint8_t*)d_work + (c->equalize_hist ? (size_t)images * 256 * sizeof(unsigned) : 0);
  uint8_t* pre = (uint8_t*)(((uintptr_t)(lut + (c->equalize_hist ? (size_t)images * 256 : 0)) + 15) & ~(uintptr_t)15);
  uint8_t* geom_out = c->gaussian_blur ? pre : (uint8_t*)d_dst;
  bgs::IngestArgs a{};
  a.src = (const uint8_t*)d_src, a.dst = geom_out, a.src_rows = src_rows, a.src_cols = src_cols, a.rw = pl.rw, a.rh = pl.rh, a.rows = pl.rows, a.cols = pl.cols;
  a.x0 = pl.x0, a.y0 = pl.y0, a.flip = c->flip != 0, a.mode = pl.mode, a.src_step = src_step, a.scale_x = pl.scale_x, a.scale_y = pl.scale_y;
  const dim3 grid(blocks_for(n), 1, images), block(bgs::kBlock);
  if (channels == 3)
    hipLaunchKernelGGL((bgs::ingest_geom_kernel<3>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((bgs::ingest_geom_kernel<1>), grid, block, 0, s, a);
  if (c->equalize_hist) {
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)images * 256 * sizeof(unsigned), s));
    hipLaunchKernelGGL(bgs::ingest_hist_kernel, dim3(std::min<unsigned>(blocks_for(n), 1024), 1, images), block, 0, s, (const uint8_t*)geom_out, n, hist);
    hipLaunchKernelGGL(bgs::ingest_lut_kernel, dim3(images), block, 0, s, (const unsigned*)hist, lut, (unsigned)n);
    hipLaunchKernelGGL(bgs::ingest_apply_lut_kernel, grid, block, 0, s, geom_out, n, (const uint8_t*)lut);
  }
  if (c->gaussian_blur) {
    bgs::BlurArgs b{};
    b.src = pre, b.dst = (uint8_t*)d_dst, b.rows = pl.rows, b.cols = pl.cols;
    gaussian7_kernel(b.k);
    const dim3 bgrid((pl.cols + bgs::kBlurTW - 1) / bgs::kBlurTW, (pl.rows + bgs::kBlurTH - 1) / bgs::kBlurTH, images);
    if (channels == 3)
      hipLaunchKernelGGL((bgs::ingest_blur7_kernel<3>), bgrid, block, 0, s, b);
    else
      hipLaunchKernelGGL((bgs::ingest_blur7_kernel<1>), bgrid, block, 0, s, b);
  }
  HIP_TRY(hipGetLastError());
  return BGS_OK;
}

int bgs_ingest_host(int hip_device, const bgs_ingest* c, const uint8_t* src, int src_rows, int src_cols, int channels, size_t src_step, uint8_t* dst, size_t dst_step) {
  IngestPlan pl;
  int rc = ingest_plan(c, src_rows, src_cols, &pl);
  if (rc) return rc;
  if (!src || !dst) return fail(BGS_ERR_INVALID, "bgs_ingest_host: NULL buffer");
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  const size_t out_row = (size_t)pl.cols * channels;
  if (dst_step < out_row) return fail(BGS_ERR_INVALID, "dst_step %zu < cols*channels", dst_step);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(BGS_ERR_HIP, "no HIP device visible: libbgs_hip has no CPU path");
  HIP_TRY(hipSetDevice(hip_device));
  const size_t in_bytes = (size_t)src_rows * src_cols * channels, out_bytes = (size_t)pl.rows * out_row, ws = bgs_ingest_workspace(c, 1, src_rows, src_cols, channels);
  uint8_t *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr, *h_in = nullptr, *h_out = nullptr;
  hipStream_t s = nullptr;
  auto cleanup = [&]() {
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_ws) (void)hipFree(d_ws);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
    if (s) (void)hipStreamDestroy(s);
  };
  hipError_t er = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (er == hipSuccess) er = hipMalloc((void**)&d_in, in_bytes);
  if (er == hipSuccess) er = hipMalloc((void**)&d_out, out_bytes);
  if (er == hipSuccess && ws) er = hipMalloc((void**)&d_ws, ws);
  // the caller's images never go to hipMemcpy themselves (d2h_staged above: the runtime's by-address cache of pinned caller
  // buffers): rows are packed into / unpacked from page-locked images of this call's own (the caller's step may exceed the row)
  if (er == hipSuccess) er = hipHostMalloc((void**)&h_in, in_bytes, hipHostMallocDefault);
  if (er == hipSuccess) er = hipHostMalloc((void**)&h_out, out_bytes, hipHostMallocDefault);
  if (er == hipSuccess) {
    const size_t in_row = (size_t)src_cols * channels;
    for (int y = 0; y < src_rows; ++y) std::memcpy(h_in + (size_t)y * in_row, src + (size_t)y * src_step, in_row);
    er = hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s);
  }
  if (er != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return fail(BGS_ERR_HIP, "bgs_ingest_host: %s", hipGetErrorString(er));
  }
  rc = bgs_ingest_device(hip_device, c, d_in, 1, src_rows, src_cols, channels, (size_t)src_cols * channels, d_out, d_ws, s);
  if (!rc) {
    er = hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess)
      rc = fail(BGS_ERR_HIP, "bgs_ingest_host: %s", hipGetErrorString(er));
    else
      for (int y = 0; y < pl.rows; ++y) std::memcpy(dst + (size_t)y * dst_step, h_out + (size_t)y * out_row, out_row);
  } else {
    (void)hipStreamSynchronize(s);
  }
  cleanup();
  return rc;
}


}  // extern "C"

#include "engine_group.h"
