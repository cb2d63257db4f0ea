// engine_group.h — bgs_group: several classes on the same frames (included by bgs_hip.hip).
//
// FrameProcessor::process hands ONE pre-processed frame to every enabled IBGS, one call after the other
// (FrameProcessor.cpp:169-340); BASELINE configs[2] is two of them on the same 3840x2160 frames.  A group runs the byte-stream
// classes among them (FrameDifference, StaticFrameDifference, WeightedMovingMean, WeightedMovingVariance,
// AdaptiveBackgroundLearning, SigmaDelta - at most one instance of each) as ONE kernel over one read of the frame and one
// shared history ring (kernel_fanout.h); every other class (and a second instance of a fused one) is a member engine fed the
// same device frame.  Outputs, warm-up conventions and states are those of the separate engines, bit for bit.
// The streams of a group advance in lock-step (whole-batch calls); the host entry point serves single-stream groups, which is
// what FrameProcessor is.
#pragma once
#include "kernel_fanout.h"

struct bgs_group {
  int device = 0, S = 1, rows = 0, cols = 0, ch = 0;
  size_t n = 0;
  std::vector<bgs_algo> algos;
  std::vector<bgs_params> params;
  std::vector<unsigned> fan_bit;        // per class: its kFan* bit, 0 = member engine
  std::vector<bgs_engine*> member;      // per class: own engine, or null when fused
  unsigned fused = 0;                   // union of the fan bits
  uint8_t* ring[3] = {nullptr, nullptr, nullptr};
  int nring = 0;
  bool borrow = false;                  // BGS_OPT_BORROW_FRAMES: the caller's previous d_frames buffers are the history
  const uint8_t* borrowed[2] = {nullptr, nullptr};
  uint8_t *sfd_bg = nullptr, *abl_state = nullptr, *abl_lut = nullptr, *sd_mt = nullptr, *sd_vt = nullptr;
  double abl_lut_alpha = 0;
  bool abl_lut_valid = false;
  int64_t seen = 0, abl_counter = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  // host staging (single-stream groups)
  uint8_t *h_in = nullptr, *d_in = nullptr;
  bool staged = false;  // bgs_group_process: the whole staging set exists
  std::vector<uint8_t*> h_fg, h_bg, d_fg, d_bg;
  // kernel timing of the fused launch
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

namespace {

unsigned fan_bit_of(bgs_algo a) {
  switch (a) {
    case BGS_FRAME_DIFF: return bgs::kFanFD;
    case BGS_STATIC_FRAME_DIFF: return bgs::kFanSFD;
    case BGS_WMM: return bgs::kFanWMM;
    case BGS_WMV: return bgs::kFanWMV;
    case BGS_ABL: return bgs::kFanABL;
    case BGS_SIGMA_DELTA: return bgs::kFanSD;
    default: return 0;
  }
}

void group_free_staging(bgs_group* g) {  // bgs_group_process's host staging, whatever of it came to be
  if (g->d_in) (void)hipFree(g->d_in), g->d_in = nullptr;
  if (g->h_in) (void)hipHostFree(g->h_in), g->h_in = nullptr;
  for (auto* v : {&g->h_fg, &g->h_bg})
    for (uint8_t*& p : *v)
      if (p) (void)hipHostFree(p), p = nullptr;
  for (auto* v : {&g->d_fg, &g->d_bg})
    for (uint8_t*& p : *v)
      if (p) (void)hipFree(p), p = nullptr;
  g->staged = false;
}

void group_free(bgs_group* g) {
  void* dev[] = {g->ring[0], g->ring[1], g->ring[2], g->sfd_bg, g->abl_state, g->abl_lut, g->sd_mt, g->sd_vt};
  for (void* d : dev)
    if (d) (void)hipFree(d);
  g->ring[0] = g->ring[1] = g->ring[2] = nullptr, g->sfd_bg = g->abl_state = g->abl_lut = g->sd_mt = g->sd_vt = nullptr;
  group_free_staging(g);
  for (auto& ev : g->events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  g->events.clear();
  g->abl_lut_valid = false;
}

int group_index_of(const bgs_group* g, unsigned bit) {
  for (size_t i = 0; i < g->fan_bit.size(); ++i)
    if (g->fan_bit[i] == bit) return (int)i;
  return -1;
}

int group_build_lut(bgs_group* g) {
  const int i = group_index_of(g, bgs::kFanABL);
  if (i < 0) return BGS_OK;
  const double alpha = g->params[i].alpha;
  if (g->abl_lut_valid && alpha == g->abl_lut_alpha) return BGS_OK;
  // a launch still in flight on some stream may be reading the table: let the device drain before it is rewritten
  if (g->abl_lut_valid) HIP_TRY(hipDeviceSynchronize());
  if (!g->abl_lut) HIP_TRY(hipMalloc((void**)&g->abl_lut, 256 * 256));
  hipLaunchKernelGGL(bgs::abl_lut_kernel, dim3(256), dim3(bgs::kBlock), 0, g->stream, g->abl_lut, alpha, 1 - alpha);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g->stream));
  g->abl_lut_alpha = alpha, g->abl_lut_valid = true;
  return BGS_OK;
}

int group_allocate(bgs_group* g, int rows, int cols, int ch) {
  if (rows <= 0 || cols <= 0) return fail(BGS_ERR_INVALID, "bad geometry %dx%d", rows, cols);
  if (ch != 1 && ch != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3, got %d", ch);
  if ((g->fused & bgs::kFanSD) && ch != 3) return fail(BGS_ERR_UNSUPPORTED, "SigmaDeltaBGS is 3-channel only (sdLaMa091AllocInit_8u_C3R, SigmaDeltaBGS.cpp:35)");
  HIP_TRY(hipSetDevice(g->device));
  g->rows = rows, g->cols = cols, g->ch = ch, g->n = (size_t)rows * cols;
  const size_t fb = g->n * g->S * ch;
  if (!g->stream) HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  g->nring = (g->fused & (bgs::kFanWMM | bgs::kFanWMV)) ? 3 : (g->fused & bgs::kFanFD) ? 2 : 0;
  for (int i = 0; i < g->nring; ++i) HIP_TRY(hipMalloc((void**)&g->ring[i], fb));
  if (g->fused & bgs::kFanSFD) HIP_TRY(hipMalloc((void**)&g->sfd_bg, fb));
  if (g->fused & bgs::kFanABL) HIP_TRY(hipMalloc((void**)&g->abl_state, fb));
  if (g->fused & bgs::kFanSD) {
    HIP_TRY(hipMalloc((void**)&g->sd_mt, fb));
    HIP_TRY(hipMalloc((void**)&g->sd_vt, fb));
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0) g->n_cu = prop.multiProcessorCount;
  int rc = group_build_lut(g);
  if (rc) return rc;
  for (size_t i = 0; i < g->member.size(); ++i)
    if (g->member[i]) {
      rc = bgs_set_geometry(g->member[i], rows, cols, ch);
      if (rc) return rc;
    }
  return BGS_OK;
}

template <int G, int C>
void group_launch_fan(bgs_group* g, const bgs::FanArgs& a, hipStream_t s) {
  const bool lut = (a.mask & bgs::kFanABL) && a.abl_update;
  if (lut) {
    static std::atomic<int> per_cu_cache{0};  // a property of the code object; atomic because groups may be driven from several host threads
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (!per_cu) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bgs::fan_kernel<G, C, true>, bgs::kAblBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
      per_cu_cache.store(per_cu, std::memory_order_relaxed);
    }
    const size_t per_tile = (size_t)bgs::kAblBlock * G, ntiles = (a.npix + per_tile - 1) / per_tile;
    const dim3 grid((unsigned)std::min<size_t>(ntiles, (size_t)per_cu * g->n_cu));
    hipLaunchKernelGGL((bgs::fan_kernel<G, C, true>), grid, dim3(bgs::kAblBlock), 0, s, a, (const uint8_t*)g->abl_lut);
  } else {
    hipLaunchKernelGGL((bgs::fan_kernel<G, C, false>), dim3(blocks_for((a.npix + G - 1) / G)), dim3(bgs::kBlock), 0, s, a, (const uint8_t*)nullptr);
  }
}

// One frame of every stream for every class; device pointers, asynchronous on s.
int group_process_device(bgs_group* g, const uint8_t* d_frames, uint8_t* const* d_fg, uint8_t* const* d_bg, hipStream_t s, uint32_t* out_flags) {
  const int K = (int)g->algos.size();
  if (out_flags)
    for (int i = 0; i < K; ++i) out_flags[i] = 0;
  if (!g->n) return fail(BGS_ERR_INVALID, "geometry not set: call bgs_group_set_geometry or bgs_group_process first");
  if (!d_frames) return fail(BGS_ERR_INVALID, "d_frames is NULL");
  HIP_TRY(hipSetDevice(g->device));
  const int C = g->ch;
  const size_t npix = g->n * g->S, fb = npix * C;
  const int64_t t = g->seen;
  std::vector<uint32_t> flags((size_t)K, 0);
  if (g->fused) {
    int rc = group_build_lut(g);  // alpha may have changed through bgs_group_set_params
    if (rc) return rc;
    bgs::FanArgs a{};
    a.npix = npix;
    const uint8_t *cur = d_frames, *h1 = nullptr, *h2 = nullptr;
    if (g->nring) {
      const int R = g->nring;
      if (g->borrow) {
        h1 = g->borrowed[0], h2 = g->borrowed[1];
        g->borrowed[1] = g->borrowed[0], g->borrowed[0] = d_frames;
      } else {
        uint8_t* slot = g->ring[t % R];
        HIP_TRY(hipMemcpyAsync(slot, cur, fb, hipMemcpyDeviceToDevice, s));  // a private copy as history
        cur = slot;
        if (t >= 1) h1 = g->ring[(t - 1) % R];
        if (t >= 2 && R == 3) h2 = g->ring[(t - 2) % R];
      }
    }
    a.cur = cur, a.p1 = h1, a.p2 = h2;
    auto out_of = [&](unsigned bit, bgs::FanOut& o, bool with_bg) -> int {
      const int i = group_index_of(g, bit);
      const bgs_params& p = g->params[i];
      o.fg = d_fg ? d_fg[i] : nullptr, o.bg = (with_bg && d_bg) ? d_bg[i] : nullptr, o.bits = nullptr;
      o.thr = p.threshold, o.enable_thr = p.enable_threshold, o.enable_weight = p.enable_weight;
      return i;
    };
    if ((g->fused & bgs::kFanFD) && t >= 1) {
      const int i = out_of(bgs::kFanFD, a.fd, false);
      a.mask |= bgs::kFanFD, flags[i] = BGS_FG_VALID;
    }
    if (g->fused & bgs::kFanSFD) {
      const int i = out_of(bgs::kFanSFD, a.sfd, false);
      if (t == 0) HIP_TRY(hipMemcpyAsync(g->sfd_bg, d_frames, fb, hipMemcpyDeviceToDevice, s));  // img_input.copyTo(img_background)
      a.sfd_bg = g->sfd_bg, a.mask |= bgs::kFanSFD, flags[i] = BGS_FG_VALID | BGS_BG_VALID;
      if (d_bg && d_bg[i]) HIP_TRY(hipMemcpyAsync(d_bg[i], g->sfd_bg, fb, hipMemcpyDeviceToDevice, s));
    }
    if ((g->fused & bgs::kFanWMM) && t >= 2) {
      const int i = out_of(bgs::kFanWMM, a.wmm, true);
      a.mask |= bgs::kFanWMM, flags[i] = BGS_FG_VALID | BGS_BG_VALID;
    }
    if ((g->fused & bgs::kFanWMV) && t >= 2) {
      const int i = out_of(bgs::kFanWMV, a.wmv, false);
      a.mask |= bgs::kFanWMV, flags[i] = BGS_FG_VALID;
    }
    if (g->fused & bgs::kFanABL) {
      const int i = out_of(bgs::kFanABL, a.abl, true);
      const bgs_params& p = g->params[i];
      if (t == 0) HIP_TRY(hipMemcpyAsync(g->abl_state, d_frames, fb, hipMemcpyDeviceToDevice, s));
      a.abl_state = g->abl_state;
      a.abl_update = ((p.limit > 0 && p.limit < g->abl_counter) || p.limit == -1) ? 1 : 0;
      if (a.abl_update && p.limit > 0 && p.limit < g->abl_counter) g->abl_counter++;
      a.mask |= bgs::kFanABL, flags[i] = BGS_FG_VALID | BGS_BG_VALID;
    }
    if (g->fused & bgs::kFanSD) {
      const int i = out_of(bgs::kFanSD, a.sd, false);
      const bgs_params& p = g->params[i];
      if (t == 0) {  // SigmaDeltaBGS.cpp:33-39: allocate + initialise, return without output
        HIP_TRY(hipMemcpyAsync(g->sd_mt, d_frames, fb, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(bgs::sigmadelta_init_vt_kernel, dim3(blocks_for(fb)), dim3(bgs::kBlock), 0, s, g->sd_vt, fb, g->cols, (int)(uint8_t)p.sd_min_var);
      } else {
        a.sd_mt = g->sd_mt, a.sd_vt = g->sd_vt, a.sd_N = (uint32_t)p.sd_amp_factor, a.sd_vmin = (uint8_t)p.sd_min_var, a.sd_vmax = (uint8_t)p.sd_max_var;
        a.mask |= bgs::kFanSD, flags[i] = BGS_FG_VALID;
      }
    }
    if (a.mask) {
      const void* ptrs[] = {a.cur, a.p1, a.p2, a.fd.fg, a.sfd.fg, a.wmm.fg, a.wmm.bg, a.wmv.fg, a.abl.fg, a.abl.bg, a.sd.fg};
      int G = npix % 4 ? 1 : 4;
      for (const void* q : ptrs)
        if (q && !aligned(q, 4)) G = 1;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (g->timing && g->events.size() < 16384 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, s);
      if (C == 3) {
        if (G == 4) group_launch_fan<4, 3>(g, a, s);
        else group_launch_fan<1, 3>(g, a, s);
      } else {
        if (G == 4) group_launch_fan<4, 1>(g, a, s);
        else group_launch_fan<1, 1>(g, a, s);
      }
      if (e0 && e1) {
        (void)hipEventRecord(e1, s);
        g->events.emplace_back(e0, e1);
      }
      HIP_TRY(hipGetLastError());
    }
  }
  for (int i = 0; i < K; ++i)
    if (g->member[i]) {
      int rc = process_range(g->member[i], 0, g->S, d_frames, d_fg ? d_fg[i] : nullptr, d_bg ? d_bg[i] : nullptr, nullptr, s, &flags[i]);
      if (rc) return rc;
    }
  g->seen++;
  if (out_flags)
    for (int i = 0; i < K; ++i) out_flags[i] = flags[i];
  return BGS_OK;
}

}  // namespace

extern "C" {

int bgs_group_create(const bgs_algo* algos, const bgs_params* const* params, int n_algos, int hip_device, int n_streams, bgs_group** out) {
  if (!algos || !out || n_algos < 1 || n_algos > 64) return fail(BGS_ERR_INVALID, "bgs_group_create: bad argument");
  if (n_streams < 1) return fail(BGS_ERR_INVALID, "n_streams must be >= 1");
  *out = nullptr;
  bgs_group* g = new bgs_group();
  g->device = hip_device, g->S = n_streams;
  for (int i = 0; i < n_algos; ++i) {
    bgs_params p;
    std::memset(&p, 0, sizeof(p));
    p.struct_size = sizeof(p);
    int rc = bgs_default_params(algos[i], &p);
    if (!rc && params && params[i]) {
      if (params[i]->struct_size != sizeof(bgs_params)) rc = fail(BGS_ERR_INVALID, "bgs_params.struct_size mismatch (class %d)", i);
      else p = *params[i];
    }
    unsigned bit = fan_bit_of(algos[i]);
    if (bit & g->fused) bit = 0;  // a second instance of a fused class runs as a member engine
    bgs_engine* m = nullptr;
    if (!rc && !bit) rc = bgs_create(algos[i], &p, hip_device, n_streams, &m);
    if (rc) {
      for (bgs_engine* e : g->member)
        if (e) bgs_destroy(e);
      delete g;
      return rc;
    }
    g->algos.push_back(algos[i]), g->params.push_back(p), g->fan_bit.push_back(bit), g->member.push_back(m);
    g->fused |= bit;
  }
  if (g->fused) {  // no CPU path, like bgs_create
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || hip_device < 0 || hip_device >= ndev) {
      delete g;
      return fail(BGS_ERR_HIP, "no HIP device visible: libbgs_hip has no CPU path");
    }
  }
  g->h_fg.assign(n_algos, nullptr), g->h_bg.assign(n_algos, nullptr), g->d_fg.assign(n_algos, nullptr), g->d_bg.assign(n_algos, nullptr);
  *out = g;
  return BGS_OK;
}

void bgs_group_destroy(bgs_group* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (bgs_engine* e : g->member)
    if (e) bgs_destroy(e);
  group_free(g);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

int bgs_group_size(const bgs_group* g) { return g ? (int)g->algos.size() : BGS_ERR_INVALID; }

/* 1 when class `index` runs inside the fused kernel, 0 when it is a member engine */
int bgs_group_is_fused(const bgs_group* g, int index) {
  if (!g || index < 0 || index >= (int)g->algos.size()) return BGS_ERR_INVALID;
  return g->fan_bit[index] != 0;
}

int bgs_group_set_params(bgs_group* g, int index, const bgs_params* params) {
  if (!g || !params) return fail(BGS_ERR_INVALID, "NULL argument");
  if (index < 0 || index >= (int)g->algos.size()) return fail(BGS_ERR_INVALID, "class %d outside 0..%d", index, (int)g->algos.size() - 1);
  if (params->struct_size != sizeof(bgs_params)) return fail(BGS_ERR_INVALID, "bgs_params.struct_size mismatch");
  if (g->member[index]) return bgs_set_params(g->member[index], params);
  g->params[index] = *params;
  return BGS_OK;
}

int bgs_group_set_option(bgs_group* g, int option, int64_t value) {
  if (!g) return fail(BGS_ERR_INVALID, "group is NULL");
  if (option == BGS_OPT_BORROW_FRAMES) {
    g->borrow = value != 0;
    for (bgs_engine* e : g->member)
      if (e && (e->algo == BGS_FRAME_DIFF || e->algo == BGS_WMM || e->algo == BGS_WMV)) (void)bgs_set_option(e, option, value);
    return BGS_OK;
  }
  for (bgs_engine* e : g->member)
    if (e) (void)bgs_set_option(e, option, value);
  return BGS_OK;
}

int bgs_group_set_geometry(bgs_group* g, int rows, int cols, int channels) {
  if (!g) return fail(BGS_ERR_INVALID, "group is NULL");
  if (g->n) {
    if (rows == g->rows && cols == g->cols && channels == g->ch) return BGS_OK;
    return fail(BGS_ERR_GEOMETRY, "a group keeps its geometry (%dx%dx%d); create a new one for %dx%dx%d", g->rows, g->cols, g->ch, rows, cols, channels);
  }
  int rc = group_allocate(g, rows, cols, channels);
  if (rc) group_free(g), g->n = 0;
  return rc;
}

int bgs_group_process_batch_device(bgs_group* g, const void* d_frames, void* const* d_fg, void* const* d_bg, void* hip_stream, uint32_t* out_flags) {
  if (!g) return fail(BGS_ERR_INVALID, "group is NULL");
  return group_process_device(g, (const uint8_t*)d_frames, (uint8_t* const*)d_fg, (uint8_t* const*)d_bg, (hipStream_t)hip_stream, out_flags);
}

int bgs_group_process(bgs_group* g, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* const* fg, const size_t* fg_step,
                      uint8_t* const* bg, const size_t* bg_step, uint32_t* out_flags) {
  if (!g) return fail(BGS_ERR_INVALID, "group is NULL");
  const int K = (int)g->algos.size();
  if (out_flags)
    for (int i = 0; i < K; ++i) out_flags[i] = 0;
  if (g->S != 1) return fail(BGS_ERR_INVALID, "bgs_group_process serves single-stream groups (this one has %d streams): use bgs_group_process_batch_device", g->S);
  if (!in || rows <= 0 || cols <= 0) return BGS_OK;  // if(img_input.empty()) return;
  if (channels != 1 && channels != 3) return fail(BGS_ERR_UNSUPPORTED, "channels must be 1 or 3");
  if (in_step < (size_t)cols * channels) return fail(BGS_ERR_INVALID, "in_step %zu < cols*channels", in_step);
  int rc = bgs_group_set_geometry(g, rows, cols, channels);
  if (rc) return rc;
  const size_t fb = g->n * channels, row = (size_t)cols * channels;
  if (!g->staged) {
    // all or nothing: a staging set with some class's buffers missing would silently drop that class's outputs
    hipError_t er = hipHostMalloc((void**)&g->h_in, fb, hipHostMallocDefault);
    if (er == hipSuccess) er = hipMalloc((void**)&g->d_in, fb);
    for (int i = 0; i < K && er == hipSuccess; ++i) {
      const size_t bgb = g->n * (g->algos[i] == BGS_ASBL ? 1 : channels);
      er = hipHostMalloc((void**)&g->h_fg[i], g->n, hipHostMallocDefault);
      if (er == hipSuccess) er = hipHostMalloc((void**)&g->h_bg[i], bgb, hipHostMallocDefault);
      if (er == hipSuccess) er = hipMalloc((void**)&g->d_fg[i], g->n);
      if (er == hipSuccess) er = hipMalloc((void**)&g->d_bg[i], bgb);
    }
    if (er != hipSuccess) {
      (void)hipGetLastError();
      group_free_staging(g);
      return fail(BGS_ERR_HIP, "bgs_group_process: staging buffers: %s", hipGetErrorString(er));
    }
    g->staged = true;
  }
  for (int y = 0; y < rows; ++y) std::memcpy(g->h_in + (size_t)y * row, in + (size_t)y * in_step, row);
  HIP_TRY(hipMemcpyAsync(g->d_in, g->h_in, fb, hipMemcpyHostToDevice, g->stream));  // ONE upload for all the classes
  std::vector<uint8_t*> dfg((size_t)K), dbg((size_t)K);
  for (int i = 0; i < K; ++i) dfg[i] = (fg && fg[i]) ? g->d_fg[i] : nullptr, dbg[i] = (bg && bg[i]) ? g->d_bg[i] : nullptr;
  std::vector<uint32_t> flags((size_t)K, 0);
  const bool saved = g->borrow;
  g->borrow = false;  // the staging buffer is reused every frame: history must be a private copy
  rc = group_process_device(g, g->d_in, dfg.data(), dbg.data(), g->stream, flags.data());
  g->borrow = saved;
  if (rc) return rc;
  for (int i = 0; i < K; ++i) {
    const size_t bgc = g->algos[i] == BGS_ASBL ? 1 : channels;
    if (dfg[i] && (flags[i] & BGS_FG_VALID)) HIP_TRY(hipMemcpyAsync(g->h_fg[i], g->d_fg[i], g->n, hipMemcpyDeviceToHost, g->stream));
    if (dbg[i] && (flags[i] & BGS_BG_VALID)) HIP_TRY(hipMemcpyAsync(g->h_bg[i], g->d_bg[i], g->n * bgc, hipMemcpyDeviceToHost, g->stream));
  }
  HIP_TRY(hipStreamSynchronize(g->stream));
  for (int i = 0; i < K; ++i) {
    const size_t bgc = g->algos[i] == BGS_ASBL ? 1 : channels;
    if (dfg[i] && (flags[i] & BGS_FG_VALID))
      for (int y = 0; y < rows; ++y) std::memcpy(fg[i] + (size_t)y * (fg_step ? fg_step[i] : (size_t)cols), g->h_fg[i] + (size_t)y * cols, (size_t)cols);
    if (dbg[i] && (flags[i] & BGS_BG_VALID))
      for (int y = 0; y < rows; ++y) std::memcpy(bg[i] + (size_t)y * (bg_step ? bg_step[i] : (size_t)cols * bgc), g->h_bg[i] + (size_t)y * cols * bgc, (size_t)cols * bgc);
  }
  if (out_flags)
    for (int i = 0; i < K; ++i) out_flags[i] = flags[i];
  return BGS_OK;
}

int64_t bgs_group_get_state(bgs_group* g, int index, int stream, const char* plane, void* dst, size_t cap) {
  if (!g || !plane || !dst) return fail(BGS_ERR_INVALID, "NULL argument");
  if (index < 0 || index >= (int)g->algos.size()) return fail(BGS_ERR_INVALID, "class %d outside 0..%d", index, (int)g->algos.size() - 1);
  if (g->member[index]) return bgs_get_state(g->member[index], stream, plane, dst, cap);
  if (!g->n) return fail(BGS_ERR_STATE, "no model yet");
  if (stream < 0 || stream >= g->S) return fail(BGS_ERR_INVALID, "stream %d outside 0..%d", stream, g->S - 1);
  if (hipSetDevice(g->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(BGS_ERR_HIP, "device sync failed");
  const size_t nb = g->n * g->ch, off = nb * stream;
  const uint8_t* src = nullptr;
  const unsigned bit = g->fan_bit[index];
  const int64_t t = g->seen;
  if (!strcmp(plane, "bg") && bit == bgs::kFanSFD) src = g->sfd_bg;
  if (!strcmp(plane, "bg") && bit == bgs::kFanABL) src = g->abl_state;
  if (!strcmp(plane, "mt") && bit == bgs::kFanSD && t >= 1) src = g->sd_mt;
  if (!strcmp(plane, "vt") && bit == bgs::kFanSD && t >= 1) src = g->sd_vt;
  if (!g->borrow && (bit & (bgs::kFanFD | bgs::kFanWMM | bgs::kFanWMV)) && g->nring) {
    if (!strcmp(plane, "prev1") && t >= 1) src = g->ring[(t - 1) % g->nring];
    if (!strcmp(plane, "prev2") && t >= 2 && g->nring == 3) src = g->ring[(t - 2) % g->nring];
  }
  if (!src) return fail(BGS_ERR_STATE, "unknown state plane '%s' for class %d of the group", plane, index);
  if (cap < nb) return fail(BGS_ERR_STATE, "buffer too small for plane %s", plane);
  if (d2h_staged(dst, src + off, nb) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
  return (int64_t)nb;
}

int64_t bgs_group_frames_seen(const bgs_group* g) { return g ? g->seen : BGS_ERR_INVALID; }

int bgs_group_enable_kernel_timing(bgs_group* g, int on) {
  if (!g) return fail(BGS_ERR_INVALID, "group is NULL");
  g->timing = on != 0;
  for (auto& ev : g->events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  g->events.clear();
  return BGS_OK;
}

/* mean duration in ms of the fused launches since timing was enabled (the device is drained first) */
int bgs_group_kernel_timing(bgs_group* g, double* avg_ms, int64_t* launches) {
  if (!g || !avg_ms || !launches) return fail(BGS_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipDeviceSynchronize());
  double tot = 0;
  for (auto& ev : g->events) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev.first, ev.second);
    tot += ms;
  }
  *launches = (int64_t)g->events.size();
  *avg_ms = g->events.empty() ? 0.0 : tot / (double)g->events.size();
  return BGS_OK;
}

}  // extern "C"
