// engine_dp.h — host side of the package_bgs/dp/ models (kernel_dp.h); included inside bgs_hip.hip's anonymous namespace.
// State: `dp_planes` float planes, tiled over the global pixel index (kernel_dp.h), plus e->bgstate (mode count per pixel,
// or the AdaptiveMedian byte image).

bool is_dp(bgs_algo a) { return a >= BGS_DP_ZIVKOVIC_AGMM && a <= BGS_DP_ADAPTIVE_MEDIAN; }

int dp_planes_of(const bgs_engine* e) {
  switch (e->algo) {
    case BGS_DP_ZIVKOVIC_AGMM: return e->p.dp_gaussians * 5;
    case BGS_DP_GRIMSON_GMM: return e->p.dp_gaussians * 6;
    case BGS_DP_WREN_GA: return 4;
    case BGS_DP_MEAN: return 3;
    default: return 0;
  }
}

template <bool GRIMSON>
void dp_launch_gmm(int K, unsigned blocks, hipStream_t s, const bgs::DpArgs& a) {
  switch (K) {
    case 1: hipLaunchKernelGGL((bgs::dp_gmm_kernel<1, GRIMSON>), dim3(blocks), dim3(bgs::kBlock), 0, s, a); break;
    case 2: hipLaunchKernelGGL((bgs::dp_gmm_kernel<2, GRIMSON>), dim3(blocks), dim3(bgs::kBlock), 0, s, a); break;
    case 3: hipLaunchKernelGGL((bgs::dp_gmm_kernel<3, GRIMSON>), dim3(blocks), dim3(bgs::kBlock), 0, s, a); break;
    case 4: hipLaunchKernelGGL((bgs::dp_gmm_kernel<4, GRIMSON>), dim3(blocks), dim3(bgs::kBlock), 0, s, a); break;
    default: hipLaunchKernelGGL((bgs::dp_gmm_kernel<5, GRIMSON>), dim3(blocks), dim3(bgs::kBlock), 0, s, a); break;
  }
}

int dp_allocate(bgs_engine* e) {
  if (e->ch != 3) return fail(BGS_ERR_UNSUPPORTED, "the dp/ models read RgbImage pixels: 3-channel frames only (dp/Image.h:257-265)");
  const size_t P = e->n * e->S;
  const int planes = dp_planes_of(e);
  if (planes) {
    const size_t tiles = (P + bgs::kDpTile - 1) / bgs::kDpTile, bytes = tiles * planes * bgs::kDpTile * sizeof(float);
    if (!e->stream) HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    int rc = model_allocate(e, (void**)&e->dp_state, bytes);  // chunked placement for multi-GB models (bgs_hip.hip)
    if (rc) return rc;
  }
  // Nothing is initialised here: InitModel runs in dp_process at a stream's first frame, on the launch stream (an
  // allocation-time memset on another stream is not ordered before a kernel on the caller's / a non-blocking stream).
  return BGS_OK;
}

// one frame (number t, 0-based = the wrappers' frameNumber) for streams [first, first+count)
// frames > 1 (Zivkovic / Grimson only, from process_clip): that many consecutive frames in one launch, d_* point at the first
// slab: pixels from one frame of a clip to the next (0: the run is the whole slab)
int dp_process(bgs_engine* e, int first, int count, int64_t t, const uint8_t* d_frames, uint8_t* d_fg, uint64_t* d_bits, hipStream_t s, uint32_t* flags,
               int frames = 1, size_t slab = 0) {
  const bgs_params& p = e->p;
  bgs::DpArgs a{};
  if (!slab) slab = e->n * count;
  a.frames = frames, a.frame_stride = slab * 3, a.fg_stride = slab, a.bits_stride = slab / 64;
  a.frame = d_frames, a.state = e->dp_state, a.bstate = e->bgstate, a.fg = d_fg, a.fg_bits = d_bits;
  a.n = e->n, a.npix = e->n * count, a.first = first;
  a.low = p.dp_threshold, a.high = 2 * a.low, a.alpha = p.dp_alpha;  // HighThreshold = 2*LowThreshold, e.g. DPZivkovicAGMMBGS.cpp:58
  a.update = 0, a.xcd_swizzle = e->xcd_swizzle;
  const unsigned blocks = blocks_for(a.npix);
  if (t == 0 && (e->algo == BGS_DP_ZIVKOVIC_AGMM || e->algo == BGS_DP_GRIMSON_GMM))  // InitModel: all modes and counts 0
    hipLaunchKernelGGL(bgs::dp_gmm_clear_kernel, dim3(blocks), dim3(bgs::kBlock), 0, s, a, dp_planes_of(e));
  if (t == 0 && (e->algo == BGS_DP_WREN_GA || e->algo == BGS_DP_MEAN))  // InitModel from the first frame
    hipLaunchKernelGGL(bgs::dp_init_kernel, dim3(blocks), dim3(bgs::kBlock), 0, s, a, e->algo == BGS_DP_WREN_GA ? 4 : 3, 36.0f);
  if (t == 0 && e->algo == BGS_DP_ADAPTIVE_MEDIAN)
    HIP_TRY(hipMemcpyAsync(e->bgstate + (size_t)first * e->n * 3, d_frames, a.npix * 3, hipMemcpyDeviceToDevice, s));
  switch (e->algo) {
    case BGS_DP_ZIVKOVIC_AGMM: {
      Timed tm(e, s, "dp_gmm_kernel");
      dp_launch_gmm<false>(p.dp_gaussians, blocks, s, a);
      break;
    }
    case BGS_DP_GRIMSON_GMM: {
      Timed tm(e, s, "dp_gmm_kernel");
      dp_launch_gmm<true>(p.dp_gaussians, blocks, s, a);
      break;
    }
    case BGS_DP_WREN_GA: {
      Timed tm(e, s, "dp_wren_kernel");
      hipLaunchKernelGGL(bgs::dp_wren_kernel, dim3(blocks), dim3(bgs::kBlock), 0, s, a);
      break;
    }
    case BGS_DP_MEAN: {
      Timed tm(e, s, "dp_mean_kernel");
      hipLaunchKernelGGL(bgs::dp_mean_kernel, dim3(blocks), dim3(bgs::kBlock), 0, s, a);
      break;
    }
    default: {
      a.update = (t % p.dp_sampling_rate) == 1;  // AdaptiveMedianBGS.cpp:60
      Timed tm(e, s, "dp_median_kernel");
      const uint8_t* med0 = e->bgstate + (size_t)first * e->n * 3;
      if (a.npix % 4 == 0 && aligned(d_frames, 4) && aligned(med0, 4) && (!d_fg || aligned(d_fg, 4)))
        hipLaunchKernelGGL((bgs::dp_median_kernel<4>), dim3(blocks_for(a.npix / 4)), dim3(bgs::kBlock), 0, s, a);
      else
        hipLaunchKernelGGL((bgs::dp_median_kernel<1>), dim3(blocks), dim3(bgs::kBlock), 0, s, a);
      break;
    }
  }
  *flags = BGS_FG_VALID;  // img_bgmodel is never written by the dp wrappers
  return BGS_OK;
}

// canonical export [plane][n] of one stream from the tiled device layout
int64_t dp_export_planes(bgs_engine* e, int stream, int planes, void* dst, size_t cap) {
  const size_t n = e->n, g0 = (size_t)stream * n, T = bgs::kDpTile;
  if (cap < (size_t)planes * n * 4) return fail(BGS_ERR_STATE, "buffer too small for %d planes", planes);
  const size_t t0 = g0 / T, t1 = (g0 + n - 1) / T + 1, TF = (size_t)planes * T;
  std::vector<float> tiles((t1 - t0) * TF);
  if (d2h_staged(tiles.data(), e->dp_state + t0 * TF, tiles.size() * 4) != BGS_OK) return fail(BGS_ERR_HIP, "hipMemcpy failed");
  for (int q = 0; q < planes; ++q)
    for (size_t i = 0; i < n; ++i) {
      const size_t g = g0 + i;
      ((float*)dst)[(size_t)q * n + i] = tiles[(g / T - t0) * TF + (size_t)q * T + g % T];
    }
  return (int64_t)planes * n * 4;
}
