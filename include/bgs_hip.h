/*
 * bgs_hip.h — C ABI of libbgs_hip, the MI355X (gfx950) foreground-detection engine.
 *
 * This is the drop-in boundary for the reference's package_bgs hot path.  The
 * reference's own surface is C++ with OpenCV types in the signature
 *     IBGS::process(const cv::Mat&, cv::Mat&, cv::Mat&)     package_bgs/IBGS.h:24
 *     USTC_BGS::Process(IplImage*) / GetMask()              ustc_src/ustc_bgs.cpp:79-113
 * so what crosses here is exactly what those calls carry once the cv::Mat /
 * IplImage header is peeled off: (data pointer, rows, cols, channels, row step)
 * for the input frame, the foreground mask and the background image.
 * No OpenCV, HIP or torch types appear in any signature; every function returns
 * 0 on success or a negative bgs_status, never throws.
 *
 * One engine owns the model state of n_streams independent camera streams of one
 * geometry, resident in HBM as SoA planes.  An engine is not
 * thread-safe (neither is an IBGS instance); use one per thread.
 */
#ifndef BGS_HIP_H
#define BGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGS_ABI_VERSION 1

typedef struct bgs_engine bgs_engine;

/* One id per reference IBGS class on the hot path.  The comment
 * names the reference process() each one replaces. */
typedef enum bgs_algo {
  BGS_FRAME_DIFF = 0,        /* FrameDifferenceBGS::process        package_bgs/FrameDifferenceBGS.cpp:29-61 */
  BGS_STATIC_FRAME_DIFF = 1, /* StaticFrameDifferenceBGS::process  package_bgs/StaticFrameDifferenceBGS.cpp:29-57 */
  BGS_WMM = 2,               /* WeightedMovingMeanBGS::process     package_bgs/WeightedMovingMeanBGS.cpp:29-96 */
  BGS_WMV = 3,               /* WeightedMovingVarianceBGS::process package_bgs/WeightedMovingVarianceBGS.cpp:30-117 */
  BGS_ABL = 4,               /* AdaptiveBackgroundLearning::process package_bgs/AdaptiveBackgroundLearning.cpp:30-83 */
  BGS_ASBL = 5,              /* AdaptiveSelectiveBackgroundLearning::process package_bgs/AdaptiveSelectiveBackgroundLearning.cpp:31-105 */
  BGS_MOG2 = 6,              /* MixtureOfGaussianV2BGS::process    package_bgs/MixtureOfGaussianV2BGS.cpp:29-74 */
  BGS_MOG1 = 7,              /* MixtureOfGaussianV1BGS::process    package_bgs/MixtureOfGaussianV1BGS.cpp:29-71 */
  BGS_GMG = 8,               /* GMG::process                       package_bgs/GMG.cpp:35-77 */
  BGS_SUBSENSE = 9,          /* SuBSENSEBGS::process               package_bgs/pl/SuBSENSE.cpp:21-45 */
  BGS_LBSP_DESC = 10,        /* LBSP::computeRGBDescriptor         package_bgs/pl/LBSP.h:50-95 */
  BGS_SIGMA_DELTA = 11,      /* SigmaDeltaBGS::process             package_bgs/bl/SigmaDeltaBGS.cpp */
  /* the in-tree dp/ models (3-channel frames only; output = the high-threshold mask, no background image) */
  BGS_DP_ZIVKOVIC_AGMM = 12, /* DPZivkovicAGMMBGS::process         package_bgs/dp/DPZivkovicAGMMBGS.cpp:29-80 */
  BGS_DP_GRIMSON_GMM = 13,   /* DPGrimsonGMMBGS::process           package_bgs/dp/DPGrimsonGMMBGS.cpp:29-82 */
  BGS_DP_WREN_GA = 14,       /* DPWrenGABGS::process               package_bgs/dp/DPWrenGABGS.cpp:29-81 */
  BGS_DP_MEAN = 15,          /* DPMeanBGS::process                 package_bgs/dp/DPMeanBGS.cpp:29-82 */
  BGS_DP_ADAPTIVE_MEDIAN = 16, /* DPAdaptiveMedianBGS::process     package_bgs/dp/DPAdaptiveMedianBGS.cpp:29-81 */
  BGS_LOBSTER = 17,          /* LOBSTERBGS::process                package_bgs/pl/LOBSTER.cpp:20-45 */
  BGS_ALGO_COUNT
} bgs_algo;

typedef enum bgs_status {
  BGS_OK = 0,
  BGS_ERR_INVALID = -1,      /* bad argument (NULL, negative size, unknown enum) */
  BGS_ERR_UNSUPPORTED = -2,  /* input the reference would CV_Assert on (e.g. MOG2 background image of a 1-channel frame) */
  BGS_ERR_GEOMETRY = -3,     /* rows/cols/channels differ from the first frame of this engine */
  BGS_ERR_HIP = -4,          /* HIP runtime failure; text in bgs_last_error() */
  BGS_ERR_NOMEM = -5,
  BGS_ERR_STATE = -6         /* unknown state plane / buffer too small */
} bgs_status;

/* out_flags bits of bgs_process*: the reference leaves img_output / img_bgmodel
 * untouched on warm-up frames and for algorithms that never write a background
 * (SURVEY.md App. C 1-2); callers must not read a buffer whose bit is clear. */
#define BGS_FG_VALID 1u
#define BGS_BG_VALID 2u

/*
 * Parameters.  Field defaults are the values the reference's constructors and
 * loadConfig() fall back to when ./config/<Class>.xml is absent (file:line in
 * the comments).  Zero-initialise, set struct_size = sizeof(bgs_params), call
 * bgs_default_params(algo, &p), then override.
 */
typedef struct bgs_params {
  uint32_t struct_size;

  /* wrapper-level, shared by every IBGS class: cv::threshold(fg, threshold, 255, THRESH_BINARY) */
  int32_t enable_threshold;   /* 1   e.g. FrameDifferenceBGS.cpp:84 */
  int32_t threshold;          /* 15  (ASBL: 25, AdaptiveSelectiveBackgroundLearning.cpp:127) */

  /* WeightedMovingMeanBGS / WeightedMovingVarianceBGS */
  int32_t enable_weight;      /* 1   WeightedMovingMeanBGS.cpp:117 */

  /* AdaptiveBackgroundLearning, MixtureOfGaussianV1/V2 (learning rate) */
  double alpha;               /* 0.05  AdaptiveBackgroundLearning.cpp:103, MixtureOfGaussianV2BGS.cpp:92 ; MOG1: 0.05 */
  int32_t limit;              /* -1  AdaptiveBackgroundLearning.cpp:104 */

  /* AdaptiveSelectiveBackgroundLearning */
  int32_t learning_frames;    /* 90    AdaptiveSelectiveBackgroundLearning.cpp:124 */
  double alpha_learn;         /* 0.05  :125 */
  double alpha_detection;     /* 0.05  :126 */

  /* cv::BackgroundSubtractorMOG2 defaults */
  int32_t mog2_history;        /* 500 */
  int32_t mog2_nmixtures;      /* 5 (compile-time K of the kernel; other values -> BGS_ERR_UNSUPPORTED) */
  float mog2_var_threshold;    /* Tb = 16 */
  float mog2_background_ratio; /* TB = 0.9 */
  float mog2_var_threshold_gen;/* Tg = 9 */
  float mog2_var_init;         /* 15 */
  float mog2_var_min;          /* 4 */
  float mog2_var_max;          /* 75 */
  float mog2_ct;               /* 0.05 */
  float mog2_tau;              /* 0.5 */
  int32_t mog2_detect_shadows; /* 1 */
  int32_t mog2_shadow_value;   /* 127 */

  /* cv::BackgroundSubtractorMOG defaults */
  int32_t mog1_history;        /* 200 */
  int32_t mog1_nmixtures;      /* 5 */
  double mog1_background_ratio;/* 0.7 */
  double mog1_var_threshold;   /* 2.5*2.5 */
  double mog1_noise_sigma;     /* 15 (30*0.5) */

  /* LBSP descriptor / SuBSENSE (package_bgs/pl/SuBSENSE.cpp:8-14).  LOBSTER (package_bgs/pl/LOBSTER.cpp:5-12) uses the same
   * fields: lbsp_rel_threshold (0.365), lbsp_threshold_offset (0), subsense_min_color_dist_threshold = nColorDistThreshold (30),
   * subsense_n_samples = nBGSamples (35), subsense_n_required (2), subsense_desc_dist_threshold_offset = nDescDistThreshold (4). */
  float lbsp_rel_threshold;    /* 0.333 */
  int32_t lbsp_threshold_offset;/* 0 (base-ctor default) */
  int32_t subsense_min_color_dist_threshold; /* 30 */
  int32_t subsense_n_samples;  /* 50 */
  int32_t subsense_n_required; /* 2 */
  int32_t subsense_samples_for_moving_avgs; /* 100 */

  /* SigmaDelta (package_bgs/bl/SigmaDeltaBGS.cpp) */
  int32_t sd_amp_factor;       /* 1 */
  int32_t sd_min_var;          /* 15 */
  int32_t sd_max_var;          /* 255 */

  int32_t subsense_desc_dist_threshold_offset; /* 3 (BGSSUBSENSE_DEFAULT_DESC_DIST_THRESHOLD_OFFSET) */

  /* cv::BackgroundSubtractorGMG as GMG::process configures it (package_bgs/GMG.cpp:44-45) */
  int32_t gmg_max_features;        /* 64 (compile-time bound of the kernel: <= 64) */
  int32_t gmg_init_frames;         /* 20  GMG.cpp:19, :44 (OpenCV default 120) */
  int32_t gmg_quantization_levels; /* 16 */
  int32_t gmg_smoothing_radius;    /* 7 (cv::medianBlur kernel size; 0 = off) */
  int32_t gmg_update_background_model; /* 1 */
  int32_t dp_sampling_rate;        /* dp/ AdaptiveMedian: 7  DPAdaptiveMedianBGS.cpp:19 */
  double gmg_learning_rate;        /* 0.025 */
  double gmg_background_prior;     /* 0.8 */
  double gmg_decision_threshold;   /* 0.7  GMG.cpp:19, :45 (OpenCV default 0.8) */

  /* package_bgs/dp/ wrappers (DP*BGS.cpp:19 constructors): threshold = LowThreshold (HighThreshold = 2x, the mask that is
   * returned), alpha, gaussians = MaxModes (1..8).  learningFrames of WrenGA / Mean / AdaptiveMedian = learning_frames above
   * (default 30 for those three; it has no effect because the wrappers clear the update mask every frame). */
  float dp_threshold;   /* Zivkovic 25, Grimson 9, WrenGA 12.25, Mean 2700, AdaptiveMedian 40 */
  float dp_alpha;       /* Zivkovic 0.001, Grimson 0.01, WrenGA 0.005, Mean 1e-6 */
  int32_t dp_gaussians; /* 3 */
} bgs_params;

int bgs_abi_version(void);

/* Fill *p (struct_size must be set) with the reference defaults for `algo`. */
int bgs_default_params(bgs_algo algo, bgs_params* p);

/* Create an engine for `n_streams` independent streams on HIP device `hip_device`.
 * params == NULL selects the reference defaults.  Model memory is allocated
 * lazily at the first frame (like every reference class) or by bgs_set_geometry. */
int bgs_create(bgs_algo algo, const bgs_params* params, int hip_device, int n_streams, bgs_engine** out);

/* Replace the wrapper-level parameters between frames.  The reference re-reads
 * ./config/<Class>.xml at the top of every process() (e.g. MixtureOfGaussianV2BGS.cpp:34),
 * so thresholds / alpha may change mid-stream; structural fields (nmixtures, n_samples)
 * must stay what they were at bgs_create. */
int bgs_set_params(bgs_engine* e, const bgs_params* params);

/* Engine options.
 *   BGS_OPT_BORROW_FRAMES  device path of the history-keeping classes (FrameDifference, WeightedMovingMean/Variance):
 *                          1 = use the caller's previous d_frames buffers as history instead of copying each frame
 *                          into the engine's ring; the buffers handed to the previous one (FD) or two (WMM/WMV)
 *                          whole-batch calls must then stay valid and unchanged.  Default 0 (private copy).
 *   BGS_OPT_MOG2_PIXELS_PER_LANE, BGS_OPT_MOG2_TILED  round-2 A/B knobs of the sorted-array MOG2 kernel; accepted and ignored since
 *                          round 3 (one pixel per lane, one model layout: ranked weights + fixed-slot records + summaries).
 *   BGS_OPT_XCD_SWIZZLE    XCD-aware workgroup order: 1 (default) = for the kernels that stream a multi-plane model
 *                          (MOG2, MOG1, dp/), 2 = also for the byte-stream kernels (slower there: A/B only), 0 = off.
 *   BGS_OPT_PLACEMENT_PROBE  rounds 1-2 looked for a fast physical placement of a multi-GB model by trial; accepted and ignored since
 *                          round 3, which places models deterministically (BGS_OPT_MODEL_CHUNK_MB).
 *   BGS_OPT_MODEL_CHUNK_MB multi-GB models (MOG2, MOG1, dp/) are ONE virtual range backed by separately created physical chunks of this
 *                          many MiB (default 256; any size from 2 MiB to 1 GiB avoids the slow placement - a physically contiguous run
 *                          above ~1 GiB - that one big hipMalloc of a long-lived process sometimes gets: 8-10 % on the MOG2 kernel; it is
 *                          not faster than a fresh plain allocation); 0 = one plain allocation; before the geometry is set. */
#define BGS_OPT_BORROW_FRAMES 1
#define BGS_OPT_MOG2_PIXELS_PER_LANE 2
#define BGS_OPT_MOG2_TILED 3
#define BGS_OPT_XCD_SWIZZLE 4
#define BGS_OPT_PLACEMENT_PROBE 5
#define BGS_OPT_MOG2_SPARSE 6   /* which MOG2 per-frame kernel runs (identical results, different traffic): 0 dense (everything read and
                                   written back: A/B, placement probe); 1 eager (every record read, only what changed written); 2 count
                                   (only the modes a pixel has are read); 4 filter (4-byte summaries first, then only the records they
                                   cannot rule out); 3 (default) automatic, from what sampled workgroups count */
#define BGS_OPT_CLIP_FUSE 7     /* 1 (default): bgs_process_clip_device runs 8 / 4 / 2 consecutive frames of a mixture model per launch with the
                                   model held in registers; 0: one launch per frame.  Identical results, only speed differs. */
#define BGS_OPT_MODEL_CHUNK_MB 9
#define BGS_OPT_MODEL_CHUNK_MIN_MB 10 /* models smaller than this many MiB take one plain allocation (default 768: below that a good part of the
                                   model sits in the 256 MiB Infinity Cache and placement does not matter); before the geometry is set.
                                   Lowered by the test that exercises the chunked construction and its release on a small model. */
#define BGS_OPT_HOST_REGISTER 8 /* bgs_process (host buffers): bit 0 input frame, bit 1 mask, bit 2 background image.  A buffer passed in an
                                   enabled role that comes back with the same address and size as in the previous call is page-locked once
                                   (hipHostRegister) and from then on read / written by the DMA engine in place - no staging copy by the
                                   CPU (OpenCV's capture loop hands IBGS::process the same frame buffer every frame, VideoCapture.cpp:158-218).
                                   Contract: such a buffer stays allocated until a different one is passed in that role or the engine is
                                   destroyed (the engine cannot see a buffer being freed and another mapped at the same address).
                                   Default 0: every image is staged through the engine's own pinned buffers.  Rows must be contiguous. */
int bgs_set_option(bgs_engine* e, int option, int64_t value);

/* Fix rows x cols x channels up front and allocate the model (device path). */
int bgs_set_geometry(bgs_engine* e, int rows, int cols, int channels);

/*
 * One frame of one stream, host buffers (the IBGS::process call).
 *   in       rows x cols x channels uint8, interleaved (BGR), row stride in_step bytes (>= cols*channels)
 *   fg       rows x cols uint8 mask or NULL; written only if BGS_FG_VALID is reported
 *   bg       rows x cols x channels (ASBL: x1) uint8 or NULL; written only if BGS_BG_VALID
 * in == NULL or rows*cols == 0 is the reference's `if(img_input.empty()) return;`:
 * returns BGS_OK with *out_flags = 0 and no state change.
 * Synchronous: the outputs are complete on return.
 */
int bgs_process(bgs_engine* e, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step,
                uint8_t* fg, size_t fg_step, uint8_t* bg, size_t bg_step, uint32_t* out_flags);

/*
 * One frame of EVERY stream, device buffers, asynchronous on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  This is the
 * roofline path: no PCIe traffic, one launch over streams x pixels.
 *   d_frames  [n_streams][rows][cols][channels] uint8, contiguous
 *   d_fg      [n_streams][rows][cols] uint8 or NULL
 *   d_bg      [n_streams][rows][cols][channels] uint8 or NULL
 *   d_fg_bits [n_streams][W] uint64, W = ceil(rows*cols/64): bit i of word j of a stream = its pixel 64j+i is foreground; the bits of
 *             a stream's last word past pixel rows*cols-1 are zero; or NULL.  (rows*cols a multiple of 64 - 1080p, 4K, 720p, VGA,
 *             320x176 ... - is the fast case: the update kernels write the words from wave ballots; for any other size the words are
 *             made from the byte mask by one extra small launch, through an engine-owned buffer when d_fg is NULL.)
 * The streams need NOT be in lock-step: cameras join, drop frames and are reset independently (the reference creates and deletes
 * one IBGS object per stream whenever it likes, FrameProcessor.cpp:35-155, :342-482).  Streams whose next frame needs the same
 * kernel arguments share a launch - for the mixture models that is "first frame or not" plus the learning rate, i.e. every
 * stream past its first frame under the wrappers' fixed alpha; for the history classes the warm-up level; for SuBSENSE / LOBSTER
 * the frame index itself - so a batch fed by whole-batch calls is ONE launch, and a straggler costs one more.
 * out_flags: what holds for every stream of the call; per stream see bgs_stream_flags.
 * (BGS_OPT_BORROW_FRAMES still needs lock-step: the borrowed history is one buffer for the whole batch.)
 */
int bgs_process_batch_device(bgs_engine* e, const void* d_frames, void* d_fg, void* d_bg, void* d_fg_bits,
                             void* hip_stream, uint32_t* out_flags);

/* Same, restricted to streams [first, first+count): d_* point at the first of `count` images. */
int bgs_process_range_device(bgs_engine* e, int first, int count, const void* d_frames, void* d_fg, void* d_bg,
                             void* d_fg_bits, void* hip_stream, uint32_t* out_flags);

/*
 * A CLIP: `nframes` consecutive frames of each stream in [first, first+count) in one call - the file-fed case of the reference
 * (VideoCapture::start reads a video file, VideoCapture.cpp:158-207: frames are available ahead of the model), or a live
 * deployment that accepts nframes-1 frame times of latency.  Results are exactly those of nframes successive
 * bgs_process_range_device calls (masks, backgrounds, model state, frame counts).
 *   d_frames  [nframes][count][rows][cols][channels]   frame t of all streams, then frame t+1 ...
 *   d_fg      [nframes][count][rows][cols] or NULL;  d_bg [nframes][count][rows][cols][channels] or NULL
 *   d_fg_bits [nframes][count][ceil(rows*cols/64)] or NULL;  out_flags: nframes words or NULL
 * The mixture models (MixtureOfGaussianV2BGS, MixtureOfGaussianV1BGS, DPZivkovicAGMMBGS, DPGrimsonGMMBGS) take runs of 8 / 4 / 2
 * frames through ONE launch that loads each pixel's model once, applies the frames in order in registers and writes the model back
 * once (model traffic per frame / 8, / 4, / 2); FrameDifference / WeightedMovingMean / WeightedMovingVariance take their frame history
 * from the clip itself instead of copying every frame into the engine's ring; every other class runs the same launches as the
 * frame-by-frame calls.
 */
int bgs_process_clip_device(bgs_engine* e, int first, int count, int nframes, const void* d_frames, void* d_fg, void* d_bg,
                            void* d_fg_bits, void* hip_stream, uint32_t* out_flags);

/*
 * Copy one model plane of one stream to host memory, as a dense array in the
 * canonical order documented in DESIGN.md §3 (e.g. MOG2: "w" float[K][N],
 * "var" float[K][N], "mu" float[K][C][N], "nmodes" uint8[N]).  Returns the
 * number of bytes written (>= 0) or a negative status.  Parity tests only.
 */
int64_t bgs_get_state(bgs_engine* e, int stream, const char* plane, void* dst, size_t cap);

/* Number of frames stream `stream` has consumed so far (since bgs_create or its last bgs_reset_stream). */
int64_t bgs_frames_seen(const bgs_engine* e, int stream);

/*
 * GROUPS - several classes on the same frames.  FrameProcessor::process hands ONE pre-processed frame to every enabled IBGS, one
 * call after the other (FrameProcessor.cpp:169-340); BASELINE configs[2] is WeightedMovingVarianceBGS + AdaptiveBackgroundLearning
 * on the same 3840x2160 frames.  A group takes `n_algos` classes; the byte-stream ones among them - FrameDifference,
 * StaticFrameDifference, WeightedMovingMean, WeightedMovingVariance, AdaptiveBackgroundLearning, SigmaDelta, one instance each -
 * run as ONE kernel over one read of the frame and one shared history ring (17 B/pixel for configs[2] instead of 10 + 10; 26
 * instead of 47 for the five history / state classes together), every other class as a member engine fed the same device frame.
 * Masks, backgrounds, warm-up conventions and model states are those of n separate engines, bit for bit.
 *   params      NULL, or n_algos pointers (NULL entries = reference defaults)
 *   d_fg, d_bg  arrays of n_algos device pointers (entries may be NULL), each laid out as for bgs_process_batch_device
 *   out_flags   n_algos words (BGS_FG_VALID / BGS_BG_VALID per class)
 * The streams of a group advance in lock-step (whole-batch calls).  bgs_group_process is the host-buffer call for single-stream
 * groups - one upload of the frame for all classes - with per-class output pointers and row steps (NULL steps = packed rows).
 */
typedef struct bgs_group bgs_group;
int bgs_group_create(const bgs_algo* algos, const bgs_params* const* params, int n_algos, int hip_device, int n_streams, bgs_group** out);
void bgs_group_destroy(bgs_group* g);
int bgs_group_size(const bgs_group* g);
int bgs_group_is_fused(const bgs_group* g, int index); /* 1: class `index` runs inside the fused kernel; 0: member engine */
int bgs_group_set_params(bgs_group* g, int index, const bgs_params* params);
int bgs_group_set_option(bgs_group* g, int option, int64_t value); /* BGS_OPT_BORROW_FRAMES: the shared history too */
int bgs_group_set_geometry(bgs_group* g, int rows, int cols, int channels);
int bgs_group_process_batch_device(bgs_group* g, const void* d_frames, void* const* d_fg, void* const* d_bg, void* hip_stream, uint32_t* out_flags);
int bgs_group_process(bgs_group* g, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* const* fg, const size_t* fg_step,
                      uint8_t* const* bg, const size_t* bg_step, uint32_t* out_flags);
int64_t bgs_group_get_state(bgs_group* g, int index, int stream, const char* plane, void* dst, size_t cap);
int64_t bgs_group_frames_seen(const bgs_group* g);
int bgs_group_enable_kernel_timing(bgs_group* g, int on);
int bgs_group_kernel_timing(bgs_group* g, double* avg_ms, int64_t* launches); /* the fused launches, HIP events on the launch stream */

/*
 * Several cameras on the host path.  bgs_submit queues one frame of one stream - upload, kernels, download - on that stream's own
 * lane (HIP stream, pinned staging, device images) and returns; bgs_wait blocks until that submission is done and the outputs are
 * in the caller's images (out_flags as for bgs_process).  Submissions of different streams overlap: while camera A's frame is on
 * the bus camera B's kernel runs and camera C's mask comes back, where a synchronous bgs_process per camera takes turns.
 * Same arguments and results as bgs_process; the caller's buffers must stay valid and untouched until bgs_wait; one submission per
 * stream in flight (bgs_process on such a stream collects it first; the device-path calls - bgs_process_batch_device / _range_device /
 * _clip_device - REFUSE a range that covers a stream with a submission in flight, BGS_ERR_STATE: they run on the caller's HIP stream
 * and would race with the lane).  All bgs_submit / bgs_wait calls of one engine come from ONE host thread (an engine is not thread-safe,
 * like an IBGS instance).  Frames prepared by bgs_set_ingest go through bgs_process.
 */
int bgs_submit(bgs_engine* e, int stream, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg, size_t fg_step,
               uint8_t* bg, size_t bg_step);
int bgs_wait(bgs_engine* e, int stream, uint32_t* out_flags);

/* A caller that keeps the images of ALL its cameras in one allocation (a capture ring, a frame pool) registers that allocation once:
 * [ptr, ptr + bytes) is page-locked as a whole (one hipHostRegister instead of one per camera and role) and from then on every input
 * frame, mask or background image of bgs_process / bgs_submit that lies inside it with contiguous rows is read / written by the DMA
 * engine in place - whatever BGS_OPT_HOST_REGISTER says, and without the "same buffer as last call" test.  on = 0 drops the
 * registration (same ptr).  The arena stays allocated until then or until bgs_destroy. */
int bgs_host_arena(bgs_engine* e, void* ptr, size_t bytes, int on);

/* One camera of the batch starts over - what `delete bgs; bgs = new <Class>;` is for one stream in the reference
 * (FrameProcessor.cpp:342-482 / :35-155, ustc_src/ustc_bgs.cpp:75-77): its frame count returns to 0 and its NEXT frame, on the
 * HIP stream of that call and in order with everything queued before it, re-initialises its model and restarts its warm-up
 * exactly like a first frame.  No launch happens here; the other streams are not touched. */
int bgs_reset_stream(bgs_engine* e, int stream);

/* out_flags (BGS_FG_VALID / BGS_BG_VALID) of the stream's last frame: in a call over streams of different ages the call's own
 * out_flags word only holds what is true of all of them. */
int bgs_stream_flags(const bgs_engine* e, int stream, uint32_t* out_flags);

/* Average duration in ms of the dominant kernel of this engine since the last reset, measured
 * with HIP events on the launch stream when timing is enabled (bench.py's roofline leg). */
int bgs_enable_kernel_timing(bgs_engine* e, int on);
int bgs_kernel_timing(bgs_engine* e, double* avg_ms, int64_t* launches, const char** kernel_name);
/* The same measurement launch by launch: copies the durations (ms) of the first min(cap, launches) timed launches since the
 * last reset into ms[] and returns how many were written (negative = error).  bench.py commits the series behind its
 * burst / sustained figures from this (profiles/rNN_mog2_launch_series.csv). */
int64_t bgs_kernel_timing_series(bgs_engine* e, float* ms, int64_t cap);

/*
 * Measurement aids: what THIS box delivers, measured by the library in the process that benchmarks it (bench.py `calibration`,
 * `host_path`).  Boxes of one pool differ by several percent, and so does the physical placement of a multi-GB allocation; a figure
 * that is to be compared across boxes needs the box's own yardstick beside it.
 *   bgs_calibrate_copy  float4 copy kernel over `bytes` bytes (bytes/2 read + bytes/2 written per launch; mean of `iters` launches after
 *                       two warm-ups, HIP events): through ONE plain hipMalloc (chunk_mb = 0) or through one virtual range backed by
 *                       physical chunks of chunk_mb MiB - the construction the big models use (BGS_OPT_MODEL_CHUNK_MB).  *gbps in GB/s.
 *   bgs_calibrate_pcie  `iters` copies of `bytes` bytes each way between the device and page-locked host memory: hipHostMalloc
 *                       (registered = 0: the engine's own staging) or ordinary memory page-locked in place with hipHostRegister
 *                       (registered = 1: a caller buffer under BGS_OPT_HOST_REGISTER; *register_ms = what that call took).
 * bgs_get_state(e, 0, "hostpath", double[15]) returns an engine's host-path counters since creation: buffers currently page-locked in
 * place per role [0..2] (input, mask, background) and refused per role [3..5], hipHostRegister calls [6] and their total ms [7],
 * hipHostUnregister calls [8], frames [9], bytes host-to-device [10] and device-to-host [11], CPU ms spent copying into [12] and out
 * of [13] pinned staging, arenas registered [14].
 */
int bgs_calibrate_copy(int hip_device, size_t bytes, int chunk_mb, int iters, double* gbps);
int bgs_calibrate_pcie(int hip_device, size_t bytes, int registered, int iters, double* h2d_gbps, double* d2h_gbps, double* register_ms);

void bgs_destroy(bgs_engine* e);

/* Thread-local text of the last failure on this thread ("" if none). */
const char* bgs_last_error(void);

/* ---- stand-alone device primitives (rows §8a10, §8f N1) ------------------------------ */

/* State planes of the dp/ models (bgs_get_state): "modes" f32 [K*F][n] with F = 5 (sigma, mu0, mu1, mu2, weight) for
 * Zivkovic and 6 (variance, mu0, mu1, mu2, weight, significants) for Grimson, plane index k*F + f; "nmodes" u8 [n];
 * WrenGA "gauss" f32 [4][n] (mu0..2, var); Mean "mean" f32 [3][n]; AdaptiveMedian "median" u8 [n*3]. */

/* LBSP 16-bit double-cross descriptors of a whole 8UC3 / 8UC1 image (LBSP.h:50-95,
 * LBSP_16bits_dbcross_{3ch3t,1ch}.i).  d_desc: [rows][cols][channels] uint16; the
 * 2-pixel border (LBSP::validateROI, LBSP.cpp:311-318) is written as 0.
 * t_lut: 256 absolute thresholds indexed by the centre value
 * (BackgroundSubtractorSuBSENSE.cpp:209-210, 227-228), host pointer. */
int bgs_lbsp_describe_device(int hip_device, const void* d_img, int rows, int cols, int channels,
                             const uint8_t* t_lut, void* d_desc, void* hip_stream);
/* the same for `images` frames stored back to back ([images][rows][cols][channels]) in one launch */
int bgs_lbsp_describe_batch_device(int hip_device, const void* d_img, int images, int rows, int cols, int channels,
                                   const uint8_t* t_lut, void* d_desc, void* hip_stream);

/* 3x3 morphology / median / hole-fill post-processing of a byte mask on device
 * (BackgroundSubtractorSuBSENSE.cpp:624-640). op: 0 erode3x3, 1 dilate3x3, 2 median(ksize),
 * 3 median(ksize) of a {0,255} mask (majority count, same result, faster), 4 cv::floodFill(img, Point(0,0), 255). */
int bgs_mask_morph_device(int hip_device, const void* d_src, void* d_dst, int rows, int cols, int op, int ksize,
                          int iterations, void* hip_stream);

/* Connected-component seeding of a byte mask on device: what OpenCV-legacy's blob detector derives
 * from the mask right after IBGS::process (ustc_src/trackingMain.cpp:56-57, :166; cvFindContours + bounding rectangles;
 * in-tree kin: package_bgs/jmo/BlobExtraction.cpp).  Components are the maximal 8- (or 4-) connected sets of non-zero
 * pixels.  A component is named by `root`, the raster index (y*cols + x) of its first pixel; boxes are written sorted by
 * root, so the output is deterministic.  *d_count receives the number of components found (it may exceed max_boxes;
 * only the first max_boxes are written).  d_labels (optional, [rows*cols] int32): root of each pixel's component, -1 for
 * background.  d_work: bgs_mask_components_workspace(rows, cols) bytes of device scratch; with d_work the call is
 * asynchronous on hip_stream, with NULL it allocates, synchronises and frees (slow path). */
typedef struct bgs_box {
  int32_t x, y, w, h; /* bounding rectangle */
  int32_t area;       /* pixels in the component */
  int32_t root;       /* raster index of the component's first pixel */
} bgs_box;
size_t bgs_mask_components_workspace(int rows, int cols);
int bgs_mask_components_device(int hip_device, const void* d_mask, int rows, int cols, int connectivity,
                               int32_t* d_labels, bgs_box* d_boxes, int max_boxes, int32_t* d_count, void* d_work,
                               void* hip_stream);
/* The same for `images` masks stored back to back ([images][rows][cols], e.g. the d_fg of bgs_process_batch_device): one
 * set of launches for all of them.  Boxes are sorted by (image, root); d_offsets [images+1] receives the prefix sums of the
 * per-image component counts, so image k owns d_boxes[d_offsets[k] .. d_offsets[k+1]) (entries past max_boxes are not
 * written); roots and labels are relative to their own image. */
size_t bgs_mask_components_batch_workspace(int images, int rows, int cols);
int bgs_mask_components_batch_device(int hip_device, const void* d_masks, int images, int rows, int cols, int connectivity,
                                     int32_t* d_labels, bgs_box* d_boxes, int max_boxes, int32_t* d_offsets, void* d_work,
                                     void* hip_stream);

/* ---- N2: blob list hand-off (ustc_src/trackingMain.cpp:166: the blob detector consumes the mask right after FG detection) ----
 * OpenCV-legacy's detectors turn each foreground region into a CvBlob {x, y, w, h, ID} from the region's bounding rectangle
 * and its coordinate moments (centre = centroid, size = 4 sigma; recalled from modules/legacy/src/enteringblobdetection.cpp,
 * not in the tree).  Both are produced here, per component, so the caller can apply either convention
 * (tracking_amd/host/blob.h: blob_from_box / blob_from_moments). */
typedef struct bgs_moments {
  int64_t sx, sy;   /* sum of x, sum of y over the component's pixels (image coordinates) */
  int64_t sxx, syy; /* sum of x*x, sum of y*y */
} bgs_moments;
/* bgs_mask_components_batch_device plus the moments of every component (d_moments [max_boxes], same order as d_boxes;
 * NULL = not wanted).  No labels output. */
int bgs_mask_blobs_batch_device(int hip_device, const void* d_masks, int images, int rows, int cols, int connectivity,
                                bgs_box* d_boxes, bgs_moments* d_moments, int max_boxes, int32_t* d_offsets, void* d_work,
                                void* hip_stream);
/* Host path: components of the foreground mask that the LAST bgs_process call of `stream` produced, computed on the device
 * copy of that mask (bgs_process keeps it even when its `fg` argument is NULL, so a caller that only wants blobs moves no
 * mask over PCIe); components narrower than min_w or lower than min_h are dropped; host arrays boxes / moments (optional)
 * [max_boxes] receive the first max_boxes survivors in raster order of their first pixel, *count their number.  Synchronous.
 * BGS_ERR_STATE if that call produced no valid mask (warm-up frame) or another stream has been processed since. */
int bgs_last_mask_blobs(bgs_engine* e, int stream, int connectivity, int min_w, int min_h, bgs_box* boxes, bgs_moments* moments,
                        int max_boxes, int32_t* count);

/* ---- N3: frame preparation on the device (the step BEFORE the path) --------------------------------------------------------
 * What VideoCapture::start (VideoCapture.cpp:158-207: cvResize to input_resize_percent, cvFlip(frame, frame, 0), ROI view) and
 * PreProcessor::process (PreProcessor.cpp:46-77: optional cv::equalizeHist, optional cv::GaussianBlur 7x7 sigma 1.5) do to a
 * captured frame before FrameProcessor hands it to every IBGS::process, in that order, as one device pass over frames that
 * are already in HBM (a decoder's output) or as a host-buffer convenience.  Flip and ROI are exact by definition; resize,
 * equalizeHist and GaussianBlur restate OpenCV 2.4's 8-bit fixed-point arithmetic from recall (unpinned). */
typedef struct bgs_ingest {
  uint32_t struct_size;     /* sizeof(bgs_ingest) */
  int32_t resize_percent;   /* VideoCapture input_resize_percent; 100 = same size; output = (cols*pct/100, rows*pct/100) */
  int32_t flip;             /* VideoCapture enableFlip: rows reversed (cvFlip mode 0) */
  int32_t roi_x0, roi_y0, roi_x1, roi_y1; /* VideoCapture ROI in the resized, flipped frame; used when x1 > x0 and y1 > y0 */
  int32_t equalize_hist;    /* PreProcessor equalizeHist: 1-channel frames only (cv::equalizeHist asserts CV_8UC1) */
  int32_t gaussian_blur;    /* PreProcessor gaussianBlur */
} bgs_ingest;
int bgs_ingest_default(bgs_ingest* cfg); /* the reference's defaults: 100 %, nothing else */
/* geometry of the prepared frame; BGS_ERR_INVALID for a ROI outside the resized frame or an empty result */
int bgs_ingest_size(const bgs_ingest* cfg, int src_rows, int src_cols, int* rows, int* cols);
/* bytes of device scratch bgs_ingest_device needs for `images` frames (0 when the configuration needs none) */
size_t bgs_ingest_workspace(const bgs_ingest* cfg, int images, int src_rows, int src_cols, int channels);
/* d_src: [images] frames of src_rows x src_cols x channels uint8, rows src_step bytes apart, images src_rows*src_step apart.
 * d_dst: [images][rows][cols][channels] contiguous (bgs_ingest_size) - what bgs_process_batch_device takes.  Asynchronous on
 * hip_stream; d_work may be NULL when bgs_ingest_workspace is 0. */
int bgs_ingest_device(int hip_device, const bgs_ingest* cfg, const void* d_src, int images, int src_rows, int src_cols, int channels,
                      size_t src_step, void* d_dst, void* d_work, void* hip_stream);
/* Make the frame preparation part of the engine's host path: from now on bgs_process takes the RAW captured frame (any size, but
 * the same size every call) and fg / bg come out in the prepared geometry (bgs_ingest_size), which is also the engine's geometry.
 * Flip and ROI alone are folded into the pinned staging copy (no device work at all); resize / equalizeHist / GaussianBlur run
 * as bgs_ingest_device between the upload and the model kernel.  Must be called before the first frame; cfg NULL switches it off. */
int bgs_set_ingest(bgs_engine* e, const bgs_ingest* cfg);
/* Host buffers in, host buffers out (PreProcessor::process / VideoCapture's frame preparation as one call): uploads, runs
 * bgs_ingest_device, downloads.  dst: rows x cols x channels, rows dst_step bytes apart.  Synchronous. */
int bgs_ingest_host(int hip_device, const bgs_ingest* cfg, const uint8_t* src, int src_rows, int src_cols, int channels, size_t src_step,
                    uint8_t* dst, size_t dst_step);

#ifdef __cplusplus
}
#endif
#endif /* BGS_HIP_H */
