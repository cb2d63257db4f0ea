/*
 * bgs_oracle.h — CPU restatement of the reference's package_bgs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tracking_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / the reported CPU baseline.
 *
 * Each function cites the reference file:line it follows.  The arithmetic of
 * cv::BackgroundSubtractorMOG2 / MOG and of every cv:: primitive lives in OpenCV 2.4,
 * which is absent from /root/reference and from this image, so those
 * parts restate OpenCV 2.4's published algorithm from recall:  PARITY UNPINNED for
 * FrameDifference's BGR2GRAY constants, WMM/WMV/ABL/ASBL float rounding, MOG2, MOG1.
 * Pinned parts: LBSP descriptors (checked bit-for-bit against the reference's own
 * LBSP_16bits_dbcross_*.i compiled into oracle/_ref) and SigmaDelta (checked against
 * the reference's own bl/sdLaMa091.cpp compiled into oracle/_ref).
 */
#ifndef BGS_ORACLE_H
#define BGS_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/bgs_hip.h" /* bgs_algo, bgs_params, flag bits: the interface, no code */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_engine orc_engine;

int orc_default_params(bgs_algo algo, bgs_params* p);
int orc_create(bgs_algo algo, const bgs_params* params, orc_engine** out);
int orc_set_params(orc_engine* e, const bgs_params* params);
/* number of OpenMP threads the per-pixel loops use (1 = scalar port; rows are split
 * across threads the way OpenCV's parallel_for_ splits MOG2Invoker over rows) */
int orc_set_threads(orc_engine* e, int n);
int orc_process(orc_engine* e, const uint8_t* in, int rows, int cols, int channels, size_t in_step, uint8_t* fg,
                size_t fg_step, uint8_t* bg, size_t bg_step, uint32_t* out_flags);
/* canonical SoA export, same plane names and order as bgs_get_state */
int64_t orc_get_state(orc_engine* e, const char* plane, void* dst, size_t cap);
void orc_destroy(orc_engine* e);

/* stand-alone primitives */
void orc_bgr2gray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int rows, int cols);
void orc_lbsp_lut(float rel_threshold, int offset, int channels, uint8_t lut[256]);
void orc_lbsp_describe(const uint8_t* img, size_t step, int rows, int cols, int channels, const uint8_t* t_lut,
                       uint16_t* desc /* [rows][cols][channels] */);
void orc_median_blur_u8(const uint8_t* src, uint8_t* dst, int rows, int cols, int ksize);
void orc_erode3x3(const uint8_t* src, uint8_t* dst, int rows, int cols, int iterations);
void orc_dilate3x3(const uint8_t* src, uint8_t* dst, int rows, int cols, int iterations);
void orc_floodfill_from_origin(uint8_t* img, int rows, int cols, uint8_t newval);

/* N3 frame preparation (ingest_oracle.c) */
void orc_ingest_size(const bgs_ingest* c, int src_rows, int src_cols, int* rows, int* cols);
int orc_ingest(const bgs_ingest* c, const uint8_t* src, int src_rows, int src_cols, int ch, size_t src_step, uint8_t* dst);
void orc_resize_linear_u8(const uint8_t* src, int srows, int scols, int ch, size_t sstep, uint8_t* dst, int drows, int dcols);
void orc_equalize_hist_u8(uint8_t* img, size_t total);
int orc_gaussian7_kernel(int ik[7]);
void orc_gaussian_blur7_u8(const uint8_t* src, uint8_t* dst, int rows, int cols, int ch);

#ifdef __cplusplus
}
#endif
#endif
