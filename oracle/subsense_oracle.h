/* subsense_oracle.h - see subsense_oracle.c.  TEST INFRASTRUCTURE ONLY. */
#ifndef SUBSENSE_ORACLE_H
#define SUBSENSE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/bgs_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ss_state ss_state;
uint32_t ss_rand(uint32_t frame, uint32_t pixel, uint32_t draw);
int ss_create(const bgs_params* p, const uint8_t* first_frame, int rows, int cols, int channels, ss_state** out);
int ss_process(ss_state* s, const uint8_t* img, uint8_t* fg, uint8_t* bg);
int64_t ss_get_state(ss_state* s, const char* plane, void* dst, size_t cap);
void ss_destroy(ss_state* s);
/* LOBSTERBGS (same state type, fewer planes) */
int lob_create(const bgs_params* p, const uint8_t* first_frame, int rows, int cols, int channels, ss_state** out);
int lob_process(ss_state* s, const uint8_t* img, uint8_t* fg, uint8_t* bg);
/* cv::resize(..., INTER_AREA), general (fractional ratio) path: what SuBSENSE's frame-level block uses on sizes that are not multiples of 8 */
void orc_resize_area_u8(const uint8_t* src, int srows, int scols, int ch, uint8_t* dst, int drows, int dcols);
#ifdef __cplusplus
}
#endif
#endif
