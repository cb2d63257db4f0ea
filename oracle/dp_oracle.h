/* dp_oracle.h - see dp_oracle.c.  TEST INFRASTRUCTURE ONLY. */
#ifndef DP_ORACLE_H
#define DP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/bgs_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct dp_state dp_state;
int dp_create(bgs_algo algo, const bgs_params* p, const uint8_t* first_frame, int rows, int cols, dp_state** out);
int dp_process(dp_state* s, const uint8_t* img, int64_t frame_num, uint8_t* fg);
int64_t dp_get_state(dp_state* s, const char* plane, void* dst, size_t cap);
void dp_destroy(dp_state* s);
#ifdef __cplusplus
}
#endif
#endif
