/*
 * datr_hip_internal.h -- entry points of libdatr_hip.so that are exported for the test-suite and the
 * development tools only; NOT part of the drop-in boundary (include/datr_hip.h) and free to change.
 */
#ifndef DATR_HIP_INTERNAL_H_
#define DATR_HIP_INTERNAL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The phased all-LDS pyramid forward (csrc/msda_fwd_pyr2.hip) called directly, also where the public
 * dispatch of datr_msda_forward_pyramid_f32 would pick another kernel (multi-phase plans): lets
 * tests/test_msda_gpu.py compare every plan shape with the oracle.  Arguments as
 * datr_msda_forward_pyramid_f32 without the device copies of shapes / level_start;
 * envelope_host: float[8][4][4] or NULL.  DATR_EUNSUPPORTED when no window plan exists. */
int datr_internal_msda_fwd_pyr2_d32(const float *value, const float *loc, const float *attn,
                                    const int64_t *shapes_host, const int64_t *level_start_host,
                                    const float *envelope_host, int64_t N, int64_t S, int64_t M, int64_t D,
                                    int64_t L, int64_t Lq, int64_t P, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DATR_HIP_INTERNAL_H_ */
