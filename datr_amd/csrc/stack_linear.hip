// stack_linear.hip -- the operands of ONE GEMM for two linear layers that read the same input:
//     w [Ra + Rb, C] = [ diag(scale) wa ; wb ],   b [Ra + Rb] = [ scale * ba ; bb ]        (scale may be null)
// MSDeformAttn's `sampling_offsets` and `attention_weights` both project the query
// (/root/reference/models/dino/ops/modules/ms_deform_attn.py:96-97); stacked they are one [384, 256] GEMM, and
// with 2-d reference points the division of the offsets by (W_l, H_l) (:101-104) folds into the rows of the
// offset matrix (scale = 1 / W_l, 1 / H_l per output feature).  The parameters stay separate tensors; this
// builds the stacked operands in one launch per layer and step (torch: two multiplies + two concatenations),
// and the backward scales the gradient rows of the first block in one launch (the rest are row slices).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

__global__ void stack_linear_fwd(const float4 *__restrict__ wa, const float *__restrict__ ba,
                                 const float4 *__restrict__ wb, const float *__restrict__ bb,
                                 const float *__restrict__ scale, int Ra, int Rb, int C4,
                                 float4 *__restrict__ w, float *__restrict__ b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = (Ra + Rb) * C4;
    if (i < nw) {
        const int r = i / C4;
        float4 v;
        if (r < Ra) {
            v = wa[i];
            if (scale) { const float s = scale[r]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
        } else {
            v = wb[i - Ra * C4];
        }
        w[i] = v;
    } else if (i < nw + Ra + Rb) {
        const int r = i - nw;
        b[r] = r < Ra ? (scale ? ba[r] * scale[r] : ba[r]) : bb[r - Ra];
    }
}

__global__ void stack_linear_bwd(const float4 *__restrict__ dw, const float *__restrict__ db,
                                 const float *__restrict__ scale, int Ra, int C4,
                                 float4 *__restrict__ dwa, float *__restrict__ dba)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = Ra * C4;
    if (i < nw) {
        const float s = scale[i / C4];
        float4 v = dw[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        dwa[i] = v;
    } else if (i < nw + Ra) {
        const int r = i - nw;
        dba[r] = db[r] * scale[r];
    }
}

}  // namespace

extern "C" int datr_stack_linear_forward_f32(const float *wa, const float *ba, const float *wb, const float *bb,
                                             const float *scale, int64_t Ra, int64_t Rb, int64_t C, float *w,
                                             float *b, void *stream) {
    if (!wa || !ba || !wb || !bb || !w || !b || Ra <= 0 || Rb <= 0 || C <= 0) return DATR_EINVAL;
    if (C % 4 != 0 || (Ra + Rb) * C > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int total = (int)((Ra + Rb) * (C / 4) + Ra + Rb);
    hipLaunchKernelGGL(stack_linear_fwd, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(wa), ba, reinterpret_cast<const float4 *>(wb), bb, scale,
                       (int)Ra, (int)Rb, (int)(C / 4), reinterpret_cast<float4 *>(w), b);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_stack_linear_backward_f32(const float *d_w, const float *d_b, const float *scale, int64_t Ra,
                                              int64_t C, float *d_wa, float *d_ba, void *stream) {
    if (!d_w || !d_b || !scale || !d_wa || !d_ba || Ra <= 0 || C <= 0) return DATR_EINVAL;
    if (C % 4 != 0 || Ra * C > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int total = (int)(Ra * (C / 4) + Ra);
    hipLaunchKernelGGL(stack_linear_bwd, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(d_w), d_b, scale, (int)Ra, (int)(C / 4),
                       reinterpret_cast<float4 *>(d_wa), d_ba);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
