// refine.hip -- iterative box refinement of the DINO decoder as one element-wise launch each way:
//     new_ref = sigmoid(delta + inverse_sigmoid(ref))
// (/root/reference/models/dino/deformable_transformer.py:738-744 per decoder layer and
// /root/reference/models/dino/dino.py:316-322 over the stacked layers, with
// inverse_sigmoid(x) = log(clamp(x, 0, 1).clamp(min = 1e-3) / (1 - clamp(x, 0, 1)).clamp(min = 1e-3)),
// /root/reference/util/misc.py:587-591): eight ATen launches forward and as many backward per call on
// [1100, 4, 4]-sized tensors become one.  Arithmetic follows the op sequence (no contraction):
// backward uses autograd's conventions -- sigmoid' = y (1 - y), clamp passes the gradient where the
// input lies inside the closed range.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void refine_fwd(const float *__restrict__ delta, const float *__restrict__ ref,
                                                  long n, float eps, float *__restrict__ out)
{
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = ref[i];
    const float x = fminf(fmaxf(r, 0.f), 1.f);
    const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
    const float z = delta[i] + logf(x1 / x2);
    out[i] = 1.f / (1.f + expf(-z));
}

__global__ __launch_bounds__(256) void refine_bwd(const float *__restrict__ g, const float *__restrict__ out,
                                                  const float *__restrict__ ref, long n, float eps,
                                                  float *__restrict__ d_delta, float *__restrict__ d_ref)
{
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float y = out[i];
    const float gz = g[i] * (1.f - y) * y;                       // sigmoid_backward
    if (d_delta) d_delta[i] = gz;
    if (d_ref) {
        const float r = ref[i];
        const float x = fminf(fmaxf(r, 0.f), 1.f);
        const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
        // log(x1 / x2): d/dx1 = 1 / x1, d/dx2 = -1 / x2; x2 = clamp_min(1 - x): d x2 / dx = -[1 - x >= eps]
        const float gx = (x >= eps ? gz / x1 : 0.f) + ((1.f - x) >= eps ? gz / x2 : 0.f);
        d_ref[i] = (r >= 0.f && r <= 1.f) ? gx : 0.f;
    }
}

}  // namespace

extern "C" int datr_refine_boxes_forward_f32(const float *delta, const float *ref, int64_t n, float eps, float *out,
                                             void *stream)
{
    if (n == 0) return DATR_OK;
    if (!delta || !ref || !out || n < 0) return DATR_EINVAL;
    hipLaunchKernelGGL(refine_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       delta, ref, (long)n, eps, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_refine_boxes_backward_f32(const float *grad_out, const float *out, const float *ref, int64_t n,
                                              float eps, float *grad_delta, float *grad_ref, void *stream)
{
    if (n == 0) return DATR_OK;
    if (!grad_out || !out || !ref || n < 0 || (!grad_delta && !grad_ref)) return DATR_EINVAL;
    hipLaunchKernelGGL(refine_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       grad_out, out, ref, (long)n, eps, grad_delta, grad_ref);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
