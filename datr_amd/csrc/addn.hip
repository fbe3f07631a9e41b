// addn.hip -- out = x_0 + x_1 + ... + x_{n-1} (2 <= n <= 8) in ONE pass: the gradient of a tensor that
// several consumers read.  Autograd adds such gradients pairwise (n - 1 launches of 3 passes each over
// the 91 MB encoder token tensor: the sources of an encoder layer feed the value projection, the
// query projection and the residual -- /root/reference/models/dino/deformable_transformer.py:796-806 --
// the position table feeds all six layers, the memory all six decoder layers: 36 adds = 1.4 ms per
// step); datr_amd.fused.fan_out hands every consumer an alias and sums their gradients here
// (n + 1 passes).  The summation order is the argument order: deterministic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

struct AddNArgs { const float4 *x[8]; int n; };

__global__ __launch_bounds__(256) void add_n_kernel(const AddNArgs a, long n4, float4 *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = a.x[0][i];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < a.n) {
                const float4 v = a.x[k][i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        out[i] = s;
    }
}

__global__ __launch_bounds__(256) void add_n_tail(const AddNArgs a, long first, long numel, float *__restrict__ out)
{
    const long i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    float s = reinterpret_cast<const float *>(a.x[0])[i];
    for (int k = 1; k < a.n; ++k) s += reinterpret_cast<const float *>(a.x[k])[i];
    out[i] = s;
}

}  // namespace

extern "C" int datr_add_n_f32(const float *const *xs, int64_t n, int64_t numel, float *out, void *stream)
{
    if (!xs || !out || n < 2 || n > 8 || numel < 0) return DATR_EINVAL;
    AddNArgs a{};
    a.n = (int)n;
    bool aligned = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int k = 0; k < n; ++k) {
        if (!xs[k]) return DATR_EINVAL;
        a.x[k] = reinterpret_cast<const float4 *>(xs[k]);
        aligned = aligned && (reinterpret_cast<uintptr_t>(xs[k]) & 15) == 0;
    }
    if (numel == 0) return DATR_OK;
    hipStream_t st = (hipStream_t)stream;
    const long n4 = aligned ? numel / 4 : 0;
    if (n4 > 0) {
        const long blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(add_n_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, st, a, n4,
                           reinterpret_cast<float4 *>(out));
    }
    const long rest = numel - n4 * 4;
    if (rest > 0)
        hipLaunchKernelGGL(add_n_tail, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, st, a, n4 * 4, (long)numel, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
