// adamw.hip -- the optimizer step of the training loop as ONE multi-tensor launch:
//     torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1);  optimizer.step()        (AdamW)
// (/root/reference/engine.py:99-104, /root/reference/main.py:165 `torch.optim.AdamW`) over the 552
// trainable tensors (47.8 M elements).  The stock sequence is a norm pass, a multiply pass over every
// gradient (the clip) and the fused update; here the clip coefficient is a DEVICE scalar the update
// kernel applies to the gradient as it reads it (one read of g instead of read + write + read), and a
// per-tensor `used` flag skips a tensor entirely -- no weight decay, no moment decay, no step count --
// which is what AdamW does for a parameter whose .grad is None: under the reference's
// DistributedDataParallel(find_unused_parameters=True) (main.py:156) a parameter no rank used keeps
// .grad = None, while the flat-bucket reducer of datr_amd/dist.py gives every parameter a (zero)
// gradient view; the reducer all-reduces the used flags and hands them over here without a host
// synchronisation.
// Arithmetic = torch's single-tensor AdamW, element for element in float32:
//     p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//     p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps),   t = the tensor's own step count.
// A workgroup takes one 8192-element piece of one tensor; `adamw_count` bumps the step counts after.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kPiece = 8192, kThreads = 256;

__device__ __forceinline__ void upd(float &p, float g, float &m, float &v, float coef, float decay, float omb1,
                                    float b2, float omb2, float step_size, float bc2_sqrt, float eps)
{
#pragma clang fp contract(off)
    g *= coef;
    p *= decay;
    m = m + (g - m) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (omb2 * g) * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = fmaf(-step_size, m / denom, p);           // addcdiv_: ATen's kernel contracts a + alpha (b / c) into one fma
}

__global__ __launch_bounds__(kThreads) void adamw_update(const datr_adamw_tensor *__restrict__ tensors,
                                                         const datr_adamw_piece *__restrict__ pieces,
                                                         const float *__restrict__ clip_coef,
                                                         const int *__restrict__ used, double b1d, double b2d, float eps)
{
    const datr_adamw_piece pc = pieces[blockIdx.x];
    const datr_adamw_tensor t = tensors[pc.tensor];
    if (used && t.used_index >= 0 && used[t.used_index] == 0) return;   // used_index < 0: no flag for this tensor, always updated
    const float coef = clip_coef ? *clip_coef : 1.f;
    const double tstep = (double)*t.step + 1.0;
    const double bc1 = 1.0 - pow(b1d, tstep), bc2 = 1.0 - pow(b2d, tstep);
    const float step_size = (float)((double)t.lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float decay = (float)(1.0 - (double)t.lr * (double)t.weight_decay);
    // torch hands `1 - beta` (a Python double) to the float32 tensor ops: round AFTER the subtraction
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    float *p = t.param + pc.offset, *m = t.exp_avg + pc.offset, *v = t.exp_avg_sq + pc.offset;
    const float *g = t.grad + pc.offset;
    const int64_t n = t.numel - pc.offset < kPiece ? t.numel - pc.offset : kPiece;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += kThreads) {
            float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i],
                   vv = reinterpret_cast<float4 *>(v)[i];
            const float4 gg = reinterpret_cast<const float4 *>(g)[i];
            upd(pp.x, gg.x, mm.x, vv.x, coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
            upd(pp.y, gg.y, mm.y, vv.y, coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
            upd(pp.z, gg.z, mm.z, vv.z, coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
            upd(pp.w, gg.w, mm.w, vv.w, coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
            reinterpret_cast<float4 *>(p)[i] = pp;
            reinterpret_cast<float4 *>(m)[i] = mm;
            reinterpret_cast<float4 *>(v)[i] = vv;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads)
            upd(p[i], g[i], m[i], v[i], coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
    } else {
        for (int64_t i = threadIdx.x; i < n; i += kThreads)
            upd(p[i], g[i], m[i], v[i], coef, decay, omb1, b2, omb2, step_size, bc2_sqrt, eps);
    }
}

__global__ void adamw_count(const datr_adamw_tensor *__restrict__ tensors, int ntensors, const int *__restrict__ used)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntensors) return;
    const datr_adamw_tensor t = tensors[i];
    if (used && t.used_index >= 0 && used[t.used_index] == 0) return;   // used_index < 0: no flag for this tensor, always updated
    *t.step += 1.f;
}

// sum of squares of every tensor's gradient: one partial per piece, then one workgroup adds them in a
// fixed order (deterministic) and writes norm and the clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(kThreads) void grad_sq_partial(const datr_adamw_tensor *__restrict__ tensors,
                                                            const datr_adamw_piece *__restrict__ pieces,
                                                            float *__restrict__ partial)
{
    __shared__ float red[kThreads / 64];
    const datr_adamw_piece pc = pieces[blockIdx.x];
    const datr_adamw_tensor t = tensors[pc.tensor];
    const float *g = t.grad + pc.offset;
    const int64_t n = t.numel - pc.offset < kPiece ? t.numel - pc.offset : kPiece;
    float s = 0.f;
    if (((uintptr_t)g & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += kThreads) {
            const float4 v = reinterpret_cast<const float4 *>(g)[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) s += g[i] * g[i];
    } else {
        for (int64_t i = threadIdx.x; i < n; i += kThreads) s += g[i] * g[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void grad_norm_finish(const float *__restrict__ partial, int n, float max_norm,
                                                         float *__restrict__ out /* [2]: norm, coefficient */)
{
    __shared__ double red[1024 / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 1024 / 64; ++i) tot += red[i];
        const float norm = (float)sqrt(tot);
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = c < 1.f ? c : 1.f;                 // NaN norm -> NaN coefficient, as clip_grad_norm_ propagates it
        if (c != c) out[1] = c;
    }
}

}  // namespace

extern "C" int64_t datr_adamw_piece_elements(void) { return kPiece; }

extern "C" int datr_grad_norm_clip_coef_f32(const datr_adamw_tensor *tensors, const datr_adamw_piece *pieces,
                                            int64_t npieces, float max_norm, float *partial, float *norm_coef,
                                            void *stream)
{
    if (!tensors || !pieces || !partial || !norm_coef || npieces <= 0) return DATR_EINVAL;
    if (npieces > 0x7fffffff) return DATR_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(grad_sq_partial, dim3((unsigned)npieces), dim3(kThreads), 0, st, tensors, pieces, partial);
    hipLaunchKernelGGL(grad_norm_finish, dim3(1), dim3(1024), 0, st, partial, (int)npieces, max_norm, norm_coef);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_adamw_step_f32(const datr_adamw_tensor *tensors, int64_t ntensors, const datr_adamw_piece *pieces,
                                   int64_t npieces, const float *clip_coef, const int32_t *used, double beta1,
                                   double beta2, double eps, void *stream)
{
    if (npieces == 0 || ntensors == 0) return DATR_OK;
    if (!tensors || !pieces || npieces < 0 || ntensors < 0) return DATR_EINVAL;
    if (npieces > 0x7fffffff || ntensors > 0x7fffffff) return DATR_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_update, dim3((unsigned)npieces), dim3(kThreads), 0, st, tensors, pieces, clip_coef, used,
                       beta1, beta2, (float)eps);
    hipLaunchKernelGGL(adamw_count, dim3((unsigned)((ntensors + 255) / 256)), dim3(256), 0, st, tensors, (int)ntensors, used);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
