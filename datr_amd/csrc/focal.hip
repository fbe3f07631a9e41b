// focal.hip -- fused sigmoid focal loss (forward + backward) for gfx950.
//
// Reference behaviour: /root/reference/models/dino/utils.py:79-104 (`sigmoid_focal_loss`) as
// called from `SetCriterion.loss_labels` (/root/reference/models/dino/dino.py:508-532):
//     prob = sigmoid(x);  ce = BCEWithLogits(x, t);  p_t = prob*t + (1-prob)*(1-t)
//     loss = ce * (1-p_t)^gamma * (alpha*t + (1-alpha)*(1-t))        (alpha >= 0)
//     loss.mean(1).sum() / num_boxes * num_queries   ==  sum(loss) / num_boxes
// with t the one-hot of the matched class (all-zero row for "no object").  The reference
// materialises the one-hot [B,Q,C+1], slices it, and runs ~12 element-wise kernels plus two
// reductions per call, 13 calls per step.  Here the target is the class INDEX per row
// (index == C means no positive), G independent groups (decoder layers) are handled by one
// launch, and the kernel returns the plain sum per group; the caller applies 1/num_boxes.
//
// Memory-bound, tiny (G*R*C <= ~60 k floats per call): one thread per row, C <= 32 logits in
// registers, block partial sums written to a scratch array and folded per group by a second
// single-wave kernel in a fixed order => deterministic, no float atomics.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float softplus(float x) {       // log(1 + e^x), stable
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

// loss and d loss / d x of one element; pos = (t == 1)
__device__ __forceinline__ void focal_elem(float x, bool pos, float alpha, float gamma,
                                           float &loss, float &grad) {
    const float p = 1.f / (1.f + expf(-x));
    const float log_p = -softplus(-x), log_1mp = -softplus(x);
    const float ce = pos ? -log_p : -log_1mp;
    const float q = pos ? 1.f - p : p;                      // 1 - p_t
    const float a_t = alpha >= 0.f ? (pos ? alpha : 1.f - alpha) : 1.f;
    float mod, dmod;                                         // q^gamma and gamma*q^(gamma-1)
    if (gamma == 2.f) {
        mod = q * q;
        dmod = 2.f * q;
    } else {
        mod = powf(q, gamma);
        dmod = q > 0.f ? gamma * powf(q, gamma - 1.f) : 0.f;
    }
    loss = a_t * ce * mod;
    // d ce/dx = p - t ;  d q/dx = -+ p(1-p)
    const float dce = pos ? p - 1.f : p;
    const float dq = (pos ? -1.f : 1.f) * p * (1.f - p);
    grad = a_t * (dce * mod + ce * dmod * dq);
}

__global__ __launch_bounds__(kThreads) void focal_fwd(
    const float *__restrict__ logits, const int64_t *__restrict__ target, int64_t rows_total,
    int64_t rows_per_group, int C, float alpha, float gamma, int blocks_per_group,
    float *__restrict__ partial)
{
    const int g = blockIdx.x / blocks_per_group;
    const int b = blockIdx.x % blocks_per_group;
    float acc = 0.f;
    for (int64_t r = (int64_t)b * kThreads + threadIdx.x; r < rows_per_group;
         r += (int64_t)blocks_per_group * kThreads) {
        const int64_t row = (int64_t)g * rows_per_group + r;
        const int64_t t = target[row];
        const float *x = logits + row * C;
        for (int c = 0; c < C; ++c) {
            float l, d;
            focal_elem(x[c], c == t, alpha, gamma, l, d);
            acc += l;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float wsum[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}

__global__ void focal_fold(const float *__restrict__ partial, int blocks_per_group,
                           float *__restrict__ out) {
    // one thread per group: fixed summation order
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (int b = 0; b < blocks_per_group; ++b) s += partial[(int64_t)g * blocks_per_group + b];
    out[g] = s;
}

__global__ __launch_bounds__(kThreads) void focal_bwd(
    const float *__restrict__ logits, const int64_t *__restrict__ target,
    const float *__restrict__ grad_sums, int64_t rows_total, int64_t rows_per_group, int C,
    float alpha, float gamma, float *__restrict__ grad_logits)
{
    const int64_t total = rows_total * C;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kThreads) {
        const int64_t row = i / C;
        const int c = (int)(i - row * C);
        float l, d;
        focal_elem(logits[i], c == target[row], alpha, gamma, l, d);
        grad_logits[i] = d * grad_sums[row / rows_per_group];
    }
}

int blocks_for(int64_t rows_per_group) {
    int64_t b = (rows_per_group + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

}  // namespace

extern "C" {

int64_t datr_focal_scratch_floats(int64_t G, int64_t rows_per_group) {
    return G * blocks_for(rows_per_group);
}

int datr_focal_loss_forward_f32(const float *logits, const int64_t *target, int64_t G,
                                int64_t rows_per_group, int64_t C, float alpha, float gamma,
                                float *scratch, float *out_sums, void *stream) {
    if (G < 0 || rows_per_group < 0 || C <= 0) return DATR_EINVAL;
    if (G == 0) return DATR_OK;
    if (!out_sums || !scratch) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows_per_group == 0) {
        return hipMemsetAsync(out_sums, 0, G * sizeof(float), st) == hipSuccess ? DATR_OK : DATR_ELAUNCH;
    }
    if (!logits || !target) return DATR_EINVAL;
    const int bpg = blocks_for(rows_per_group);
    hipLaunchKernelGGL(focal_fwd, dim3((unsigned)(G * bpg)), dim3(kThreads), 0, st, logits, target,
                       G * rows_per_group, rows_per_group, (int)C, alpha, gamma, bpg, scratch);
    hipLaunchKernelGGL(focal_fold, dim3((unsigned)((G + 63) / 64)), dim3(G < 64 ? (unsigned)G : 64u),
                       0, st, scratch, bpg, out_sums);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

int datr_focal_loss_backward_f32(const float *logits, const int64_t *target,
                                 const float *grad_sums, int64_t G, int64_t rows_per_group,
                                 int64_t C, float alpha, float gamma, float *grad_logits,
                                 void *stream) {
    if (G < 0 || rows_per_group < 0 || C <= 0) return DATR_EINVAL;
    const int64_t total = G * rows_per_group * C;
    if (total == 0) return DATR_OK;
    if (!logits || !target || !grad_sums || !grad_logits) return DATR_EINVAL;
    int64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(focal_bwd, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       logits, target, grad_sums, G * rows_per_group, rows_per_group, (int)C, alpha,
                       gamma, grad_logits);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

}  // extern "C"
