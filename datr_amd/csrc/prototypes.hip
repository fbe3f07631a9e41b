// prototypes.hip -- class-wise query prototypes of DATR's prototype alignment
// (/root/reference/models/dino/DA_utils.py:82-120 `get_prototype_class_wise`): given the class label of every
// decoder query (argmax of the predicted scores, computed by the caller), the mean query feature per class, which
// classes are present, and the running count-weighted global prototypes -- the reference's one-hot matrix, its
// [K, R] x [R, 256] product, three `where`s and the blend as ONE launch (about twenty small ATen launches per
// call, two calls per step), and the gradient of the prototypes with respect to the features as one more.
//   count[k]   = #{r : label[r] = k}                      present[k] = count[k] != 0
//   proto[k]   = sum_{label[r] = k} feats[r] / max(count[k], 1)
//   w[k]       = count[k] == 0 ? 0 : count[k] / (count[k] + amount[k])
//   global'[k] = global[k] (1 - w[k]) + proto[k] w[k]       amount'[k] = amount[k] + count[k]
// A workgroup = one class x 64 channels x 16 row slices; the slices meet in LDS in a fixed order (deterministic;
// the reference's GEMM fixes no order).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kSlices = 16;

__global__ __launch_bounds__(64 * kSlices) void prototypes_fwd(
    const float *__restrict__ feats, const int64_t *__restrict__ labels, const float *__restrict__ global_proto,
    const float *__restrict__ amount, int R, int C, int K, float *__restrict__ proto, float *__restrict__ present,
    float *__restrict__ new_global, float *__restrict__ new_amount, float *__restrict__ onehot)
{
    __shared__ float red[kSlices][64];
    __shared__ float cnt[kSlices];
    const int k = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float acc = 0.f, n = 0.f;
    // Eight rows of the slice at a time: their labels, then their feature values, are requested together (row by row,
    // each label decided whether the feature was read at all: two dependent memory latencies per row: 82 -> 57 us for
    // 4 400 rows).  Rows that are not the class's are read at row 0 and dropped by a select; the sums run in the same
    // order as before.  (Measured and not kept: reading every row's value without waiting for its label, 68 us;
    // thirty-two rows at a time spills at 1 024 threads per workgroup, 4 x slower.)
    constexpr int kUnroll = 8;
    for (int r0 = sl; r0 < R; r0 += kSlices * kUnroll) {
        bool mine[kUnroll];
        float v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int r = r0 + u * kSlices;
            const int64_t lab = labels[min(r, R - 1)];        // no branch around the load: eight requests in flight
            mine[u] = (r < R) & (lab == k);                   // uniform over the wave
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = feats[(size_t)(mine[u] ? r0 + u * kSlices : 0) * C + c];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int r = r0 + u * kSlices;
            acc += mine[u] ? v[u] : 0.f;
            n += mine[u] ? 1.f : 0.f;
            if (r < R && blockIdx.y == 0 && (threadIdx.x & 63) == 0) onehot[(size_t)r * K + k] = mine[u] ? 1.f : 0.f;
        }
    }
    red[sl][threadIdx.x & 63] = acc;
    if ((threadIdx.x & 63) == 0) cnt[sl] = n;
    __syncthreads();
    if (sl == 0) {
        float s = 0.f, count = 0.f;
#pragma unroll
        for (int i = 0; i < kSlices; ++i) { s += red[i][threadIdx.x & 63]; count += cnt[i]; }
        const float p = s / (count == 0.f ? 1.f : count);
        const float am = amount[k];
        const float w = count == 0.f ? 0.f : count / (count + am);
        proto[(size_t)k * C + c] = p;
        new_global[(size_t)k * C + c] = global_proto[(size_t)k * C + c] * (1.f - w) + p * w;
        if (blockIdx.y == 0 && threadIdx.x == 0) { present[k] = count != 0.f ? 1.f : 0.f; new_amount[k] = am + count; }
    }
}

// d feats[r] = d proto[label[r]] / max(count[label[r]], 1); count = new_amount - amount is passed as `count`
__global__ __launch_bounds__(256) void prototypes_bwd(const float4 *__restrict__ d_proto, const int64_t *__restrict__ labels,
                                                      const float *__restrict__ count, int R, int C4, int K,
                                                      float4 *__restrict__ d_feats)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C4) return;
    const int r = i / C4, c4 = i - r * C4;
    const int k = (int)labels[r];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= 0 && k < K) {
        const float n = count[k];
        const float inv = 1.f / (n == 0.f ? 1.f : n);
        g = d_proto[(size_t)k * C4 + c4];
        g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
    }
    d_feats[i] = g;
}

}  // namespace

extern "C" int datr_class_prototypes_forward_f32(const float *feats, const int64_t *labels, const float *global_proto,
                                                 const float *amount, int64_t R, int64_t C, int64_t K, float *proto,
                                                 float *present, float *new_global, float *new_amount, float *onehot,
                                                 void *stream) {
    if (!feats || !labels || !global_proto || !amount || !proto || !present || !new_global || !new_amount || !onehot ||
        R < 0 || C <= 0 || K <= 0) return DATR_EINVAL;
    if (C % 64 != 0 || R * C > 0x7fffffffLL || R * K > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(prototypes_fwd, dim3((unsigned)K, (unsigned)(C / 64)), dim3(64 * kSlices), 0, (hipStream_t)stream,
                       feats, labels, global_proto, amount, (int)R, (int)C, (int)K, proto, present, new_global, new_amount,
                       onehot);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_class_prototypes_backward_f32(const float *d_proto, const int64_t *labels, const float *count,
                                                  int64_t R, int64_t C, int64_t K, float *d_feats, void *stream) {
    if (!d_proto || !labels || !count || !d_feats || R < 0 || C <= 0 || K <= 0) return DATR_EINVAL;
    if (C % 4 != 0 || R * C > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    if (R == 0) return DATR_OK;
    const int total = (int)(R * (C / 4));
    hipLaunchKernelGGL(prototypes_bwd, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(d_proto), labels, count, (int)R, (int)(C / 4), (int)K,
                       reinterpret_cast<float4 *>(d_feats));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------------
// Contrastive prototype loss (/root/reference/models/dino/dino.py `loss_contrast_da`: cosine logits of the
// source / target class prototypes against the global prototypes, cross entropy against eye * class_map):
//   n_i = q_i / max(|q_i|, eps),  g^_j = g_j / max(|g_j|, eps),  z_ij = n_i . g^_j,
//   L = (1 / K) sum_i -m_i log_softmax(z_i)_i            summed over the two domains,
// and -- the loss being a scalar -- its gradients with respect to both prototype sets from the same launch:
//   dL/dz_ij = m_i (softmax(z_i)_j - [i = j]) / K,  dL/dn_i = sum_j dL/dz_ij g^_j,
//   dL/dq_i = (dL/dn_i - n_i (n_i . dL/dn_i)) / max(|q_i|, eps).
// One workgroup (K <= 16, C = 256): ~45 small ATen launches forward + backward become one.
namespace {

constexpr int kMaxK = 16;

__device__ __forceinline__ float wave_total(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(1024) void contrast_loss_kernel(const float *__restrict__ qs, const float *__restrict__ qt,
                                                             const float *__restrict__ g, const float *__restrict__ ms,
                                                             const float *__restrict__ mt, int K, float eps,
                                                             float *__restrict__ loss, float *__restrict__ dqs,
                                                             float *__restrict__ dqt)
{
    constexpr int C = 256;
    __shared__ float rows[3 * kMaxK][C];           // q_s, q_t, g -> normalised in place
    __shared__ float inv_norm[3 * kMaxK];
    __shared__ float z[2][kMaxK][kMaxK];
    __shared__ float dz[2][kMaxK][kMaxK];
    __shared__ float part[2 * kMaxK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < 3 * K * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        rows[r][c] = r < K ? qs[r * C + c] : r < 2 * K ? qt[(r - K) * C + c] : g[(r - 2 * K) * C + c];
    }
    __syncthreads();
    for (int r = wave; r < 3 * K; r += nw) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += rows[r][c] * rows[r][c];
        s = wave_total(s);
        if (lane == 0) inv_norm[r] = 1.f / fmaxf(sqrtf(s), eps);
    }
    __syncthreads();
    for (int i = tid; i < 3 * K * C; i += blockDim.x) rows[i / C][i % C] *= inv_norm[i / C];
    __syncthreads();
    for (int p = wave; p < 2 * K * K; p += nw) {               // z[d][i][j] = n_i . g^_j
        const int d = p / (K * K), i = (p / K) % K, j = p % K;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += rows[d * K + i][c] * rows[2 * K + j][c];
        s = wave_total(s);
        if (lane == 0) z[d][i][j] = s;
    }
    __syncthreads();
    if (tid < 2 * K) {                                         // one row of one domain per thread
        const int d = tid / K, i = tid % K;
        const float m = d == 0 ? ms[i] : mt[i];
        float mx = -INFINITY;
        for (int j = 0; j < K; ++j) mx = fmaxf(mx, z[d][i][j]);
        float se = 0.f;
        for (int j = 0; j < K; ++j) se += expf(z[d][i][j] - mx);
        const float lse = mx + logf(se);
        part[tid] = -m * (z[d][i][i] - lse) / (float)K;
        for (int j = 0; j < K; ++j) dz[d][i][j] = m * (expf(z[d][i][j] - lse) - (i == j ? 1.f : 0.f)) / (float)K;
    }
    __syncthreads();
    if (tid == 0) {
        float s0 = 0.f, s1 = 0.f;                               // ce(source) + ce(target), each a mean over its rows
        for (int i = 0; i < K; ++i) { s0 += part[i]; s1 += part[K + i]; }
        loss[0] = s0 + s1;
    }
    for (int r = wave; r < 2 * K; r += nw) {                   // gradient of row r (domain d, class i)
        const int d = r / K, i = r % K;
        float dn[4], dot = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            float s = 0.f;
            for (int j = 0; j < K; ++j) s += dz[d][i][j] * rows[2 * K + j][c];
            dn[u] = s;
            dot += s * rows[r][c];
        }
        dot = wave_total(dot);
        float *out = d == 0 ? dqs : dqt;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            out[i * C + c] = (dn[u] - rows[r][c] * dot) * inv_norm[r];
        }
    }
}

}  // namespace

extern "C" int datr_contrast_loss_f32(const float *q_source, const float *q_target, const float *global_proto,
                                      const float *mask_source, const float *mask_target, int64_t K, int64_t C,
                                      float eps, float *loss, float *d_q_source, float *d_q_target, void *stream) {
    if (!q_source || !q_target || !global_proto || !mask_source || !mask_target || !loss || !d_q_source || !d_q_target)
        return DATR_EINVAL;
    if (K < 1 || K > kMaxK || C != 256) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(contrast_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, q_source, q_target, global_proto,
                       mask_source, mask_target, (int)K, eps, loss, d_q_source, d_q_target);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
