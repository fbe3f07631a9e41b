// layernorm.hip -- residual add + LayerNorm over d_model = 256 channels, forward and backward, one
// HBM pass each.
//
// Post-norm transformer layers of the reference end every sub-block with
// `x = norm(x + dropout(branch))` (/root/reference/models/dino/deformable_transformer.py:796-806
// encoder, :856-893 decoder; dropout = 0 in every DA config).  ATen runs an add kernel, the
// LayerNorm forward, and in backward one kernel for d input plus two for d gamma / d beta that
// re-read the same rows.  Here:
//   forward   s = x + res;  y = (s - mean) * rstd * gamma + beta          read x, res; write y
//   backward  xhat from x + res (recomputed), dx = d(x) = d(res),          read dy, x, res; write dx
//             d gamma / d beta accumulated in registers by each wave over its rows and reduced
//             through per-workgroup partials (deterministic, no atomics)
// One wave per row: lane l holds channels 4l..4l+3 as a float4 (64 x 4 = 256); mean and variance
// are two-pass in registers (sum, then sum of squared deviations).  Traffic per row of 1 KiB:
// forward 3 KiB instead of 5, backward 4 KiB instead of 5.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kC = 256;
constexpr int kThreads = 256;                 // 4 waves = 4 rows in flight per workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kBwdThreads = 512;              // backward: 8 rows in flight per workgroup,
constexpr int kBwdWaves = kBwdThreads / 64;   // few workgroups => few gamma/beta partials
constexpr int kBwdMaxGrid = 512;
constexpr int kFinThreads = 1024;
constexpr int kFinWaves = kFinThreads / 64;

__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x141, 0xF, 0xF, false);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x140, 0xF, 0xF, false);    // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void add_ln_fwd_kernel(
    const float4 *__restrict__ x, const float4 *__restrict__ res, const float4 *__restrict__ gamma,
    const float4 *__restrict__ beta, int64_t rows, float eps, float4 *__restrict__ y,
    float *__restrict__ mean, float *__restrict__ rstd,
    const float4 *__restrict__ add /* may be null */, float4 *__restrict__ y2 /* y + add */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g = gamma[lane], b = beta[lane];
    for (int64_t r = (int64_t)blockIdx.x * kWaves + wave; r < rows; r += (int64_t)gridDim.x * kWaves) {
        float4 s = x[r * 64 + lane];
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add) p = add[r * 64 + lane];
        if (res) {
            const float4 t = res[r * 64 + lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float mu = wave_sum(s.x + s.y + s.z + s.w) * (1.f / kC);
        const float dx = s.x - mu, dy = s.y - mu, dz = s.z - mu, dw = s.w - mu;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / kC);
        const float rs = rsqrtf(var + eps);
        float4 o;
        o.x = dx * rs * g.x + b.x; o.y = dy * rs * g.y + b.y;
        o.z = dz * rs * g.z + b.z; o.w = dw * rs * g.w + b.w;
        y[r * 64 + lane] = o;
        if (add) y2[r * 64 + lane] = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

// score[r] = max_c ( LayerNorm(x[r]) . W[c] + b[c] ): the two-stage query selection's class score of every
// encoder token (/root/reference/models/dino/deformable_transformer.py:335-342: enc_output_norm, the class
// head, `.max(-1)[0]` feeding top-k) without writing the normalised rows or the [rows, C] logits -- in
// training nothing else reads them (the selected rows are re-projected with autograd).  One wave per row as
// above; the class weights sit in registers (nc <= kMaxClasses float4 per lane).
constexpr int kMaxClasses = 16;
__global__ __launch_bounds__(kThreads) void ln_class_max_kernel(
    const float4 *__restrict__ x, const float4 *__restrict__ gamma, const float4 *__restrict__ beta,
    const float4 *__restrict__ W, const float *__restrict__ bias, int nc, int64_t rows, float eps,
    const unsigned char *__restrict__ row_mask, const float4 *__restrict__ row_fill, float *__restrict__ score)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g = gamma[lane], b = beta[lane];
    float4 w[kMaxClasses];
    float bc[kMaxClasses];
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
        w[c] = c < nc ? W[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        bc[c] = c < nc ? bias[c] : 0.f;
    }
    for (int64_t r = (int64_t)blockIdx.x * kWaves + wave; r < rows; r += (int64_t)gridDim.x * kWaves) {
        // a masked row stands for x = row_fill (the projection of a zeroed token is its bias)
        const float4 s = (row_mask && row_mask[r]) ? row_fill[lane] : x[r * 64 + lane];
        const float mu = wave_sum(s.x + s.y + s.z + s.w) * (1.f / kC);
        const float dx = s.x - mu, dy = s.y - mu, dz = s.z - mu, dw = s.w - mu;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / kC);
        const float rs = rsqrtf(var + eps);
        const float ox = dx * rs * g.x + b.x, oy = dy * rs * g.y + b.y, oz = dz * rs * g.z + b.z, ow = dw * rs * g.w + b.w;
        float best = -INFINITY;
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < nc) {                          // wave-uniform
                const float p = wave_sum(ox * w[c].x + oy * w[c].y + oz * w[c].z + ow * w[c].w) + bc[c];
                best = fmaxf(best, p);
            }
        }
        if (lane == 0) score[r] = best;
    }
}

// NS = 2: partial sums of d gamma, d beta.  NS = 3: also the column sums of dx -- when the normalised tensor is
// x + linear(h), they are that linear layer's bias gradient (the FFN sub-block's db2: no separate pass over dx).
template <int NS>
__global__ __launch_bounds__(kBwdThreads) void add_ln_bwd_kernel(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float4 *__restrict__ res,
    const float *__restrict__ mean, const float *__restrict__ rstd, const float4 *__restrict__ gamma,
    int64_t rows, float4 *__restrict__ dx, float4 *__restrict__ partial /* [grid][NS][64] float4 */,
    const float4 *__restrict__ dy1 /* may be null */, const float4 *__restrict__ dy2 /* may be null */)
{
    __shared__ float4 red[kBwdWaves][NS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g = gamma[lane];
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg, ds = dg;
    for (int64_t r = (int64_t)blockIdx.x * kBwdWaves + wave; r < rows; r += (int64_t)gridDim.x * kBwdWaves) {
        float4 s = x[r * 64 + lane];
        if (res) {
            const float4 t = res[r * 64 + lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4 d = dy[r * 64 + lane];
        if (dy1) {                                // the gradients of y's other consumers, summed on load: (dy + dy1) + dy2
            const float4 t = dy1[r * 64 + lane];
            d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        }
        if (dy2) {
            const float4 t = dy2[r * 64 + lane];
            d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        }
        const float mu = mean[r], rs = rstd[r];
        const float4 xh = make_float4((s.x - mu) * rs, (s.y - mu) * rs, (s.z - mu) * rs, (s.w - mu) * rs);
        const float4 dgm = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        const float c1 = wave_sum(dgm.x + dgm.y + dgm.z + dgm.w) * (1.f / kC);
        const float c2 = wave_sum(dgm.x * xh.x + dgm.y * xh.y + dgm.z * xh.z + dgm.w * xh.w) * (1.f / kC);
        float4 o;
        o.x = (dgm.x - c1 - xh.x * c2) * rs; o.y = (dgm.y - c1 - xh.y * c2) * rs;
        o.z = (dgm.z - c1 - xh.z * c2) * rs; o.w = (dgm.w - c1 - xh.w * c2) * rs;
        dx[r * 64 + lane] = o;
        dg.x += d.x * xh.x; dg.y += d.y * xh.y; dg.z += d.z * xh.z; dg.w += d.w * xh.w;
        db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
        if (NS == 3) { ds.x += o.x; ds.y += o.y; ds.z += o.z; ds.w += o.w; }
    }
    red[wave][0][lane] = dg;
    red[wave][1][lane] = db;
    if (NS == 3) red[wave][2][lane] = ds;
    __syncthreads();
    if (wave < NS) {                              // wave 0 folds d gamma, wave 1 d beta, wave 2 the sums of dx
        float4 t = red[0][wave][lane];
        for (int w = 1; w < kBwdWaves; ++w) {
            const float4 u = red[w][wave][lane];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        partial[((int64_t)blockIdx.x * NS + wave) * 64 + lane] = t;
    }
}

// partial [nblk][2][64] float4 -> dgamma [64] float4, dbeta [64] float4.  One workgroup per output,
// 16 waves share the partial rows (unrolled independent loads), then meet in LDS.
__global__ __launch_bounds__(kFinThreads) void add_ln_finish_kernel(
    const float4 *__restrict__ partial, int nblk, int ns, float4 *__restrict__ dgamma, float4 *__restrict__ dbeta,
    float4 *__restrict__ dxsum)
{
    __shared__ float4 red[kFinWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int which = blockIdx.x;                 // 0 = gamma, 1 = beta, 2 = column sums of dx
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // 8 independent loads in flight per wave, summed in a fixed order
    for (int b0 = wave; b0 < nblk; b0 += kFinWaves * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * kFinWaves;
            v[u] = b < nblk ? partial[((int64_t)b * ns + which) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        float4 t = red[0][lane];
        for (int w = 1; w < kFinWaves; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
        (which == 0 ? dgamma : which == 1 ? dbeta : dxsum)[lane] = t;
    }
}

int grid_for(int64_t rows) {
    const int64_t want = (rows + kWaves - 1) / kWaves;
    return (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
}

int bwd_grid_for(int64_t rows) {
    const int64_t want = (rows + kBwdWaves - 1) / kBwdWaves;
    return (int)(want < kBwdMaxGrid ? (want < 1 ? 1 : want) : kBwdMaxGrid);
}

}  // namespace

extern "C" int64_t datr_add_layernorm_partial_floats(int64_t rows) {
    return (int64_t)bwd_grid_for(rows) * 3 * kC;
}

static int add_layernorm_forward(const float *x, const float *res, const float *gamma, const float *beta,
                                 int64_t rows, int64_t C, float eps, float *y, float *mean, float *rstd,
                                 const float *add, float *y2, void *stream) {
    if (rows < 0 || C != kC) return C != kC ? DATR_EUNSUPPORTED : DATR_EINVAL;
    if (rows == 0) return DATR_OK;
    if (!x || !gamma || !beta || !y || !mean || !rstd || (add && !y2)) return DATR_EINVAL;
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3((unsigned)grid_for(rows)), dim3(kThreads), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(x),
                       reinterpret_cast<const float4 *>(res), reinterpret_cast<const float4 *>(gamma),
                       reinterpret_cast<const float4 *>(beta), rows, eps, reinterpret_cast<float4 *>(y),
                       mean, rstd, reinterpret_cast<const float4 *>(add), reinterpret_cast<float4 *>(y2));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_add_layernorm_forward_f32(const float *x, const float *res, const float *gamma,
                                              const float *beta, int64_t rows, int64_t C, float eps,
                                              float *y, float *mean, float *rstd, void *stream) {
    return add_layernorm_forward(x, res, gamma, beta, rows, C, eps, y, mean, rstd, nullptr, nullptr, stream);
}

// The same, and y2 = y + add in the same pass: an encoder layer's output together with the next layer's query
// (tokens + position table, /root/reference/models/dino/deformable_transformer.py:789-798 `with_pos_embed`).
extern "C" int datr_add_layernorm_forward_query_f32(const float *x, const float *res, const float *gamma,
                                                    const float *beta, const float *add, int64_t rows, int64_t C,
                                                    float eps, float *y, float *y2, float *mean, float *rstd,
                                                    void *stream) {
    if (!add || !y2) return DATR_EINVAL;
    return add_layernorm_forward(x, res, gamma, beta, rows, C, eps, y, mean, rstd, add, y2, stream);
}

extern "C" int datr_layernorm_class_max_f32(const float *x, const float *gamma, const float *beta, const float *w,
                                            const float *bias, int64_t rows, int64_t C, int64_t classes, float eps,
                                            const uint8_t *row_mask, const float *row_fill, float *score,
                                            void *stream) {
    if (rows < 0 || C != kC || classes < 1 || classes > kMaxClasses) return rows < 0 ? DATR_EINVAL : DATR_EUNSUPPORTED;
    if (rows == 0) return DATR_OK;
    if (!x || !gamma || !beta || !w || !bias || !score || (row_mask && !row_fill)) return DATR_EINVAL;
    hipLaunchKernelGGL(ln_class_max_kernel, dim3((unsigned)grid_for(rows)), dim3(kThreads), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(gamma),
                       reinterpret_cast<const float4 *>(beta), reinterpret_cast<const float4 *>(w), bias, (int)classes,
                       rows, eps, row_mask, reinterpret_cast<const float4 *>(row_fill), score);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

static int add_layernorm_backward(const float *dy, const float *x, const float *res, const float *mean,
                                  const float *rstd, const float *gamma, int64_t rows, int64_t C, float *dx,
                                  float *partial, float *dgamma, float *dbeta, float *dxsum, void *stream,
                                  const float *dy1 = nullptr, const float *dy2 = nullptr) {
    if (rows < 0 || C != kC) return C != kC ? DATR_EUNSUPPORTED : DATR_EINVAL;
    if (!dgamma || !dbeta) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        if (hipMemsetAsync(dgamma, 0, kC * 4, st) != hipSuccess || hipMemsetAsync(dbeta, 0, kC * 4, st) != hipSuccess ||
            (dxsum && hipMemsetAsync(dxsum, 0, kC * 4, st) != hipSuccess))
            return DATR_ELAUNCH;
        return DATR_OK;
    }
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !partial) return DATR_EINVAL;
    const int nblk = bwd_grid_for(rows);
    const int ns = dxsum ? 3 : 2;
    if (dxsum)
        hipLaunchKernelGGL(add_ln_bwd_kernel<3>, dim3((unsigned)nblk), dim3(kBwdThreads), 0, st,
                           reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x),
                           reinterpret_cast<const float4 *>(res), mean, rstd,
                           reinterpret_cast<const float4 *>(gamma), rows, reinterpret_cast<float4 *>(dx),
                           reinterpret_cast<float4 *>(partial), reinterpret_cast<const float4 *>(dy1),
                           reinterpret_cast<const float4 *>(dy2));
    else
        hipLaunchKernelGGL(add_ln_bwd_kernel<2>, dim3((unsigned)nblk), dim3(kBwdThreads), 0, st,
                           reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x),
                           reinterpret_cast<const float4 *>(res), mean, rstd,
                           reinterpret_cast<const float4 *>(gamma), rows, reinterpret_cast<float4 *>(dx),
                           reinterpret_cast<float4 *>(partial), reinterpret_cast<const float4 *>(dy1),
                           reinterpret_cast<const float4 *>(dy2));
    hipLaunchKernelGGL(add_ln_finish_kernel, dim3((unsigned)ns), dim3(kFinThreads), 0, st,
                       reinterpret_cast<const float4 *>(partial), nblk, ns,
                       reinterpret_cast<float4 *>(dgamma), reinterpret_cast<float4 *>(dbeta),
                       reinterpret_cast<float4 *>(dxsum));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_add_layernorm_backward_f32(const float *dy, const float *x, const float *res,
                                               const float *mean, const float *rstd,
                                               const float *gamma, int64_t rows, int64_t C, float *dx,
                                               float *partial, float *dgamma, float *dbeta,
                                               void *stream) {
    return add_layernorm_backward(dy, x, res, mean, rstd, gamma, rows, C, dx, partial, dgamma, dbeta, nullptr, stream);
}

// The same, and dxsum[c] = sum over rows of dx[r][c] (deterministic: fixed-order partial sums): when the normalised
// tensor is x + linear(h) this is that linear layer's bias gradient.
extern "C" int datr_add_layernorm_backward_colsum_f32(const float *dy, const float *x, const float *res,
                                                      const float *mean, const float *rstd,
                                                      const float *gamma, int64_t rows, int64_t C, float *dx,
                                                      float *partial, float *dgamma, float *dbeta, float *dxsum,
                                                      void *stream) {
    if (!dxsum) return DATR_EINVAL;
    return add_layernorm_backward(dy, x, res, mean, rstd, gamma, rows, C, dx, partial, dgamma, dbeta, dxsum, stream);
}

// The same with up to three gradients of the normalised tensor, summed on load as (dy + dy1) + dy2 (dy1 / dy2 may be
// null): the consumers of an encoder layer's output -- residual, value projection, next query -- hand their
// gradients over without a separate sum over the token tensor.  dxsum may be null (no column sums).
extern "C" int datr_add_layernorm_backward_fanin_f32(const float *dy, const float *dy1, const float *dy2,
                                                     const float *x, const float *res, const float *mean,
                                                     const float *rstd, const float *gamma, int64_t rows, int64_t C,
                                                     float *dx, float *partial, float *dgamma, float *dbeta,
                                                     float *dxsum, void *stream) {
    if (!dy1 && dy2) return DATR_EINVAL;
    return add_layernorm_backward(dy, x, res, mean, rstd, gamma, rows, C, dx, partial, dgamma, dbeta, dxsum, stream,
                                  dy1, dy2);
}
