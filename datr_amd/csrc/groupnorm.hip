// groupnorm.hip -- GroupNorm on NHWC (torch.channels_last) tensors, forward and backward: the
// `nn.GroupNorm(32, 256)` that follows every input_proj convolution of DINO
// (/root/reference/models/dino/dino.py:111-126).  The backbone and the projections run in NHWC
// (MIOpen's fastest fp32 layout); ATen's group norm transposes such an input to NCHW, normalises
// there and hands NCHW on, so the [N, HW, C] token layout the transformer wants and the NHWC input
// the discriminator kernels want cost two more transposes of every level (214 us + 96 us + the
// strided gather inside the level concat at the 100 x 167 level).  These kernels keep the tensor
// in NHWC end to end: y[n, p, c] = (x[n, p, c] - mean[n, g]) * rstd[n, g] * gamma[c] + beta[c],
// g = c / (C / G), statistics over the HW * C/G elements of a (sample, group), biased variance,
// eps inside the rsqrt (torch.nn.functional.group_norm).
//
// Forward: `gn_stats` -- a 1 024-thread workgroup walks a slab of pixels, thread = (float4 column of
// the C/4 columns, pixel phase); shifted sums (x - k, k = the group's first element of the sample:
// no cancellation when |mean| >> std) reduced over the workgroup, one partial per (sample, slab,
// group); `gn_finish` (one workgroup per sample) sums the <= 64 partials in a fixed order (deterministic, no
// atomics) into mean / rstd, `gn_apply` normalises its slab: 2 reads + 1 write of the tensor in all.
// Backward: `gn_bwd_stats` accumulates per channel sum(dy) and sum(dy * xhat) per (sample, slab);
// `gn_bwd_finish` (one workgroup per sample) turns them into the two group sums, `gn_bwd_apply` writes
//   dx = rstd * (dy * gamma - (s1 + xhat * s2) / m),  s1 = sum_g dy * gamma, s2 = sum_g dy * gamma * xhat,
// and (one workgroup) dgamma = sum_n sum(dy * xhat), dbeta = sum_n sum(dy).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;                  // the apply kernels
constexpr int kStatThreads = 1024;             // the statistics kernels: <= 64 slabs x N workgroups, sixteen waves each
constexpr int kMaxSlabs = 64;
constexpr int kMaxC = 1024;

__device__ __forceinline__ int slab_begin(int s, int slabs, int HW) { return (int)((long)HW * s / slabs); }

// partial[((n * slabs + s) * G + g) * 2 + {0, 1}] = sum(x - k), sum((x - k)^2) over the slab
__global__ __launch_bounds__(kStatThreads) void gn_stats(const float *__restrict__ x, int HW, int C, int G, int slabs,
                                                     float *__restrict__ partial)
{
    __shared__ float red[2][kStatThreads];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cols = C >> 2, cpg4 = (C / G) >> 2;              // float4 columns, columns per group
    const int rows_per_pass = kStatThreads / cols;             // pixels covered by one pass of the block
    const int col = tid % cols, prow = tid / cols;
    const float *xn = x + (size_t)n * HW * C;
    const float k = xn[(col / cpg4) * (C / G)];                // shift: the group's first element of pixel 0
    float s1 = 0.f, s2 = 0.f;
    if (prow < rows_per_pass) {
        const int p1 = slab_begin(s + 1, slabs, HW);
        for (int p = slab_begin(s, slabs, HW) + prow; p < p1; p += rows_per_pass) {
            const float4 v = *reinterpret_cast<const float4 *>(xn + (size_t)p * C + col * 4);
            const float a = v.x - k, b = v.y - k, c = v.z - k, d = v.w - k;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    if (tid < G) {                                             // fixed-order sum over the group's threads
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < rows_per_pass; ++r)
            for (int c = 0; c < cpg4; ++c) {
                const int t = r * cols + tid * cpg4 + c;
                t1 += red[0][t]; t2 += red[1][t];
            }
        float *o = partial + (((size_t)n * slabs + s) * G + tid) * 2;
        o[0] = t1; o[1] = t2;
    }
}

// mean / rstd of every (sample, group) from the slab partials, summed in a fixed order -- ONCE per sample (round 5:
// every one of the <= 256 workgroups of gn_apply re-reduced them in its prologue, 128 dependent loads in front of
// 130 KB of work).  Grid = N, block = G threads (<= 256).
__global__ __launch_bounds__(kThreads) void gn_finish(const float *__restrict__ x, const float *__restrict__ partial,
                                                      int HW, int C, int G, int stat_slabs, float eps,
                                                      float *__restrict__ mean, float *__restrict__ rstd)
{
    __shared__ float r1[kThreads], r2[kThreads];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int P = kThreads / G, g = tid % G, ph = tid / G;     // P phases walk the slabs P apart, then meet in order
    float t1 = 0.f, t2 = 0.f;
    if (ph < P)
        for (int b = ph; b < stat_slabs; b += P) {
            const float *o = partial + (((size_t)n * stat_slabs + b) * G + g) * 2;
            t1 += o[0]; t2 += o[1];
        }
    r1[tid] = t1; r2[tid] = t2;
    __syncthreads();
    if (tid >= G) return;
    t1 = r1[tid]; t2 = r2[tid];
    for (int q = 1; q < P; ++q) { t1 += r1[q * G + tid]; t2 += r2[q * G + tid]; }
    const int cpg = C / G;
    const float inv_m = 1.f / ((float)HW * (float)cpg);
    const float d = t1 * inv_m;                                // mean - k
    const float var = fmaxf(t2 * inv_m - d * d, 0.f);
    mean[n * G + tid] = x[(size_t)n * HW * C + tid * cpg] + d;
    rstd[n * G + tid] = rsqrtf(var + eps);
}

__global__ __launch_bounds__(kThreads) void gn_apply(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, int HW, int C, int G,
                                                     int slabs, float *__restrict__ y)
{
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cols = C >> 2, cpg = C / G, cpg4 = cpg >> 2;
    const float *xn = x + (size_t)n * HW * C;
    const int rows_per_pass = kThreads / cols;
    const int col = tid % cols, prow = tid / cols;
    if (prow >= rows_per_pass) return;
    const float mu = mean[n * G + col / cpg4], rs = rstd[n * G + col / cpg4];
    const float4 g4 = reinterpret_cast<const float4 *>(gamma)[col], b4 = reinterpret_cast<const float4 *>(beta)[col];
    const float4 a4 = make_float4(rs * g4.x, rs * g4.y, rs * g4.z, rs * g4.w);
    const float4 c4 = make_float4(b4.x - mu * a4.x, b4.y - mu * a4.y, b4.z - mu * a4.z, b4.w - mu * a4.w);
    float *yn = y + (size_t)n * HW * C;
    const int p1 = slab_begin(s + 1, slabs, HW);
    for (int p = slab_begin(s, slabs, HW) + prow; p < p1; p += rows_per_pass) {
        const float4 v = *reinterpret_cast<const float4 *>(xn + (size_t)p * C + col * 4);
        *reinterpret_cast<float4 *>(yn + (size_t)p * C + col * 4) =
            make_float4(fmaf(v.x, a4.x, c4.x), fmaf(v.y, a4.y, c4.y), fmaf(v.z, a4.z, c4.z), fmaf(v.w, a4.w, c4.w));
    }
}

// partial[((n * slabs + s) * 2 + {0: sum dy, 1: sum dy * xhat}) * C + c]
__global__ __launch_bounds__(kStatThreads) void gn_bwd_stats(const float *__restrict__ dy, const float *__restrict__ x,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, int HW, int C, int G,
                                                         int slabs, float *__restrict__ partial)
{
    __shared__ float4 red[2][kStatThreads];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cols = C >> 2, cpg4 = (C / G) >> 2;
    const int rows_per_pass = kStatThreads / cols;
    const int col = tid % cols, prow = tid / cols;
    const float mu = mean[n * G + col / cpg4], rs = rstd[n * G + col / cpg4];
    const float *xn = x + (size_t)n * HW * C, *dn = dy + (size_t)n * HW * C;
    float4 db = make_float4(0.f, 0.f, 0.f, 0.f), dg = db;
    if (prow < rows_per_pass) {
        const int p1 = slab_begin(s + 1, slabs, HW);
        for (int p = slab_begin(s, slabs, HW) + prow; p < p1; p += rows_per_pass) {
            const float4 v = *reinterpret_cast<const float4 *>(xn + (size_t)p * C + col * 4);
            const float4 g = *reinterpret_cast<const float4 *>(dn + (size_t)p * C + col * 4);
            db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
            dg.x = fmaf(g.x, (v.x - mu) * rs, dg.x); dg.y = fmaf(g.y, (v.y - mu) * rs, dg.y);
            dg.z = fmaf(g.z, (v.z - mu) * rs, dg.z); dg.w = fmaf(g.w, (v.w - mu) * rs, dg.w);
        }
    }
    red[0][tid] = db; red[1][tid] = dg;
    __syncthreads();
    if (tid < cols) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int r = 0; r < rows_per_pass; ++r) {
            const float4 u = red[0][r * cols + tid], w = red[1][r * cols + tid];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        float *o = partial + ((size_t)n * slabs + s) * 2 * C;
        reinterpret_cast<float4 *>(o)[tid] = a;
        reinterpret_cast<float4 *>(o + C)[tid] = b;
    }
}

// Per sample: the per-channel sums of the slab partials (fixed order) -> chan[n][2][C], and the two sums of every group
// -> gsum[n][G][2].  Grid = N.  Round 5: gn_bwd_apply did this in the prologue of every workgroup, and its
// (sample 0, slab 0) workgroup walked ALL samples' partials for the parameter gradients -- 512 dependent loads per
// thread, the launch's long pole (92 us for 205 MB at the 100 x 167 level).
__global__ __launch_bounds__(kStatThreads) void gn_bwd_finish(const float *__restrict__ partial,
                                                              const float *__restrict__ gamma, int C, int G,
                                                              int stat_slabs, float *__restrict__ chan,
                                                              float *__restrict__ gsum)
{
    __shared__ float ra[kStatThreads], rb[kStatThreads], sdb[kMaxC], sdg[kMaxC];
    const int n = blockIdx.x, tid = threadIdx.x, cpg = C / G;
    const int P = kStatThreads / C, c = tid % C, ph = tid / C;  // P phases walk the slabs P apart, then meet in order
    float a = 0.f, b = 0.f;
    if (ph < P)
        for (int q = ph; q < stat_slabs; q += P) {
            const float *o = partial + ((size_t)n * stat_slabs + q) * 2 * C;
            a += o[c]; b += o[C + c];
        }
    ra[tid] = a; rb[tid] = b;
    __syncthreads();
    if (tid < C) {
        a = ra[tid]; b = rb[tid];
        for (int q = 1; q < P; ++q) { a += ra[q * C + tid]; b += rb[q * C + tid]; }
        sdb[tid] = a; sdg[tid] = b;
        chan[((size_t)n * 2) * C + tid] = a;
        chan[((size_t)n * 2 + 1) * C + tid] = b;
    }
    __syncthreads();
    if (tid < G) {
        a = 0.f; b = 0.f;
        for (int k = tid * cpg; k < (tid + 1) * cpg; ++k) { a = fmaf(gamma[k], sdb[k], a); b = fmaf(gamma[k], sdg[k], b); }
        gsum[((size_t)n * G + tid) * 2] = a;
        gsum[((size_t)n * G + tid) * 2 + 1] = b;
    }
}

__global__ __launch_bounds__(kThreads) void gn_bwd_apply(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ chan,
    const float *__restrict__ gsum, int N, int HW, int C, int G, int slabs, float *__restrict__ dx,
    float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cols = C >> 2, cpg = C / G, cpg4 = cpg >> 2;
    if (n == 0 && s == 0) {                                    // parameter gradients: sum over the samples, in order
        for (int c = tid; c < C; c += kThreads) {
            float a = 0.f, b = 0.f;
            for (int m = 0; m < N; ++m) { a += chan[((size_t)m * 2) * C + c]; b += chan[((size_t)m * 2 + 1) * C + c]; }
            dbeta[c] = a; dgamma[c] = b;
        }
    }
    const int rows_per_pass = kThreads / cols;
    const int col = tid % cols, prow = tid / cols;
    if (prow >= rows_per_pass) return;
    const int g = col / cpg4;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const float inv_m = 1.f / ((float)HW * (float)cpg);
    const float k1 = gsum[((size_t)n * G + g) * 2] * inv_m, k2 = gsum[((size_t)n * G + g) * 2 + 1] * inv_m;
    const float4 g4 = reinterpret_cast<const float4 *>(gamma)[col];
    const float *xn = x + (size_t)n * HW * C, *dn = dy + (size_t)n * HW * C;
    float *on = dx + (size_t)n * HW * C;
    const int p1 = slab_begin(s + 1, slabs, HW);
    for (int p = slab_begin(s, slabs, HW) + prow; p < p1; p += rows_per_pass) {
        const float4 v = *reinterpret_cast<const float4 *>(xn + (size_t)p * C + col * 4);
        const float4 d = *reinterpret_cast<const float4 *>(dn + (size_t)p * C + col * 4);
        float4 o;
        o.x = rs * (d.x * g4.x - (k1 + (v.x - mu) * rs * k2));
        o.y = rs * (d.y * g4.y - (k1 + (v.y - mu) * rs * k2));
        o.z = rs * (d.z * g4.z - (k1 + (v.z - mu) * rs * k2));
        o.w = rs * (d.w * g4.w - (k1 + (v.w - mu) * rs * k2));
        *reinterpret_cast<float4 *>(on + (size_t)p * C + col * 4) = o;
    }
}

bool shape_ok(int64_t N, int64_t HW, int64_t C, int64_t G) {
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0) return false;
    const int64_t cpg = C / G;
    return C % 4 == 0 && cpg % 4 == 0 && C <= kMaxC && (C / 4) <= kThreads && kThreads % (C / 4) == 0 &&
           G <= kThreads && N <= 65535 && N * HW * C <= 0x7fffffffLL * 4;
}

int stat_slabs_for(int64_t HW) { return (int)(HW < kMaxSlabs * 16 ? (HW + 15) / 16 : kMaxSlabs); }
int apply_slabs_for(int64_t HW) { return (int)(HW < 256 * 16 ? (HW + 15) / 16 : 256); }

}  // namespace

extern "C" int64_t datr_groupnorm_partial_floats(int64_t N, int64_t HW, int64_t C, int64_t G) {
    return N * stat_slabs_for(HW) * 2 * C + N * 2 * C + N * G * 2;    // the backward's need (slab partials, per-sample channel and group sums); >= the forward's
}

extern "C" int datr_groupnorm_nhwc_forward_f32(const float *x, const float *gamma, const float *beta, int64_t N,
                                               int64_t HW, int64_t C, int64_t G, float eps, float *y, float *mean,
                                               float *rstd, float *partial, void *stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || !partial) return DATR_EINVAL;
    if (!shape_ok(N, HW, C, G)) return DATR_EUNSUPPORTED;
    const int ss = stat_slabs_for(HW), as = apply_slabs_for(HW);
    hipLaunchKernelGGL(gn_stats, dim3(ss, (unsigned)N), dim3(kStatThreads), 0, (hipStream_t)stream, x, (int)HW, (int)C,
                       (int)G, ss, partial);
    hipLaunchKernelGGL(gn_finish, dim3((unsigned)N), dim3(kThreads), 0, (hipStream_t)stream, x, partial, (int)HW, (int)C,
                       (int)G, ss, eps, mean, rstd);
    hipLaunchKernelGGL(gn_apply, dim3(as, (unsigned)N), dim3(kThreads), 0, (hipStream_t)stream, x, gamma, beta,
                       mean, rstd, (int)HW, (int)C, (int)G, as, y);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_groupnorm_nhwc_backward_f32(const float *dy, const float *x, const float *mean,
                                                const float *rstd, const float *gamma, int64_t N, int64_t HW,
                                                int64_t C, int64_t G, float *dx, float *dgamma, float *dbeta,
                                                float *partial, void *stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !partial) return DATR_EINVAL;
    if (!shape_ok(N, HW, C, G)) return DATR_EUNSUPPORTED;
    const int ss = stat_slabs_for(HW), as = apply_slabs_for(HW);
    float *chan = partial + N * ss * 2 * C, *gsum = chan + N * 2 * C;
    hipLaunchKernelGGL(gn_bwd_stats, dim3(ss, (unsigned)N), dim3(kStatThreads), 0, (hipStream_t)stream, dy, x, mean,
                       rstd, (int)HW, (int)C, (int)G, ss, partial);
    hipLaunchKernelGGL(gn_bwd_finish, dim3((unsigned)N), dim3(kStatThreads), 0, (hipStream_t)stream, partial, gamma, (int)C,
                       (int)G, ss, chan, gsum);
    hipLaunchKernelGGL(gn_bwd_apply, dim3(as, (unsigned)N), dim3(kThreads), 0, (hipStream_t)stream, dy, x, mean,
                       rstd, gamma, chan, gsum, (int)N, (int)HW, (int)C, (int)G, as, dx, dgamma, dbeta);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
