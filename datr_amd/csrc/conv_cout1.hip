// conv_cout1.hip -- 3x3 / stride 1 / pad 1 convolution with ONE output channel on NHWC tensors:
// the classifier of the image-level domain discriminator
// (/root/reference/models/dino/DA_utils.py:67,78: `nn.Conv2d(ndf2, 1, kernel_size=3, padding=1)`,
// called on every pyramid level from dino.py:351-359), forward and backward.
//
// With one output channel there is no matrix to feed the MFMA units: every output pixel is a
// 9 x C dot product and the op is a stream over the activation tensor (C = 128: 512 B per pixel,
// 45.5 MB over the four levels of four 1333x800 images).  The library ran it as an implicit GEMM
// padded to a 64-wide N tile (147 us per level-0 forward call, r03_step_kernels.csv).  Here:
//   forward   32 lanes x float4 cover a pixel's C = 128 channels (the wave = 2 pixels); the 9 filter
//             taps live in registers; 9 row loads per pixel (neighbours come out of L1 / L2), one
//             5-step shuffle reduction; out = bias + sum.
//   dgrad     dA[p, c] = sum_t dY[p - t] W[t, c]; the LeakyReLU gate of the layer below
//             (DA_utils.py:76: the input of the classifier is lrelu(conv3)) is applied on the way out,
//             so the result IS the pre-activation gradient the next data-gradient launch consumes.
//   wgrad     dW[t, c] = sum_p A[p, c] dY[p - t], db = sum_p dY[p]: every workgroup walks a slice of
//             the pixels with 9 float4 accumulators per lane, partial sums per workgroup are added
//             in a fixed order by a second launch (bitwise reproducible, no atomics).
// All levels of a call share the filter; the host entry loops over the level table.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int kC = 128;                 // channels: 32 lanes x 4
constexpr int kPixPerBlock = kThreads / 32;

__device__ __forceinline__ float dot4(const f4 a, const f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// filter taps of this lane's 4 channels from the torch layout w[0, c, ky, kx]
__device__ __forceinline__ void load_taps(f4 (&wt)[9], const float *w, int c4) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
        wt[t] = f4{w[(c4 + 0) * 9 + t], w[(c4 + 1) * 9 + t], w[(c4 + 2) * 9 + t], w[(c4 + 3) * 9 + t]};
}

__global__ __launch_bounds__(kThreads) void c1_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, int N, int H, int W,
                                                   float *__restrict__ y) {
    const int lane32 = threadIdx.x & 31, sub = threadIdx.x >> 5;
    f4 wt[9];
    load_taps(wt, w, lane32 * 4);
    const float b = bias ? bias[0] : 0.f;
    const long npix = (long)N * H * W;
    for (long p = (long)blockIdx.x * kPixPerBlock + sub; p < npix; p += (long)gridDim.x * kPixPerBlock) {
        const int xw = (int)(p % W), yh = (int)((p / W) % H);
        const float *px = x + p * kC + lane32 * 4;
        // Branch-free taps: a neighbour outside the image is read at its clamped position and dropped by a select, so
        // the nine 16-byte loads of a pixel leave together (with a bounds branch per tap each load waited for the one
        // before it: load, s_waitcnt vmcnt(0), nine times).  Same sums in the same order.
        f4 v[9];
        bool in[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = yh + ky - 1, xx = xw + kx - 1;
                in[ky * 3 + kx] = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                v[ky * 3 + kx] = *reinterpret_cast<const f4 *>(px + ((yc - yh) * W + (xc - xw)) * kC);
            }
        __builtin_amdgcn_sched_barrier(0);           // all nine requests first (the scheduler otherwise re-uses one register set)
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float d = dot4(v[t], wt[t]);
            acc += in[t] ? d : 0.f;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 32);
        if (lane32 == 0) y[p] = acc + b;
    }
}

// dz[p, c] = gate(a[p, c]) * sum_t dy[p - t] w[t, c];  gate = 1 where a > 0, slope elsewhere
__global__ __launch_bounds__(kThreads) void c1_dgrad(const float *__restrict__ dy, const float *__restrict__ w,
                                                     const float *__restrict__ a, float slope, int N, int H,
                                                     int W, float *__restrict__ dz) {
    const int lane32 = threadIdx.x & 31, sub = threadIdx.x >> 5;
    f4 wt[9];
    load_taps(wt, w, lane32 * 4);
    const long npix = (long)N * H * W;
    for (long p = (long)blockIdx.x * kPixPerBlock + sub; p < npix; p += (long)gridDim.x * kPixPerBlock) {
        const int xw = (int)(p % W), yh = (int)((p / W) % H);
        const f4 av = *reinterpret_cast<const f4 *>(a + p * kC + lane32 * 4);
        float g[9];                                  // branch-free as in c1_fwd: clamped reads, selects
        bool in[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // output pixel q = p - (k - 1) saw input pixel p through tap k
                const int yy = yh - (ky - 1), xx = xw - (kx - 1);
                in[ky * 3 + kx] = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                g[ky * 3 + kx] = dy[p + ((yc - yh) * W + (xc - xw))];
            }
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += (in[t] ? g[t] : 0.f) * wt[t];
        f4 out;
        out.x = av.x > 0.f ? acc.x : acc.x * slope;
        out.y = av.y > 0.f ? acc.y : acc.y * slope;
        out.z = av.z > 0.f ? acc.z : acc.z * slope;
        out.w = av.w > 0.f ? acc.w : acc.w * slope;
        *reinterpret_cast<f4 *>(dz + p * kC + lane32 * 4) = out;
    }
}

// partial[block][10][128]: rows 0..8 = dW taps, row 9 = db (every lane carries the same value)
__global__ __launch_bounds__(kThreads) void c1_wgrad(const float *__restrict__ a, const float *__restrict__ dy,
                                                     int N, int H, int W, float *__restrict__ partial) {
    __shared__ f4 red[kPixPerBlock][10][32];
    const int lane32 = threadIdx.x & 31, sub = threadIdx.x >> 5;
    f4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
    const long npix = (long)N * H * W;
    for (long p = (long)blockIdx.x * kPixPerBlock + sub; p < npix; p += (long)gridDim.x * kPixPerBlock) {
        const int xw = (int)(p % W), yh = (int)((p / W) % H);
        const f4 av = *reinterpret_cast<const f4 *>(a + p * kC + lane32 * 4);
        db += dy[p];
        float g[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = yh - (ky - 1), xx = xw - (kx - 1);
                const bool in = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                const float d = dy[p + ((yc - yh) * W + (xc - xw))];
                g[ky * 3 + kx] = in ? d : 0.f;
            }
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += g[t] * av;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[sub][t][lane32] = acc[t];
    red[sub][9][lane32] = f4{db, 0.f, 0.f, 0.f};
    __syncthreads();
    // fixed-order sum over the block's 8 pixel groups
    for (int i = threadIdx.x; i < 10 * 32; i += kThreads) {
        const int t = i / 32, l = i % 32;
        f4 s = red[0][t][l];
#pragma unroll
        for (int g = 1; g < kPixPerBlock; ++g) s += red[g][t][l];
        *reinterpret_cast<f4 *>(partial + ((long)blockIdx.x * 10 + t) * kC + l * 4) = s;
    }
}

// dw[0, c, ky, kx] = sum over blocks, db = sum over blocks; fixed order.  One 1 024-thread workgroup per
// (tap or bias row t, 64 channels): sixteen waves walk the blocks sixteen apart (coalesced 256-B reads, 64 trips
// for 1 024 blocks instead of one thread's 1 024 dependent trips: 138 -> ~10 us), then sum through LDS in wave order.
__global__ __launch_bounds__(1024) void c1_wgrad_fold(const float *__restrict__ partial, int blocks,
                                                      float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float part[16][64];
    const int t = blockIdx.x >> 1, c = (blockIdx.x & 1) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    for (int b = g; b < blocks; b += 16) s += partial[((long)b * 10 + t) * kC + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g != 0) return;
    s = part[0][c & 63];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += part[k][c & 63];
    if (t < 9) dw[c * 9 + t] = s;
    else if (c == 0 && db) db[0] = s;
}

int blocks_for(long npix) {
    const long want = (npix + kPixPerBlock * 8 - 1) / (kPixPerBlock * 8);       // >= 8 pixels per group
    return (int)(want < 1 ? 1 : (want > 256 ? 256 : want));       // per level; the fold sums them sixteen apart
}

}  // namespace

extern "C" {

int64_t datr_conv3x3_cout1_partial_floats(const datr_c1_level *levels, int64_t nlevels, int64_t N) {
    long total = 0;
    for (int64_t l = 0; l < nlevels; ++l) total += blocks_for((long)N * levels[l].H * levels[l].W);
    return (int64_t)total * 10 * kC;
}

int datr_conv3x3_cout1_forward_f32(const datr_c1_level *levels, int64_t nlevels, int64_t N, int64_t C,
                                   const float *w, const float *bias, void *stream) {
    if (!levels || nlevels < 0 || N < 0 || !w) return DATR_EINVAL;
    if (C != kC) return DATR_EUNSUPPORTED;
    for (int64_t l = 0; l < nlevels; ++l) {
        const datr_c1_level &lv = levels[l];
        const long npix = (long)N * lv.H * lv.W;
        if (npix == 0) continue;
        if (!lv.x || !lv.y || lv.H < 1 || lv.W < 1 || npix * kC >= (1L << 40)) return DATR_EINVAL;
        hipLaunchKernelGGL(c1_fwd, dim3(blocks_for(npix)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const float *)lv.x, w, bias, (int)N, (int)lv.H, (int)lv.W, (float *)lv.y);
    }
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

int datr_conv3x3_cout1_backward_f32(const datr_c1_level *levels, int64_t nlevels, int64_t N, int64_t C,
                                    const float *w, float slope, float *dw, float *db, float *partial,
                                    void *stream) {
    if (!levels || nlevels < 0 || N < 0 || !w || !dw || !partial) return DATR_EINVAL;
    if (C != kC) return DATR_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int used = 0;
    for (int64_t l = 0; l < nlevels; ++l) {
        const datr_c1_level &lv = levels[l];
        const long npix = (long)N * lv.H * lv.W;
        if (npix == 0) continue;
        if (!lv.x || !lv.y || !lv.dx || lv.H < 1 || lv.W < 1) return DATR_EINVAL;
        const int nb = blocks_for(npix);
        // x = the classifier's input a (post-activation), y = dY [N, H, W], dx = gated gradient out
        hipLaunchKernelGGL(c1_dgrad, dim3(nb), dim3(kThreads), 0, st, (const float *)lv.y, w, (const float *)lv.x,
                           slope, (int)N, (int)lv.H, (int)lv.W, (float *)lv.dx);
        hipLaunchKernelGGL(c1_wgrad, dim3(nb), dim3(kThreads), 0, st, (const float *)lv.x, (const float *)lv.y,
                           (int)N, (int)lv.H, (int)lv.W, partial + (long)used * 10 * kC);
        used += nb;
    }
    hipLaunchKernelGGL(c1_wgrad_fold, dim3(10 * (kC / 64)), dim3(1024), 0, st, partial, used, dw, db);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

}  // extern "C"
