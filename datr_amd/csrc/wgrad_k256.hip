// wgrad_k256.hip -- weight and bias gradient of a 256 -> 256 linear layer applied to very many
// rows:  dW[256, 256] = dY^T[256, M] * X[M, 256],  db[256] = column sums of dY,  M ~ 89 000
// encoder tokens, exact fp32 on the MFMA units.
//
// These are the weight gradients of the value / output projections of `MSDeformAttn`
// (/root/reference/models/dino/ops/modules/ms_deform_attn.py:92-95) and of `enc_output`
// (/root/reference/models/dino/deformable_transformer.py:200-201) over all encoder tokens of the
// source + target pass: 21 per training step.  The library GEMM selected for this shape (tiny
// 256 x 256 output, reduction length 88 892) runs at 60 TF/s (192 us), and the bias gradient
// is a second pass over dY (41 us).
//
// Design: split the REDUCTION (the rows) evenly over one workgroup per CU; every workgroup
// accumulates the complete 256 x 256 product of its ~350 rows in registers (8 waves x 8
// accumulator blocks of 32 x 32), so there is no tile quantisation and no LDS at all:
//   * both MFMA operands of a row are contiguous in memory (a row of dY, a row of X), and the
//     column -> lane assignment of an MFMA block is free.  Lane i of block (nb, kb) owns output
//     row n = 8 i + nb and output column k = 8 i + kb, so the two dY values and four X values a
//     lane needs for its wave's 2 x 4 blocks are ADJACENT: one 8-byte and one 16-byte load per
//     lane and row pair feed eight v_mfma_f32_32x32x2_f32 (lanes 0-31 supply the even row of
//     the pair, lanes 32-63 the odd one);
//   * the loads of the next four row pairs are in flight while the current four are multiplied
//     (a deeper ring of stages measured slower: 153 vs 145 us);
//   * the dY column sums fall out of the A operands (two VALU adds per row pair in half of the
//     waves): the bias gradient costs no extra pass;
//   * every workgroup writes its partial product (+ one row of partial column sums) with 16-byte
//     stores; a second kernel adds the partials in a fixed order (deterministic, unlike atomics).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 256;                       // in_features == out_features
constexpr int kThreads = 512;
constexpr int kU = 4;                         // row pairs per pipeline stage
constexpr int kPartialRows = kD + 1;          // + the column sums of dY

__global__ __launch_bounds__(kThreads) void wgrad_k256_kernel(
    const float *__restrict__ dY, const float *__restrict__ X, int M, int rows_per_block,
    float *__restrict__ partial)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int m0 = blockIdx.x * rows_per_block;
    const int m1 = min(M, m0 + rows_per_block);
    float *pb = partial + (size_t)blockIdx.x * kPartialRows * kD;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float s0 = 0.f, s1 = 0.f;

    const float *ya = dY + l31 * 8 + 2 * wn;
    const float *xb = X + l31 * 8 + 4 * wk;
    // rows past m1 contribute zero: address clamped, A operand zeroed
    auto load = [&](int m, float2 &a, float4 &b) {
        const bool ok = m < m1;
        const size_t r = (size_t)(ok ? m : m0) * kD;
        a = *reinterpret_cast<const float2 *>(ya + r);
        b = *reinterpret_cast<const float4 *>(xb + r);
        if (!ok) { a = make_float2(0.f, 0.f); b = make_float4(0.f, 0.f, 0.f, 0.f); }
    };

    float2 an[kU];
    float4 bn[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) load(m0 + 2 * u + lhi, an[u], bn[u]);
    for (int m = m0; m < m1; m += 2 * kU) {
        float2 a[kU];
        float4 b[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { a[u] = an[u]; b[u] = bn[u]; }
        if (m + 2 * kU < m1) {
#pragma unroll
            for (int u = 0; u < kU; ++u) load(m + 2 * kU + 2 * u + lhi, an[u], bn[u]);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            s0 += a[u].x;
            s1 += a[u].y;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].z, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].w, acc[1][3], 0, 0, 0);
        }
    }

    // ---- partial product: D row r of block (i, j) is output row n = 8 r + 2 wn + i; lane l31 holds
    //      output columns 8 l31 + 4 wk + j, j = 0..3: one float4 store per (i, e) -----------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            const int n = 8 * r + 2 * wn + i;
            *reinterpret_cast<float4 *>(pb + (size_t)n * kD + l31 * 8 + 4 * wk) =
                make_float4(acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]);
        }
    // ---- partial column sums of dY: lanes of both halves hold columns 8 l31 + 2 wn + {0, 1} -------
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (wk == 0 && lhi == 0)
        *reinterpret_cast<float2 *>(pb + (size_t)kD * kD + l31 * 8 + 2 * wn) = make_float2(s0, s1);
}

// out[e] = sum_b partial[b][e] for the 257 x 256 entries: a workgroup owns 64 float4 columns and
// splits the partials over 16 thread groups (fixed order inside a group, fixed order over the
// groups afterwards).
constexpr int kFinishParts = 16;

__global__ __launch_bounds__(64 * kFinishParts) void wgrad_k256_finish(
    const float4 *__restrict__ partial, int blocks, float4 *__restrict__ dW, float4 *__restrict__ db)
{
    __shared__ float4 red[kFinishParts][64];
    constexpr int kVec = kPartialRows * kD / 4;            // float4 per partial
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < kVec) {
        const int per = (blocks + kFinishParts - 1) / kFinishParts;
        const int b0 = part * per, b1 = min(blocks, b0 + per);
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const float4 v = partial[(size_t)b * kVec + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && col < kVec) {
        float4 t = red[0][threadIdx.x];
#pragma unroll
        for (int p = 1; p < kFinishParts; ++p) {
            const float4 v = red[p][threadIdx.x];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        if (col < kD * kD / 4) dW[col] = t;
        else if (db) db[col - kD * kD / 4] = t;
    }
}

}  // namespace

extern "C" int64_t datr_wgrad_k256_scratch_floats(void) {
    return (int64_t)DATR_WGRAD_K256_MAX_BLOCKS * kPartialRows * kD;
}

extern "C" int datr_wgrad_k256_f32(const float *dy, const float *x, int64_t M, float *scratch,
                                   float *dw, float *db, void *stream) {
    if (M <= 0) return DATR_EINVAL;
    if (!dy || !x || !scratch || !dw) return DATR_EINVAL;
    if (M > 0x3fffffffLL) return DATR_EUNSUPPORTED;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return DATR_ELAUNCH;
        cus = prop.multiProcessorCount;
    }
    int blocks = cus < DATR_WGRAD_K256_MAX_BLOCKS ? cus : DATR_WGRAD_K256_MAX_BLOCKS;
    // an even number of rows per workgroup (row pairs), no empty workgroup
    int64_t rows = ((M + blocks - 1) / blocks + 1) & ~(int64_t)1;
    blocks = (int)((M + rows - 1) / rows);
    hipLaunchKernelGGL(wgrad_k256_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       dy, x, (int)M, (int)rows, scratch);
    constexpr int kVec = kPartialRows * kD / 4;
    hipLaunchKernelGGL(wgrad_k256_finish, dim3((kVec + 63) / 64), dim3(64 * kFinishParts), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(scratch), blocks, reinterpret_cast<float4 *>(dw),
                       reinterpret_cast<float4 *>(db));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
