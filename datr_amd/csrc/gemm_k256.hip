// gemm_k256.hip -- exact-fp32 MFMA GEMM for the tall-skinny projections of the deformable
// encoder:  Y[M, N] = X[M, 256] * B[256, N] (+ bias[N]),  M ~ 89 000 tokens, N = 256 or 384.
//
// These are the value / output / sampling-offset projections of `MSDeformAttn`
// (/root/reference/models/dino/ops/modules/ms_deform_attn.py:92-95,118-124) and their data
// gradients: 11.65 GFLOP each, ~24 per training step.
//
// EXPERIMENTAL, not on the product path: measured on MI355X at M = 88 892 this kernel takes
// 123 / 127 / 187 us (N = 256 forward, N = 256 data gradient, N = 384) = 95 / 92 / 93 TF/s, the
// library GEMM that PyTorch dispatches to (hipBLASLt / rocBLAS with the committed TunableOp
// selections) 108 / 100 / 156 us = 108 / 116 / 112 TF/s; 768 threads per workgroup are slower
// (134 us).  The per-SIMD tile quantisation (5.43 tiles of 32 rows per SIMD -> 6) and the clock
// drop under sustained fp32 MFMA load bound it at ~100 us; kept as the measured baseline for a
// fused projection kernel (value_proj + mask fill, or offsets + softmax epilogue).
//
// Shape-specialised design (no K loop over LDS tiles, no barrier after start-up):
//   * a PERSISTENT workgroup per CU owns 128 output columns and keeps the whole [256 x 128]
//     slice of B in LDS (128 KB) for its lifetime; the column slices ("halves") are interleaved
//     over the XCDs so that the workgroups that stream the same rows of X at the same time share
//     an L2;
//   * each WAVE independently walks 32-row tiles of X: the A operand comes straight from global
//     memory into registers -- lane (r, h) loads the 64 contiguous bytes X[row r][k0 + 16 h ..]
//     as four float4, two lanes cover a 128-B line, nothing is staged or shared, so there is no
//     barrier and no LDS write in the main loop; the next K chunk is in flight while the current
//     one is multiplied (64 MFMAs ~ 2 us of cover);
//   * v_mfma_f32_32x32x2_f32, K pairing (j, 16 + j) to match that load; the B operand of a step is
//     ONE ds_read_b128: LDS layout [k][col % 32][col / 32], i.e. the four 32-column blocks of a lane
//     sit in one 16-byte word, and the 16 lanes of each hardware read group cover 64 banks;
//   * 4 accumulators (32 x 128 outputs) per wave; the epilogue adds the bias and stores 128-B row
//     segments.
// B is addressed as B[k][n] = Bp[k * ldk + n * ldn]: (ldk, ldn) = (1, 256) multiplies by W^T for
// a row-major weight W[N][256] (forward), (N, 1) multiplies by W[256][N] (data gradient).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kK = 256;
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kCols = 128;                         // output columns per workgroup
constexpr int kLdsBytes = kK * kCols * 4;          // 128 KB

__global__ __launch_bounds__(kThreads) void gemm_k256_kernel(
    const float *__restrict__ X, const float *__restrict__ Bp, const float *__restrict__ bias,
    float *__restrict__ Y, int M, int N, long ldk, long ldn, int slices, int slots)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];          // [256][32][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // blockIdx -> (XCD, column slice, slot): slices of one slot sit on the same XCD
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int slice = rest % slices, slot = xcd + 8 * (rest / slices);
    const int n0 = slice * kCols;

    // ---- B slice -> LDS, once ------------------------------------------------------------------
    if (ldn == 1) {            // B rows contiguous in n: consecutive threads take consecutive n
        for (int e = tid; e < kK * kCols; e += kThreads) {
            const int k = e >> 7, c = e & 127;
            wl[k * kCols + (c & 31) * 4 + (c >> 5)] = Bp[(long)k * ldk + n0 + c];
        }
    } else {                   // contiguous in k (W[n][k], ldk == 1): a lane reads four k of one column,
                               // consecutive lanes take consecutive columns (LDS stride 4 floats)
        for (int e = tid; e < (kK / 4) * kCols; e += kThreads) {
            const int c = e & 127, k4 = e >> 7;
            const float4 v = *reinterpret_cast<const float4 *>(Bp + (long)(n0 + c) * ldn + k4 * 4);
            float *dst = wl + (k4 * 4) * kCols + (c & 31) * 4 + (c >> 5);
            dst[0] = v.x; dst[kCols] = v.y; dst[2 * kCols] = v.z; dst[3 * kCols] = v.w;
        }
    }
    __syncthreads();

    const int tiles = (M + 31) >> 5;
    const int stride = slots * kWaves;
    int tile = slot * kWaves + wave;
    if (tile >= tiles) return;

    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + cb * 32 + l31] : 0.f;
    const float4 *wl4 = reinterpret_cast<const float4 *>(wl) + lhi * 16 * 32 + l31;   // + k * 32

    auto row_ptr = [&](int t) {
        const int row = min(t * 32 + l31, M - 1);
        return reinterpret_cast<const float4 *>(X + (size_t)row * kK + lhi * 16);
    };
    const float4 *xa = row_ptr(tile);
    float4 nxt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[i] = xa[i];

    while (true) {
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
        const int next_tile = tile + stride;
        const float4 *xn = row_ptr(next_tile < tiles ? next_tile : tile);

#pragma unroll 1
        for (int kb = 0; kb < 8; ++kb) {
            float4 cur[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
            const float4 *src = kb < 7 ? xa + (kb + 1) * 8 : xn;
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt[i] = src[i];
            const float4 *wk = wl4 + kb * 32 * 32;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 b = wk[j * 32];
                const float4 a4 = cur[j >> 2];
                const float a = (j & 3) == 0 ? a4.x : (j & 3) == 1 ? a4.y : (j & 3) == 2 ? a4.z : a4.w;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, acc[3], 0, 0, 0);
            }
        }

        // ---- epilogue: bias, 32 lanes = 128 contiguous bytes of an output row ------------------
        const int r0 = tile * 32 + 4 * lhi;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + (e & 3) + 8 * (e >> 2);
            if (row < M) {
                float *yb = Y + (size_t)row * N + n0 + l31;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) yb[cb * 32] = acc[cb][e] + bv[cb];
            }
        }
        if (next_tile >= tiles) break;
        tile = next_tile;
        xa = xn;
    }
}

}  // namespace

extern "C" int datr_gemm_k256_f32(const float *x, const float *b, int64_t ldk, int64_t ldn,
                                  const float *bias, int64_t M, int64_t N, float *y, void *stream) {
    if (M < 0 || N <= 0) return DATR_EINVAL;
    if (M == 0) return DATR_OK;
    if (!x || !b || !y) return DATR_EINVAL;
    if (N % kCols != 0 || N / kCols > 16 || M > 0x3fffffffLL) return DATR_EUNSUPPORTED;
    if (ldn != 1 && (ldk != 1 || ldn % 4 != 0)) return DATR_EUNSUPPORTED;
    static int cus = 0;
    static bool attr_set = false;
    if (!attr_set) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return DATR_ELAUNCH;
        cus = prop.multiProcessorCount;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_k256_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
            return DATR_ELAUNCH;
        attr_set = true;
    }
    const int slices = (int)(N / kCols);
    int per_xcd = cus / (8 * slices);                  // slots per XCD
    if (per_xcd < 1) per_xcd = 1;
    const int slots = 8 * per_xcd;
    hipLaunchKernelGGL(gemm_k256_kernel, dim3((unsigned)(slots * slices)), dim3(kThreads), kLdsBytes,
                       (hipStream_t)stream, x, b, bias, y, (int)M, (int)N, (long)ldk, (long)ldn, slices,
                       slots);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
