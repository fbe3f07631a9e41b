// stem.hip -- the ResNet-50 stem: 7x7 / stride 2 / pad 3 convolution 3 -> 64 + frozen batch norm +
// ReLU in one launch, forward only (/root/reference/models/dino/backbone.py:79-81 freezes conv1 and
// layer1, :62-72 is the frozen norm; the input image needs no gradient).  It was the library's
// (MIOpen implicit GEMM, 248 us at 4 x 1333x800) followed by a separate scale / shift / ReLU pass
// over the 275 MB result (110 us).
//
// Exact-fp32 MFMA implicit GEMM, positions as rows, the 64 output channels as columns, K = 7 filter
// rows x (7 taps x 3 channels = 21 contiguous floats of an image row, + 1 zero pad) = 154.  A
// persistent workgroup keeps the 154 x 64 weight matrix in LDS (row stride 96 floats: the two k of
// an MFMA step sit 32 banks apart) and walks 8 x 16-position tiles; a tile's input patch (21 rows x
// 111 floats, zero outside the image) travels global -> registers one tile ahead -> LDS.  In the
// patch a position's K-row for filter row r is 21 consecutive floats at (2 py + r) * 112 + 6 px:
// no im2col, no gather.  Wave w owns positions 32 w .. 32 w + 31 of the tile and both column blocks.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 16;
constexpr int PH = (TH - 1) * 2 + 7;            // 21 patch rows
constexpr int PWF = 112;                        // floats per patch row: 37 pixels x 3 + 1 pad
constexpr int KR = 22, KT = 7 * KR;             // k per filter row (21 + pad), total
constexpr int WSTR = 96;                        // weight row stride in LDS
constexpr int kLoads = (PH * PWF + kThreads - 1) / kThreads;     // 10 dwords per thread and tile

__global__ __launch_bounds__(kThreads, 2) void stem_conv(const float *__restrict__ x, const float *__restrict__ wk,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         float *__restrict__ y, int N, int H, int W, int Ho, int Wo,
                                                         int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wl = smem;                                                    // [KT][WSTR]: 59 KB
    float *patch = smem + KT * WSTR;                                     // [PH][PWF]: 9.4 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;

    for (int i = tid; i < KT * 16; i += kThreads) {                      // 154 rows x 16 float4
        const int k = i >> 4, q = i & 15;
        *reinterpret_cast<float4 *>(&wl[k * WSTR + q * 4]) = *reinterpret_cast<const float4 *>(wk + k * 64 + q * 4);
    }
    float sc[2], sh[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) { sc[jn] = scale[jn * 32 + l31]; sh[jn] = shift[jn * 32 + l31]; }

    const int ntiles = N * tiles_y * tiles_x;
    float pre[kLoads];
    auto fetch = [&](int tile) {
        int b = tile;
        const int tx = b % tiles_x; b /= tiles_x;
        const int ty = b % tiles_y; b /= tiles_y;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)b * H * W * 3), 0, H * W * 3 * 4, 0x00020000);
        const int y0 = ty * TH * 2 - 3, xf0 = (tx * TW * 2 - 3) * 3;
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int e = tid + u * kThreads;
            const int pr = e / PWF, j = e - pr * PWF;
            const int yy = y0 + pr, xf = xf0 + j;
            const bool in = e < PH * PWF && j < PWF - 1 && yy >= 0 && yy < H && xf >= 0 && xf < W * 3;
            pre[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                xr, in ? (yy * W * 3 + xf) * 4 : (int)0x80000000, 0, 0));
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    const int py = wave * 2 + (l31 >> 4), px = l31 & 15;
    const int abase = (2 * py) * PWF + 6 * px + lhi;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                    // the previous tile's readers are done (and wl is written)
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int e = tid + u * kThreads;
            if (e < PH * PWF) patch[e] = pre[u];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);

        f32x16 acc[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
#pragma unroll
            for (int jj = 0; jj < KR / 2; ++jj) {
                const float av = patch[abase + r * PWF + 2 * jj];
                const float *bp = &wl[(r * KR + 2 * jj + lhi) * WSTR + l31];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32], acc[1], 0, 0, 0);
            }
        }
        int b = tile;
        const int tx = b % tiles_x; b /= tiles_x;
        const int ty = b % tiles_y; b /= tiles_y;
        // Stores through a per-image buffer descriptor: a position outside the map gets an out-of-range offset and is
        // dropped, so there is no branch around the stores and the compiler knows how many are in flight -- the wait
        // for the next tile's prefetched patch at the top of the loop then lets these 32 stores drain behind it
        // (with bounds branches it waited for everything, s_waitcnt vmcnt(0), once per tile).
        const int bu = __builtin_amdgcn_readfirstlane(b);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            y + (size_t)bu * Ho * Wo * 64, 0, Ho * Wo * 64 * 4, 0x00020000);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = (e & 3) + 8 * (e >> 2) + 4 * lhi;          // position within the wave's 32
            const int oy = ty * TH + wave * 2 + (prow >> 4), ox = tx * TW + (prow & 15);
            const unsigned off = ((oy < Ho) & (ox < Wo)) ? (unsigned)(((oy * Wo + ox) * 64 + l31) * 4) : 0x80000000u;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const float v = acc[jn][e] * sc[jn] + sh[jn];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v > 0.f ? v : 0.f), yr, off, jn * 128, 2);
            }
        }
    }
}

}  // namespace

extern "C" int datr_stem_conv7x7_bn_relu_nhwc_f32(const float *x, const float *wk, const float *scale,
                                                  const float *shift, int64_t N, int64_t H, int64_t W, float *y,
                                                  void *stream)
{
    if (!x || !wk || !scale || !shift || !y || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if (H * W * 3 * 4 >= (1LL << 31) || ((H + 1) / 2) * ((W + 1) / 2) * 64 * 4 >= (1LL << 31)) return DATR_EUNSUPPORTED;
    if (N == 0) return DATR_OK;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const long ntiles = (long)N * tiles_x * tiles_y;
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);          // two workgroups per CU
    constexpr size_t lds = (size_t)(KT * WSTR + PH * PWF) * sizeof(float);
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(stem_conv),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(stem_conv, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, x, wk, scale, shift, y, (int)N,
                       (int)H, (int)W, Ho, Wo, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
