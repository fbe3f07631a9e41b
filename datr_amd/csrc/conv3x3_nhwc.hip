// conv3x3_nhwc.hip -- exact-fp32 MFMA 3x3 / stride 1 / pad 1 convolution on NHWC tensors with a
// fused bias + LeakyReLU epilogue: the layers of DATR's image-level domain discriminator
// (`FCDiscriminator_img`, /root/reference/models/dino/DA_utils.py:61-79: conv3x3 256 -> 256 ->
// 128 -> 128 (-> 1), LeakyReLU(0.2) in between, on the four pyramid levels of all 2B images behind
// a gradient-reversal layer, /root/reference/models/dino/dino.py:351-359).
//
//   Y[n, y, x, co] = out_scale * lrelu( bias[co] + sum_{r,s,ci} W[r,s,ci,co] * X[n, y+r-1, x+s-1, ci] )
//
// Implicit GEMM with pixels as MFMA rows and output channels as MFMA columns (a lane's
// accumulator column is an output channel, so 32 lanes store 128 contiguous bytes of NHWC output):
//   * workgroup = 256 threads = 2 x 2 waves; tile = 8 x 16 output pixels x 128 output channels;
//     a wave owns 64 pixels (4 image rows of 16) x 64 channels = 2 x 2 MFMA 32x32 blocks;
//   * K runs over input-channel chunks of 16 and, inside a chunk, over the 9 filter taps.  The
//     INPUT PATCH of a chunk (10 x 18 pixels x 16 channels, halo included, zero outside the
//     image) is staged in LDS ONCE and re-used by all 9 taps -- 9x fewer activation loads than an
//     im2col-style K loop; only the 16 x 128 weight slab changes per tap (8 KB, brought in by
//     LDS-DMA -- `global_load_lds_dwordx4`, two instructions per wave, no staging registers --
//     while the previous tap is multiplied);
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak).  The K pairing is chosen for the LDS: a
//     lane fetches FOUR consecutive channels of its pixel with one ds_read_b128 (lanes 0-31:
//     channels 0-3 of an 8-channel group, lanes 32-63: channels 4-7), and MFMA step t multiplies
//     channel pair (t, 4 + t); the weight operand is read with the same pairing.  The patch row
//     stride is 20 floats (16 + 4 pad): with it the 16 lanes of every ds_read_b128 hardware
//     group ({0-3,12-15,20-27}, ...) fall on 64 distinct banks.
// The same kernel computes the data gradient: dX = conv3x3(dY, W') with
// W'[r,s,co,ci] = W[2-r,2-s,ci,co] (the caller passes the transformed weights); out_scale = -1
// folds a gradient-reversal layer in.
// Weight layout expected: Wt[9][Cin][Cout] (tap-major, output channel contiguous) =
// W.permute(2, 3, 1, 0).  Cin % 16 == 0, Cout % 128 == 0.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 16;                 // output pixels per workgroup
constexpr int PH = TH + 2, PW = TW + 2;        // patch with halo
constexpr int CK = 16;                         // input channels per chunk
constexpr int PSTR = 20;                       // patch row stride in floats (16 + 4 pad)
constexpr int BN = 128;                        // output channels per workgroup

__global__ __launch_bounds__(kThreads) void conv3x3_nhwc_mfma(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
    float *__restrict__ Y, int H, int W, int Cin, int Cout, int tiles_x, int tiles_y, float slope,
    float out_scale)
{
    __shared__ __attribute__((aligned(16))) float patch[PH * PW * PSTR];        // 14.4 KB
    __shared__ __attribute__((aligned(16))) float wsl[2][CK][BN];               // 16 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int n = b;
    const int co0 = blockIdx.y * BN;
    const int y0 = ty * TH, x0 = tx * TW;
    const float *Xn = X + (size_t)n * H * W * Cin;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // weight slab [CK][BN] of (tap, chunk): 16 rows of 512 contiguous bytes -> LDS by LDS-DMA
    // (1 KiB = two rows per wave instruction, two instructions per wave): no staging registers
    auto dma_w = [&](int tap, int ci0, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (wave * 2 + u) * 2 + (lane >> 5);            // 0..15
            const float *src = Wt + ((size_t)tap * Cin + ci0 + row) * Cout + co0 + (lane & 31) * 4;
            __builtin_amdgcn_global_load_lds(
                (__attribute__((address_space(1))) const void *)src,
                (__attribute__((address_space(3))) void *)&wsl[buf][(wave * 2 + u) * 2][0], 16, 0, 0);
        }
    };
    // patch staging: PH * PW pixels x 4 float4 (16 channels) = 720 float4
    auto load_patch = [&](int ci0) {
        for (int f = tid; f < PH * PW * 4; f += kThreads) {
            const int pix = f >> 2, q = f & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int yy = y0 + py - 1, xx = x0 + px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                v = *reinterpret_cast<const float4 *>(Xn + ((size_t)yy * W + xx) * Cin + ci0 + q * 4);
            *reinterpret_cast<float4 *>(&patch[pix * PSTR + q * 4]) = v;
        }
    };

    // A-operand base: pixel block i of this wave: rows 2 i, 2 i + 1 of its 4 rows; lane l31 ->
    // (row = l31 >> 4, col = l31 & 15)
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = wm * 4 + i * 2 + (l31 >> 4), px = l31 & 15;
        abase[i] = (py * PW + px) * PSTR + lhi * 4;
    }

    const int nchunks = Cin / CK;
    dma_w(0, 0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int ci0 = ch * CK;
        __syncthreads();                        // previous chunk's readers are done with the patch
        load_patch(ci0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // 9 taps per chunk: the slab buffer parity follows the global step count
            const int buf = (ch * 9 + tap) & 1;
            const bool more = tap < 8 || ch + 1 < nchunks;
            if (more) dma_w(tap < 8 ? tap + 1 : 0, tap < 8 ? ci0 : ci0 + CK, buf ^ 1);
            const int r = tap / 3, s = tap - r * 3;
            const int toff = (r * PW + s) * PSTR;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {          // two 8-channel groups of the chunk
                float4 a[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&patch[abase[i] + toff + grp * 8]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = grp * 8 + lhi * 4 + t;
                    const float b0 = wsl[buf][k][wn * 64 + l31];
                    const float b1 = wsl[buf][k][wn * 64 + 32 + l31];
                    const float a0 = t == 0 ? a[0].x : t == 1 ? a[0].y : t == 2 ? a[0].z : a[0].w;
                    const float a1 = t == 0 ? a[1].x : t == 1 ? a[1].y : t == 2 ? a[1].z : a[1].w;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            if (more && tap < 8) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next slab landed (this wave's part)
                __syncthreads();
            }
        }
    }

    // ---- epilogue: bias + LeakyReLU, 32 lanes = 32 consecutive output channels (128 B) ----------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = (e & 3) + 8 * (e >> 2) + 4 * lhi;          // pixel within the 32-block
            const int py = wm * 4 + i * 2 + (prow >> 4), px = prow & 15;
            const int yy = y0 + py, xx = x0 + px;
            if (yy < H && xx < W) {
                float *yb = Y + (((size_t)n * H + yy) * W + xx) * Cout + co0 + wn * 64 + l31;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    float v = acc[i][jn][e] + (bias ? bias[co0 + wn * 64 + jn * 32 + l31] : 0.f);
                    v = v > 0.f ? v : v * slope;
                    yb[jn * 32] = v * out_scale;
                }
            }
        }
    }
}

}  // namespace

extern "C" int datr_conv3x3_nhwc_forward_f32(const float *x, const float *wt, const float *bias,
                                             int64_t N, int64_t H, int64_t W, int64_t Cin,
                                             int64_t Cout, float slope, float out_scale, float *y,
                                             void *stream) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DATR_EINVAL;
    if (!x || !wt || !y) return DATR_EINVAL;
    if (Cin % CK != 0 || Cout % BN != 0) return DATR_EUNSUPPORTED;
    if (N * H * W * (Cin > Cout ? Cin : Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int tiles_x = (int)((W + TW - 1) / TW), tiles_y = (int)((H + TH - 1) / TH);
    dim3 grid((unsigned)(N * tiles_x * tiles_y), (unsigned)(Cout / BN));
    hipLaunchKernelGGL(conv3x3_nhwc_mfma, grid, dim3(kThreads), 0, (hipStream_t)stream, x, wt, bias, y,
                       (int)H, (int)W, (int)Cin, (int)Cout, tiles_x, tiles_y, slope, out_scale);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
