// conv3x3.hip -- fp32 MFMA implicit-GEMM 3x3 / stride 1 / pad 1 convolution, NCHW, for gfx950.
//
// Used by the image-level domain discriminator of DATR (`FCDiscriminator_img`,
// /root/reference/models/dino/DA_utils.py:61-79: conv3x3 256->256->128->128->1 with
// LeakyReLU(0.2) in between, applied to the four pyramid levels of all 2B images behind a
// gradient-reversal layer, /root/reference/models/dino/dino.py:351-359).  The reference runs
// cuDNN convs plus separate bias / LeakyReLU kernels; here
//
//   Y[n, co, y, x] = act( bias[co] + sum_{ci,r,s} W[co,ci,r,s] * X[n, ci, y+r-1, x+s-1] )
//
// is ONE kernel: a GEMM  D[Cout x Pixels] = Wt^T[Cout x K] * im2col(X)[K x Pixels],
// K = 9*Cin ordered (r,s)-major / ci-minor so that a 16-deep K step never straddles a filter
// tap and the im2col gather needs no integer division in the loop.
//
//   * exact fp32: v_mfma_f32_32x32x2_f32 (157 TF/s peak, bitwise an fmaf chain);
//   * orientation: MFMA rows = output channels, MFMA columns = pixels, so that a lane's
//     accumulator column is a pixel and the epilogue stores 32 consecutive pixels (128 B) per
//     half-wave straight into NCHW -- no LDS transpose;
//   * workgroup 256 threads = 2x2 waves, tile 128 (cout) x 128 (pixels), each wave 64x64 =
//     2x2 MFMA blocks (64 accumulator VGPRs); BK = 16; operands go global -> registers -> LDS
//     with the next K step's loads in flight while the current one is multiplied;
//   * pixels are flattened over (n, y, x) per level, so the four images of a level share one
//     launch; out-of-image taps read as zero (padding) through a per-lane predicate;
//   * epilogue fuses bias and LeakyReLU (slope 1 = identity; slope 0 = ReLU).
// The same kernel computes the data gradient: dX = conv3x3(dY, W') with
// W'[ci,co,r,s] = W[co,ci,2-r,2-s] (the caller passes the transformed Wt).
// Weight layout expected: Wt[K = 9*Cin][Cout] (row k = (r*3+s)*Cin + ci), i.e.
// W.permute(2,3,1,0).reshape(9*Cin, Cout).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int BM = 128;     // output channels per workgroup
constexpr int BN = 128;     // pixels per workgroup
constexpr int BK = 16;

__global__ __launch_bounds__(kThreads) void conv3x3_mfma_f32(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ bias,
    float *__restrict__ Y, int N, int Cin, int Cout, int H, int W, float slope, float out_scale)
{
    __shared__ float As[2][BK][BM];      // weights: As[k][co]
    __shared__ float Bs[2][BK][BN];      // im2col : Bs[k][pixel]

    const int HW = H * W;
    const int P = N * HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = blockIdx.y * BM;
    const int p0 = blockIdx.x * BN;
    const int K = 9 * Cin;

    // ---- this thread's slice of the operand tiles ----------------------------------------------
    // A tile [BK][BM]: thread t loads column (t % 128), rows (t / 128) * 8 .. + 7
    // B tile [BK][BN]: same shape, column = pixel
    const int col = tid & 127, krow0 = (tid >> 7) * 8;
    const int a_co = co0 + col;
    const bool a_ok = a_co < Cout;
    const int p = p0 + col;
    const bool p_ok = p < P;
    const int n = p_ok ? p / HW : 0;
    const int pix = p_ok ? p - n * HW : 0;
    const int y = pix / W, x = pix - y * W;
    const float *xbase = X + (size_t)n * Cin * HW + pix;

    float ra[8], rb[8];
    auto load_tile = [&](int k0) {
        // weights
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + krow0 + i;
            ra[i] = (a_ok && k < K) ? Wt[(size_t)k * Cout + a_co] : 0.f;
        }
        // im2col: one filter tap per K step (Cin % BK == 0 is checked on the host)
        const int rs = k0 / Cin, ci0 = k0 - rs * Cin + krow0;
        const int r = rs / 3, s = rs - r * 3;
        const int yy = y + r - 1, xx = x + s - 1;
        const bool ok = p_ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float *src = xbase + (size_t)ci0 * HW + (r - 1) * W + (s - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = ok ? src[(size_t)i * HW] : 0.f;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            As[buf][krow0 + i][col] = ra[i];
            Bs[buf][krow0 + i][col] = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    const int nk = K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);            // in flight during the MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = As[buf][kk + lhi][wm * 64 + l31];
            const float a1 = As[buf][kk + lhi][wm * 64 + 32 + l31];
            const float b0 = Bs[buf][kk + lhi][wn * 64 + l31];
            const float b1 = Bs[buf][kk + lhi][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + LeakyReLU, 32 consecutive pixels per half-wave into NCHW -------------
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int pp = p0 + wn * 64 + jn * 32 + l31;
        if (pp >= P) continue;
        const int nn = pp / HW, px = pp - nn * HW;
        float *ybase = Y + (size_t)nn * Cout * HW + px;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (co < Cout) {
                    float v = acc[i][jn][e] + (bias ? bias[co] : 0.f);
                    v = v > 0.f ? v : v * slope;
                    ybase[(size_t)co * HW] = v * out_scale;
                }
            }
        }
    }
}

}  // namespace

extern "C" int datr_conv3x3_forward_f32(const float *x, const float *wt, const float *bias,
                                        int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                        float slope, float out_scale, float *y, void *stream) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DATR_EINVAL;
    if (!x || !wt || !y) return DATR_EINVAL;
    if (Cin % BK != 0) return DATR_EUNSUPPORTED;
    const int64_t P = N * H * W;
    if (P > 0x7fffffff || N * Cin * H * W > 0x7fffffff || N * Cout * H * W > 0x7fffffff)
        return DATR_EUNSUPPORTED;
    dim3 grid((unsigned)((P + BN - 1) / BN), (unsigned)((Cout + BM - 1) / BM));
    hipLaunchKernelGGL(conv3x3_mfma_f32, grid, dim3(kThreads), 0, (hipStream_t)stream, x, wt, bias,
                       y, (int)N, (int)Cin, (int)Cout, (int)H, (int)W, slope, out_scale);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
