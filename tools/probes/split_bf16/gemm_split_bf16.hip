// gemm_split_bf16.hip -- EXPERIMENT, not part of the product library: an fp32 GEMM C = A B^T (A [M, K], B [N, K],
// K contiguous) on the bf16 matrix pipe with both operands split exactly into three bf16 pieces each,
//     a = a_hi + a_mid + a_lo   (24 mantissa bits = 3 x 8; every residual a - bf16(a) is exact in fp32),
// and the six largest of the nine piece products accumulated in fp32 (the three dropped ones are below 2^-32
// of the product).  v_mfma_f32_32x32x16_bf16 does 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32, so
// six products leave 2.67x the fp32 pipe's rate -- if the split (5.5 VALU operations per element) stays out of
// the way.  This file measures that: accuracy against float64 and time against the fp32-MFMA family.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC gemm_split_bf16.hip -o libsplit_bf16.so
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LDK = 20;      // LDS row stride in floats (16 + 4 pad)

struct Split { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Split split8(const f4 u, const f4 v) {
    Split s;
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)r2;
    }
    return s;
}

template <int PRODUCTS>
__global__ __launch_bounds__(256, 2) void gemm_nt_split(const float *__restrict__ A, const float *__restrict__ B,
                                                        float *__restrict__ C, int M, int N, int K)
{
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x % ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 256, row = idx >> 2, kq = idx & 3;
            const int am = min(m0 + row, M - 1);
            ra[u] = *reinterpret_cast<const f4 *>(A + (size_t)am * K + k0 + kq * 4);
            rb[u] = *reinterpret_cast<const f4 *>(B + (size_t)(n0 + row) * K + k0 + kq * 4);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 256, row = idx >> 2, kq = idx & 3;
            *reinterpret_cast<f4 *>(&As[buf][row][kq * 4]) = ra[u];
            *reinterpret_cast<f4 *>(&Bs[buf][row][kq * 4]) = rb[u];
        }
    };
    const int nk = K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        Split a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *pa = &As[buf][wm * 64 + t * 32 + l31][8 * lhi];
            const float *pb = &Bs[buf][wn * 64 + t * 32 + l31][8 * lhi];
            a[t] = split8(*reinterpret_cast<const f4 *>(pa), *reinterpret_cast<const f4 *>(pa + 4));
            b[t] = split8(*reinterpret_cast<const f4 *>(pb), *reinterpret_cast<const f4 *>(pb + 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                if (PRODUCTS >= 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].lo, b[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].lo, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b[j].mid, acc[i][j], 0, 0, 0);
                }
                if (PRODUCTS >= 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].mid, acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].hi, acc[i][j], 0, 0, 0);
            }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (m < M) C[(size_t)m * N + n] = acc[i][j][e];
            }
}

}  // namespace

// products: 1 (bf16 x bf16), 3, 6
extern "C" int split_bf16_gemm_nt(const float *A, const float *B, float *C, int M, int N, int K, int products, void *stream) {
    if (N % BN || K % BK) return 1;
    dim3 grid((unsigned)(((M + BM - 1) / BM) * (N / BN)));
    if (products == 6) hipLaunchKernelGGL(gemm_nt_split<6>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
    else if (products == 3) hipLaunchKernelGGL(gemm_nt_split<3>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
    else hipLaunchKernelGGL(gemm_nt_split<1>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
