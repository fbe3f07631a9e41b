// ffn.hip -- the one element-wise pass of the FFN backward that is not a GEMM.
//
// Encoder / decoder FFN of the reference: linear2(dropout(relu(linear1(x))))
// (/root/reference/models/dino/deformable_transformer.py:803-806, :879-883).  With h = relu(z)
// saved from the forward, backward needs dz = dh * (h > 0) and db1 = sum_rows(dz).  ATen runs
// threshold_backward (read dh, h; write dz) and then a separate column reduction that reads dz
// again; here both happen in ONE pass, in place on dh, with a deterministic two-stage column
// sum (per-workgroup partials, then a tiny finishing kernel -- no float atomics, so db1 is
// bitwise reproducible run to run).
//   rows x cols fp32, row-major, cols % 4 == 0.  HBM traffic: read dh + h, write dz.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;

// RELU = true : dh <- dh * (h > 0) in place, column partial sums of the result
// RELU = false: column partial sums of dh (read-only; `h` unused)
template <bool RELU>
__global__ __launch_bounds__(kThreads) void relu_bwd_bias_kernel(
    float4 *__restrict__ dh, const float4 *__restrict__ h, int64_t rows, int cols4,
    float4 *__restrict__ partial)
{
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= cols4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // A row slice is a CONTIGUOUS range of rows (not every gridDim.y-th row): a workgroup streams
    // one stretch of memory -- one or two 2 MB pages instead of a new page on every iteration, which
    // is what held the strided walk at 4.6 TB/s on the [88892, 2048] encoder activation.
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    int64_t r = (int64_t)blockIdx.y * per;
    const int64_t rows_end = r + per < rows ? r + per : rows;
    const int64_t stride = 1;
    rows = rows_end;
    // two rows in flight per thread: a wave covers 1 KiB of each row, fully coalesced
    for (; r + stride < rows; r += 2 * stride) {
        const int64_t i0 = r * cols4 + c, i1 = (r + stride) * cols4 + c;
        const float4 g0 = dh[i0], g1 = dh[i1];
        float4 z0 = g0, z1 = g1;
        if (RELU) {
            const float4 a0 = h[i0], a1 = h[i1];
            z0.x = a0.x > 0.f ? g0.x : 0.f; z0.y = a0.y > 0.f ? g0.y : 0.f;
            z0.z = a0.z > 0.f ? g0.z : 0.f; z0.w = a0.w > 0.f ? g0.w : 0.f;
            z1.x = a1.x > 0.f ? g1.x : 0.f; z1.y = a1.y > 0.f ? g1.y : 0.f;
            z1.z = a1.z > 0.f ? g1.z : 0.f; z1.w = a1.w > 0.f ? g1.w : 0.f;
            dh[i0] = z0; dh[i1] = z1;
        }
        s.x += z0.x; s.y += z0.y; s.z += z0.z; s.w += z0.w;
        s.x += z1.x; s.y += z1.y; s.z += z1.z; s.w += z1.w;
    }
    if (r < rows) {
        const int64_t i0 = r * cols4 + c;
        const float4 g0 = dh[i0];
        float4 z0 = g0;
        if (RELU) {
            const float4 a0 = h[i0];
            z0.x = a0.x > 0.f ? g0.x : 0.f; z0.y = a0.y > 0.f ? g0.y : 0.f;
            z0.z = a0.z > 0.f ? g0.z : 0.f; z0.w = a0.w > 0.f ? g0.w : 0.f;
            dh[i0] = z0;
        }
        s.x += z0.x; s.y += z0.y; s.z += z0.z; s.w += z0.w;
    }
    partial[(int64_t)blockIdx.y * cols4 + c] = s;
}

// partial [nblk][cols4] float4 -> out [cols4] float4.  A workgroup owns 8 float4 columns; 32 row
// slices per column sum nblk/32 partials each (fixed order), then meet in LDS (fixed order).
__global__ __launch_bounds__(kThreads) void colsum_finish_kernel(
    const float4 *__restrict__ partial, int nblk, int cols4, float4 *__restrict__ out)
{
    __shared__ float4 red[32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols4)
        for (int b = sl; b < nblk; b += 32) {
            const float4 v = partial[(int64_t)b * cols4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < cols4) {
        float4 t = red[0][cl];
        for (int i = 1; i < 32; ++i) { t.x += red[i][cl].x; t.y += red[i][cl].y; t.z += red[i][cl].z; t.w += red[i][cl].w; }
        out[c] = t;
    }
}

// Column sums of a SHORT matrix (rows <= kSmallRows: the decoder's [4400, 256] gradients) in one
// launch: a workgroup owns 4 float4 columns, 64 row slices per column (fixed order), LDS tree.
constexpr int kSmallRows = 8192;

__global__ __launch_bounds__(kThreads) void colsum_small_kernel(
    const float4 *__restrict__ x, int rows, int cols4, float4 *__restrict__ out)
{
    __shared__ float4 red[64][4];
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols4) {
#pragma unroll 4
        for (int r = sl; r < rows; r += 64) {
            const float4 v = x[(int64_t)r * cols4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < cols4) {
        float4 t = red[0][cl];
        for (int i = 1; i < 64; ++i) { t.x += red[i][cl].x; t.y += red[i][cl].y; t.z += red[i][cl].z; t.w += red[i][cl].w; }
        out[c] = t;
    }
}

}  // namespace

extern "C" int64_t datr_relu_bwd_bias_partial_rows(int64_t rows) {
    // enough row-slices to fill the chip (2 column blocks x 512 = 1024 workgroups at cols 2048)
    // without making the per-column finishing sum long; small inputs: ~32 rows per slice
    int64_t n = rows / 32;
    n = n < 1 ? 1 : (n > 512 ? 512 : n);
    return n;
}

extern "C" int datr_relu_bwd_bias_f32(float *dh, const float *h, int64_t rows, int64_t cols,
                                      float *partial, float *db, void *stream) {
    if (rows < 0 || cols <= 0 || (cols & 3) || cols > (1 << 20)) return DATR_EINVAL;
    if (!db) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        return hipMemsetAsync(db, 0, (size_t)cols * sizeof(float), st) == hipSuccess ? DATR_OK
                                                                                    : DATR_ELAUNCH;
    }
    if (!dh || !h || !partial) return DATR_EINVAL;
    const int cols4 = (int)(cols / 4);
    const int nblk = (int)datr_relu_bwd_bias_partial_rows(rows);
    dim3 grid((unsigned)((cols4 + kThreads - 1) / kThreads), (unsigned)nblk);
    hipLaunchKernelGGL(relu_bwd_bias_kernel<true>, grid, dim3(kThreads), 0, st,
                       reinterpret_cast<float4 *>(dh), reinterpret_cast<const float4 *>(h), rows,
                       cols4, reinterpret_cast<float4 *>(partial));
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((cols4 + 7) / 8)), dim3(kThreads), 0, st,
                       reinterpret_cast<const float4 *>(partial), nblk, cols4,
                       reinterpret_cast<float4 *>(db));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

// Column sums of a rows x cols fp32 matrix (the bias gradient of a linear layer), same two-stage
// deterministic scheme; `partial` = datr_relu_bwd_bias_partial_rows(rows) * cols floats.
extern "C" int datr_colsum_f32(const float *x, int64_t rows, int64_t cols, float *partial,
                               float *out, void *stream) {
    if (rows < 0 || cols <= 0 || (cols & 3) || cols > (1 << 20)) return DATR_EINVAL;
    if (!out) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0)
        return hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), st) == hipSuccess ? DATR_OK
                                                                                     : DATR_ELAUNCH;
    if (!x || !partial) return DATR_EINVAL;
    const int cols4 = (int)(cols / 4);
    if (rows <= kSmallRows) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)((cols4 + 3) / 4)), dim3(kThreads), 0, st,
                           reinterpret_cast<const float4 *>(x), (int)rows, cols4,
                           reinterpret_cast<float4 *>(out));
        return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
    }
    const int nblk = (int)datr_relu_bwd_bias_partial_rows(rows);
    dim3 grid((unsigned)((cols4 + kThreads - 1) / kThreads), (unsigned)nblk);
    hipLaunchKernelGGL(relu_bwd_bias_kernel<false>, grid, dim3(kThreads), 0, st,
                       reinterpret_cast<float4 *>(const_cast<float *>(x)),
                       static_cast<const float4 *>(nullptr), rows, cols4,
                       reinterpret_cast<float4 *>(partial));
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((cols4 + 7) / 8)), dim3(kThreads), 0, st,
                       reinterpret_cast<const float4 *>(partial), nblk, cols4,
                       reinterpret_cast<float4 *>(out));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
