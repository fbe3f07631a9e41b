// subsample.hip -- the even pixels of an NHWC tensor, and the adjoint: what a 1x1 / stride-2
// convolution reads before its GEMM (the downsample branch of the first bottleneck of layer2-4,
// /root/reference/models/dino/backbone.py:109-128 -> torchvision Bottleneck.downsample).
//   gather   y[n, oy, ox, :] = x[n, 2 oy, 2 ox, :]
//   scatter  dx[n, y, x, :]  = dy[n, y / 2, x / 2, :] on even (y, x), 0 elsewhere (ONE pass over dx:
//            no separate zero fill)
// Streaming kernels, one float4 per thread and step; a pixel's channels are contiguous, so every
// access is a run of C * 4 bytes.  (ATen's generic strided copy took 100-190 us per call here.)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

__global__ __launch_bounds__(256) void even_gather(const float4 *__restrict__ x, int H, int W, int Ho, int Wo, int C4,
                                                   long total, float4 *__restrict__ y)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const long n = p / Ho;
        y[i] = x[((n * H + 2 * oy) * W + 2 * ox) * C4 + c];
    }
}

__global__ __launch_bounds__(256) void even_scatter(const float4 *__restrict__ dy, int H, int W, int Ho, int Wo, int C4,
                                                    long total, float4 *__restrict__ dx)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const long n = p / H;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!((yy | xx) & 1)) v = dy[((n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C4 + c];
        dx[i] = v;
    }
}

unsigned blocks_for(long total) {
    const long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

int datr_even_pixels_nhwc_f32(const float *x, int64_t N, int64_t H, int64_t W, int64_t C, float *y, void *stream)
{
    if (!x || !y || N < 0 || H < 1 || W < 1 || C < 1) return DATR_EINVAL;
    if (C % 4) return DATR_EUNSUPPORTED;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    const long total = (long)N * Ho * Wo * (C / 4);
    if (total == 0) return DATR_OK;
    hipLaunchKernelGGL(even_gather, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), (int)H, (int)W, Ho, Wo, (int)(C / 4), total,
                       reinterpret_cast<float4 *>(y));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

int datr_even_pixels_scatter_nhwc_f32(const float *dy, int64_t N, int64_t H, int64_t W, int64_t C, float *dx,
                                      void *stream)
{
    if (!dy || !dx || N < 0 || H < 1 || W < 1 || C < 1) return DATR_EINVAL;
    if (C % 4) return DATR_EUNSUPPORTED;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return DATR_OK;
    hipLaunchKernelGGL(even_scatter, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(dy), (int)H, (int)W, Ho, Wo, (int)(C / 4), total,
                       reinterpret_cast<float4 *>(dx));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

}  // extern "C"
