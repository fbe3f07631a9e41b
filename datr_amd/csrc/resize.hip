// resize.hip -- bilinear resize (and horizontal flip) of uint8 HWC images on the device, bit-exact
// with Pillow: the `RandomResize` / `RandomHorizontalFlip` stage of the reference's data pipeline
// (/root/reference/datasets/da_transforms.py:62-140: `F.hflip`, `F.resize(image, size)` on PIL
// images = `Image.resize(size, BILINEAR)`; Pillow is un-vendored -- its 8-bit resampler is restated:
// separable, horizontal pass then vertical pass through an 8-bit intermediate, per output pixel a
// window [xmin, xmin + n) of the source with 22-bit fixed-point weights, accumulator started at
// 1 << 21, result (acc >> 22) clamped to 0..255).  The weights and windows are computed on the host
// in double precision exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do
// (datr_amd/input_pipeline.py::pillow_coeffs) and handed in; the flip is applied to the source
// reads of the horizontal pass (the reference flips BEFORE it resizes).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// tmp[y][xx][c] = horizontal pass of source row y (all rows)
__global__ void resize_h(const uint8_t *__restrict__ src, int H, int W, int flip,
                         const int32_t *__restrict__ bounds, const int32_t *__restrict__ kk, int ksize, int ow,
                         uint8_t *__restrict__ tmp)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= ow) return;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t *k = kk + (size_t)xx * ksize;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    const uint8_t *row = src + (size_t)y * W * 3;
    for (int x = 0; x < n; ++x) {
        const int sx = flip ? W - 1 - (xmin + x) : xmin + x;
        const uint8_t *p = row + sx * 3;
        s0 += p[0] * k[x]; s1 += p[1] * k[x]; s2 += p[2] * k[x];
    }
    uint8_t *o = tmp + ((size_t)y * ow + xx) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst[yy][xx][c] = vertical pass over tmp
__global__ void resize_v(const uint8_t *__restrict__ tmp, int ow, const int32_t *__restrict__ bounds,
                         const int32_t *__restrict__ kk, int ksize, int oh, uint8_t *__restrict__ dst)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (xx >= ow) return;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t *k = kk + (size_t)yy * ksize;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const uint8_t *p = tmp + ((size_t)(ymin + y) * ow + xx) * 3;
        s0 += p[0] * k[y]; s1 += p[1] * k[y]; s2 += p[2] * k[y];
    }
    uint8_t *o = dst + ((size_t)yy * ow + xx) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// plain copy with optional flip (a pass Pillow skips when the size does not change)
__global__ void copy_flip(const uint8_t *__restrict__ src, int W, int flip, uint8_t *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t *p = src + ((size_t)y * W + (flip ? W - 1 - x : x)) * 3;
    uint8_t *o = dst + ((size_t)y * W + x) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

}  // namespace

extern "C" int datr_resize_bilinear_u8(const uint8_t *src, int64_t H, int64_t W, int flip,
                                       const int32_t *xbounds, const int32_t *xk, int64_t ksx,
                                       const int32_t *ybounds, const int32_t *yk, int64_t ksy, int64_t oh,
                                       int64_t ow, uint8_t *tmp, uint8_t *dst, void *stream) {
    if (!src || !dst || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return DATR_EINVAL;
    if (H > 65535 || oh > 65535 || W > (1 << 24) || ow > (1 << 24)) return DATR_EUNSUPPORTED;
    const bool horiz = ow != W, vert = oh != H;            // Pillow skips a pass that would not change the size
    if ((horiz && (!xbounds || !xk || ksx <= 0)) || (vert && (!ybounds || !yk || ksy <= 0))) return DATR_EINVAL;
    if (horiz && vert && !tmp) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(256);
    if (horiz) {
        uint8_t *out = vert ? tmp : dst;
        hipLaunchKernelGGL(resize_h, dim3((unsigned)((ow + 255) / 256), (unsigned)H), blk, 0, st, src, (int)H, (int)W,
                           flip, xbounds, xk, (int)ksx, (int)ow, out);
        if (vert)
            hipLaunchKernelGGL(resize_v, dim3((unsigned)((ow + 255) / 256), (unsigned)oh), blk, 0, st, tmp, (int)ow,
                               ybounds, yk, (int)ksy, (int)oh, dst);
    } else if (vert) {
        const uint8_t *in = src;
        if (flip) {
            if (!tmp) return DATR_EINVAL;
            hipLaunchKernelGGL(copy_flip, dim3((unsigned)((W + 255) / 256), (unsigned)H), blk, 0, st, src, (int)W, 1, tmp);
            in = tmp;
        }
        hipLaunchKernelGGL(resize_v, dim3((unsigned)((ow + 255) / 256), (unsigned)oh), blk, 0, st, in, (int)ow, ybounds,
                           yk, (int)ksy, (int)oh, dst);
    } else {
        hipLaunchKernelGGL(copy_flip, dim3((unsigned)((W + 255) / 256), (unsigned)H), blk, 0, st, src, (int)W, flip, dst);
    }
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
