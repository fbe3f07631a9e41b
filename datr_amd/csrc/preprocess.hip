// preprocess.hip -- the tail of the input pipeline on the device (first piece):
// ToTensor + Normalize + pad-to-batch + padding mask in ONE pass per image.
//
// The reference does these on the host, per image and per worker:
//   F.to_tensor   : uint8 HWC -> float CHW / 255           (datasets/da_transforms.py:250-255)
//   F.normalize   : (x - mean[c]) / std[c]                  (datasets/da_transforms.py:266-276)
//   nested_tensor_from_tensor_list: zero-padded [B,3,Hmax,Wmax] batch + bool mask, True = padding
//                                                            (util/misc.py:387-409)
// and then copies 4 bytes per sub-pixel to the device.  Here the uint8 image crosses PCIe (4x
// fewer bytes) and one kernel writes its slot of the batch -- NCHW or NHWC (the backbone's
// layout), padding zeros included, so the batch needs no memset -- and its slot of the mask.
// Arithmetic is the reference's, operation for operation in fp32 (IEEE division): bit-exact.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;

template <bool NHWC>
__global__ __launch_bounds__(kThreads) void normalize_pad_kernel(
    const uint8_t *__restrict__ img, int H, int W, float m0, float m1, float m2, float s0, float s1,
    float s2, int Hp, int Wp, float *__restrict__ out, uint8_t *__restrict__ mask)
{
    const int64_t npix = (int64_t)Hp * Wp;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < npix;
         p += (int64_t)gridDim.x * kThreads) {
        const int y = (int)(p / Wp), x = (int)(p - (int64_t)y * Wp);
        const bool inside = y < H && x < W;
        float r = 0.f, g = 0.f, b = 0.f;
        if (inside) {
            const uint8_t *s = img + ((int64_t)y * W + x) * 3;
            r = ((float)s[0] / 255.f - m0) / s0;
            g = ((float)s[1] / 255.f - m1) / s1;
            b = ((float)s[2] / 255.f - m2) / s2;
        }
        if (NHWC) {
            out[p * 3 + 0] = r; out[p * 3 + 1] = g; out[p * 3 + 2] = b;
        } else {
            out[p] = r; out[npix + p] = g; out[2 * npix + p] = b;
        }
        mask[p] = inside ? 0 : 1;
    }
}

}  // namespace

extern "C" int datr_normalize_pad_u8_f32(const uint8_t *img, int64_t H, int64_t W, const float *mean,
                                         const float *std, int64_t Hp, int64_t Wp,
                                         int channels_last, float *out, uint8_t *mask, void *stream) {
    if (H < 0 || W < 0 || Hp < H || Wp < W || Hp <= 0 || Wp <= 0 || Hp > 32768 || Wp > 32768)
        return DATR_EINVAL;
    if (!mean || !std || !out || !mask || (!img && H * W > 0)) return DATR_EINVAL;
    if (std[0] == 0.f || std[1] == 0.f || std[2] == 0.f) return DATR_EINVAL;
    const int64_t npix = Hp * Wp;
    int64_t blocks = (npix + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    if (channels_last)
        hipLaunchKernelGGL(normalize_pad_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, img, (int)H, (int)W, mean[0], mean[1], mean[2], std[0],
                           std[1], std[2], (int)Hp, (int)Wp, out, mask);
    else
        hipLaunchKernelGGL(normalize_pad_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, img, (int)H, (int)W, mean[0], mean[1], mean[2], std[0],
                           std[1], std[2], (int)Hp, (int)Wp, out, mask);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
