// ema.hip -- the EMA teacher update as ONE multi-tensor launch: the reference's
//     for k, v in ema.state_dict().items():  v *= d;  v += (1. - d) * msd[k].detach()
// (/root/reference/models/dino/EMA.py:46-50, CosineEMA :123-128) over the 640 state_dict keys.
// The reference walks KEYS, so a tensor stored under several keys (the detection heads shared by
// the six decoder layers, 12 keys each) is updated once per key; here every distinct tensor is one
// table entry with its repeat count and the update is replayed that many times in registers --
// v <- fl(fl(v * d) + fl((1 - d) * m)) each time, products rounded before the sum (no fused
// multiply-add), i.e. the reference's arithmetic bit for bit -- with one read and one write per
// element (the key walk reads and writes an aliased tensor 12 times, in 2 kernels per key).
// A workgroup takes one 4096-element piece of one tensor; the piece table (which tensor, where)
// and the tensor table are built once by the host side (datr_amd/ema.py) and stay on the device.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

#pragma clang fp contract(off)

constexpr int kPiece = 4096, kThreads = 256;

__device__ __forceinline__ float step(float v, float m, float d, float omd) {
#pragma clang fp contract(off)
    const float a = v * d;
    const float b = omd * m;
    return a + b;
}

__global__ __launch_bounds__(kThreads) void ema_update(const datr_ema_tensor *__restrict__ tensors,
                                                       const datr_ema_piece *__restrict__ pieces, float d, float omd)
{
    const datr_ema_piece pc = pieces[blockIdx.x];
    const datr_ema_tensor t = tensors[pc.tensor];
    float *dst = t.dst + pc.offset;
    const float *src = t.src + pc.offset;
    const int64_t n = t.numel - pc.offset < kPiece ? t.numel - pc.offset : kPiece;
    const int reps = t.repeats;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += kThreads) {
            float4 v = reinterpret_cast<float4 *>(dst)[i];
            const float4 m = reinterpret_cast<const float4 *>(src)[i];
            for (int r = 0; r < reps; ++r) {
                v.x = step(v.x, m.x, d, omd); v.y = step(v.y, m.y, d, omd);
                v.z = step(v.z, m.z, d, omd); v.w = step(v.w, m.w, d, omd);
            }
            reinterpret_cast<float4 *>(dst)[i] = v;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) {
            float v = dst[i];
            for (int r = 0; r < reps; ++r) v = step(v, src[i], d, omd);
            dst[i] = v;
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += kThreads) {
            float v = dst[i];
            for (int r = 0; r < reps; ++r) v = step(v, src[i], d, omd);
            dst[i] = v;
        }
    }
}

}  // namespace

extern "C" int64_t datr_ema_piece_elements(void) { return kPiece; }

extern "C" int datr_ema_update_f32(const datr_ema_tensor *tensors, const datr_ema_piece *pieces, int64_t npieces,
                                   double decay, void *stream) {
    if (npieces == 0) return DATR_OK;
    if (!tensors || !pieces || npieces < 0) return DATR_EINVAL;
    if (npieces > 0x7fffffff) return DATR_EUNSUPPORTED;
    // the reference's scalars: `v *= d` and `(1. - d) * m` with d a Python float -- the difference is
    // taken in double, each factor then rounded to the tensor's float32
    hipLaunchKernelGGL(ema_update, dim3((unsigned)npieces), dim3(kThreads), 0, static_cast<hipStream_t>(stream), tensors,
                       pieces, (float)decay, (float)(1.0 - decay));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
