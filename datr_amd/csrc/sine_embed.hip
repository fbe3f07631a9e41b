// sine_embed.hip -- sine embedding of the decoder's reference boxes, one launch per decoder layer.
//
// `gen_sineembed_for_position` (/root/reference/models/dino/utils.py:138-163): for every query
// box (x, y[, w, h]) and every coordinate in the order (y, x[, w, h]), 128 features
//   p_k = coord * 2 pi / T_k,  T_k = 10000^(2 floor(k / 2) / 128),  out_k = k even ? sin(p_k) : cos(p_k)
// concatenated to 256 (2-d) or 512 (4-d) features.  The reference (and a literal torch
// restatement) spends 27 tiny launches per layer on it (mul, div, strided sin / cos, stack,
// flatten, cat); the boxes carry no gradient (detached between layers,
// /root/reference/models/dino/deformable_transformer.py:694), so one forward kernel is all it takes.
// The temperature table T_k is passed in (computed once with the same torch expression as the
// reference), so p_k is the same float as in the torch formulation: (coord * 2 pi) / T_k.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

// one thread = 4 consecutive features (sin, cos, sin, cos) of one coordinate of one box
__global__ __launch_bounds__(256) void sine_embed_kernel(const float *__restrict__ pos,
                                                         const float *__restrict__ dim_t, long rows,
                                                         int ncoord, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
    const int per_row = ncoord * 32;                                 // float4 per row
    if (i >= rows * per_row) return;
    const long r = i / per_row;
    const int f4 = (int)(i - r * per_row);
    const int c = f4 >> 5, k = (f4 & 31) * 4;
    const int src = c == 0 ? 1 : c == 1 ? 0 : c;                     // (y, x, w, h)
    const float v = pos[r * ncoord + src] * 6.283185307179586f;
    const float4 t = *reinterpret_cast<const float4 *>(dim_t + k);
    float4 o;
    o.x = sinf(v / t.x);
    o.y = cosf(v / t.y);
    o.z = sinf(v / t.z);
    o.w = cosf(v / t.w);
    reinterpret_cast<float4 *>(out)[i] = o;
}

}  // namespace

extern "C" int datr_sine_embed_f32(const float *pos, const float *dim_t, int64_t rows, int64_t ncoord,
                                   float *out, void *stream) {
    if (rows < 0 || (ncoord != 2 && ncoord != 4)) return DATR_EINVAL;
    if (rows == 0) return DATR_OK;
    if (!pos || !dim_t || !out) return DATR_EINVAL;
    const int64_t n = rows * ncoord * 32;
    if (n > 0x7fffffffLL * 256) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(sine_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, pos, dim_t, (long)rows, (int)ncoord, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
