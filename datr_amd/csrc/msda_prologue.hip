// msda_prologue.hip -- sampling locations and attention weights of MSDeformAttn from the merged
// query projection, one launch each way.
//
// `MSDeformAttn.forward` (/root/reference/models/dino/ops/modules/ms_deform_attn.py:96-117):
//   attention_weights = softmax over the L*P = 16 (level, point) logits of a head
//   sampling_locations = ref[..., :2] + offsets                               (2-d references;
//                        the division by (W_l, H_l) is folded into the projection weights)
//                      = ref[..., :2] + offsets / P * ref[..., 2:] * 0.5      (4-d references)
// The query projection arrives as ONE [rows, M*L*P*3] matrix (offsets | logits, msda.py); with
// torch ops the split, the softmax, the location arithmetic and their backward are 7-10 launches
// per layer and direction (6 encoder + 6 decoder layers).  Here a lane owns one (row, head,
// level): 4 points = 2 float4 of offsets + 1 float4 of logits; the softmax over the head's 16
// entries is a reduction over 4 registers and the 4 lanes of a quad (DPP).  Reference points
// carry no gradient on this path (encoder grid / detached decoder boxes).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

#ifndef DATR_PROLOGUE_NT
#define DATR_PROLOGUE_NT 1
#endif

namespace {

constexpr int kM = 8, kL = 4, kP = 4;            // heads, levels, points: the DINO configuration
constexpr int kOff = kM * kL * kP * 2;           // 256 offset columns
constexpr int kCols = kOff + kM * kL * kP;       // 384

__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    return v;
}

template <int REF>       // 2 or 4 reference coordinates
__global__ __launch_bounds__(256) void prologue_fwd(const float *__restrict__ both,
                                                    const float *__restrict__ ref, long rows,
                                                    float *__restrict__ loc, float *__restrict__ attn)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (row, head, level)
    const bool ok = i < rows * (kM * kL);
    const long ic = ok ? i : 0;
    const long r = ic / (kM * kL);
    const int ml = (int)(ic - r * (kM * kL)), l = ml & (kL - 1);
    const float *row = both + r * kCols;
    const float4 o0 = *reinterpret_cast<const float4 *>(row + ml * 8);
    const float4 o1 = *reinterpret_cast<const float4 *>(row + ml * 8 + 4);
    const float4 lg = *reinterpret_cast<const float4 *>(row + kOff + ml * 4);
    // softmax over the head's 16 logits = 4 registers x the 4 lanes (levels) of this quad
    const float mx = quad_max(fmaxf(fmaxf(lg.x, lg.y), fmaxf(lg.z, lg.w)));
    const float4 e = make_float4(__expf(lg.x - mx), __expf(lg.y - mx), __expf(lg.z - mx), __expf(lg.w - mx));
    const float inv = 1.f / quad_sum((e.x + e.y) + (e.z + e.w));
    float4 a0, a1;
    if (REF == 2) {
        const float2 c = *reinterpret_cast<const float2 *>(ref + (r * kL + l) * 2);
        a0 = make_float4(c.x + o0.x, c.y + o0.y, c.x + o0.z, c.y + o0.w);
        a1 = make_float4(c.x + o1.x, c.y + o1.y, c.x + o1.z, c.y + o1.w);
    } else {
        const float4 c = *reinterpret_cast<const float4 *>(ref + (r * kL + l) * 4);
        a0 = make_float4(c.x + o0.x / kP * c.z * 0.5f, c.y + o0.y / kP * c.w * 0.5f,
                         c.x + o0.z / kP * c.z * 0.5f, c.y + o0.w / kP * c.w * 0.5f);
        a1 = make_float4(c.x + o1.x / kP * c.z * 0.5f, c.y + o1.y / kP * c.w * 0.5f,
                         c.x + o1.z / kP * c.z * 0.5f, c.y + o1.w / kP * c.w * 0.5f);
    }
    if (ok) {
        // streamed once by the MSDA kernel that follows: non-temporal, so that they do not push the
        // value tensor (gathered ~16 times per element) out of L2 / the infinity cache
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v0 = {a0.x, a0.y, a0.z, a0.w}, v1 = {a1.x, a1.y, a1.z, a1.w};
        const f4 v2 = {e.x * inv, e.y * inv, e.z * inv, e.w * inv};
#if DATR_PROLOGUE_NT
        __builtin_nontemporal_store(v0, reinterpret_cast<f4 *>(loc + i * 8));
        __builtin_nontemporal_store(v1, reinterpret_cast<f4 *>(loc + i * 8 + 4));
        __builtin_nontemporal_store(v2, reinterpret_cast<f4 *>(attn + i * 4));
#else
        *reinterpret_cast<f4 *>(loc + i * 8) = v0;
        *reinterpret_cast<f4 *>(loc + i * 8 + 4) = v1;
        *reinterpret_cast<f4 *>(attn + i * 4) = v2;
#endif
    }
}

template <int REF>
__global__ __launch_bounds__(256) void prologue_bwd(const float *__restrict__ d_loc,
                                                    const float *__restrict__ d_attn,
                                                    const float *__restrict__ attn,
                                                    const float *__restrict__ ref, long rows,
                                                    float *__restrict__ d_both)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < rows * (kM * kL);
    const long ic = ok ? i : 0;
    const long r = ic / (kM * kL);
    const int ml = (int)(ic - r * (kM * kL)), l = ml & (kL - 1);
    float4 g0 = *reinterpret_cast<const float4 *>(d_loc + ic * 8);
    float4 g1 = *reinterpret_cast<const float4 *>(d_loc + ic * 8 + 4);
    const float4 ga = *reinterpret_cast<const float4 *>(d_attn + ic * 4);
    const float4 a = *reinterpret_cast<const float4 *>(attn + ic * 4);
    // softmax backward: a * (g - sum_j a_j g_j) over the head's 16 entries
    const float dot = quad_sum((a.x * ga.x + a.y * ga.y) + (a.z * ga.z + a.w * ga.w));
    const float4 dl = make_float4(a.x * (ga.x - dot), a.y * (ga.y - dot), a.z * (ga.z - dot), a.w * (ga.w - dot));
    if (REF == 4) {          // d/d offset of (offset / P * wh * 0.5), in autograd's order
        const float4 c = *reinterpret_cast<const float4 *>(ref + (r * kL + l) * 4);
        g0 = make_float4(g0.x * 0.5f * c.z / kP, g0.y * 0.5f * c.w / kP, g0.z * 0.5f * c.z / kP, g0.w * 0.5f * c.w / kP);
        g1 = make_float4(g1.x * 0.5f * c.z / kP, g1.y * 0.5f * c.w / kP, g1.z * 0.5f * c.z / kP, g1.w * 0.5f * c.w / kP);
    }
    if (ok) {
        float *row = d_both + r * kCols;
        *reinterpret_cast<float4 *>(row + ml * 8) = g0;
        *reinterpret_cast<float4 *>(row + ml * 8 + 4) = g1;
        *reinterpret_cast<float4 *>(row + kOff + ml * 4) = dl;
    }
}

}  // namespace

extern "C" int datr_msda_prologue_forward_f32(const float *both, const float *ref, int64_t rows,
                                              int64_t ref_dim, float *loc, float *attn, void *stream) {
    if (rows < 0 || (ref_dim != 2 && ref_dim != 4)) return DATR_EINVAL;
    if (rows == 0) return DATR_OK;
    if (!both || !ref || !loc || !attn) return DATR_EINVAL;
    const int64_t n = rows * kM * kL;
    if (n > 0x7fffffffLL * 128) return DATR_EUNSUPPORTED;
    dim3 grid((unsigned)((n + 255) / 256));
    if (ref_dim == 2)
        hipLaunchKernelGGL(prologue_fwd<2>, grid, dim3(256), 0, (hipStream_t)stream, both, ref, (long)rows, loc, attn);
    else
        hipLaunchKernelGGL(prologue_fwd<4>, grid, dim3(256), 0, (hipStream_t)stream, both, ref, (long)rows, loc, attn);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_msda_prologue_backward_f32(const float *d_loc, const float *d_attn, const float *attn,
                                               const float *ref, int64_t rows, int64_t ref_dim,
                                               float *d_both, void *stream) {
    if (rows < 0 || (ref_dim != 2 && ref_dim != 4)) return DATR_EINVAL;
    if (rows == 0) return DATR_OK;
    if (!d_loc || !d_attn || !attn || !ref || !d_both) return DATR_EINVAL;
    const int64_t n = rows * kM * kL;
    if (n > 0x7fffffffLL * 128) return DATR_EUNSUPPORTED;
    dim3 grid((unsigned)((n + 255) / 256));
    if (ref_dim == 2)
        hipLaunchKernelGGL(prologue_bwd<2>, grid, dim3(256), 0, (hipStream_t)stream, d_loc, d_attn, attn, ref, (long)rows, d_both);
    else
        hipLaunchKernelGGL(prologue_bwd<4>, grid, dim3(256), 0, (hipStream_t)stream, d_loc, d_attn, attn, ref, (long)rows, d_both);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

// `value.masked_fill(input_padding_mask[..., None], 0)` (ms_deform_attn.py:101-102) and its backward,
// in place: only the rows of padded tokens are touched (one mask byte read per 16 output bytes, a
// store for masked rows only) instead of a read + write pass over the whole [N, S, C] tensor.
namespace {
__global__ __launch_bounds__(256) void zero_rows_kernel(float4 *__restrict__ x, const uint8_t *__restrict__ mask,
                                                        long rows, int cols4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols4) return;
    if (mask[i / cols4]) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" int datr_zero_rows_f32(float *x, const uint8_t *mask, int64_t rows, int64_t cols, void *stream) {
    if (rows < 0 || cols <= 0 || (cols & 3)) return DATR_EINVAL;
    if (rows == 0) return DATR_OK;
    if (!x || !mask || ((uintptr_t)x & 15)) return DATR_EINVAL;
    const int64_t n = rows * (cols / 4);
    if (n > 0x7fffffffLL * 256) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4 *>(x), mask, (long)rows, (int)(cols / 4));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
