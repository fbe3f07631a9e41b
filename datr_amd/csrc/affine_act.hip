// affine_act.hip -- fused frozen-batch-norm (+ residual) (+ ReLU) for the ResNet-50 trunk, gfx950.
//
// Reference behaviour: `FrozenBatchNorm2d.forward` (/root/reference/models/dino/backbone.py:62-72)
//     scale = w * rsqrt(var + 1e-5);  shift = b - mean * scale;  y = x * scale + shift
// followed in torchvision's Bottleneck by ReLU, or by `+ identity` and ReLU after the third
// conv.  The reference runs these as 2 (mul, add) + 1 (add) + 1 (relu) element-wise kernels,
// i.e. up to 4 read+write passes over activations of up to 274 MB; here it is ONE pass:
//     y = act(x * scale[c] + shift[c] (+ res))          c = (i / inner) % C
// (`inner` = H*W for NCHW, 1 for NHWC).  Backward: g = dy * (y > 0);  dx = g * scale[c];
// d res = g.  scale / shift are buffers (never trained), so they get no gradient.
// Purely HBM-bound: float4 per lane when `inner` is a multiple of 4 (NCHW: the 4 elements share
// a channel) or when inner == 1 and C % 4 == 0 (NHWC: 4 consecutive channels, float4 scale /
// shift), scalar otherwise; grid-stride over <= 4096 workgroups.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 4096;

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_fwd4(
    const float4 *__restrict__ x, const float4 *__restrict__ res, const float *__restrict__ scale,
    const float *__restrict__ shift, int64_t n4, int C, int64_t inner4, float4 *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * kThreads) {
        const int c = (int)((i / inner4) % C);
        const float s = scale[c], b = shift[c];
        float4 v = x[i];
        v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
        if (RES) {
            const float4 r = res[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        y[i] = v;
    }
}

// NHWC (inner == 1, C % 4 == 0): a float4 holds 4 consecutive channels
template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_fwd4c(
    const float4 *__restrict__ x, const float4 *__restrict__ res, const float4 *__restrict__ scale,
    const float4 *__restrict__ shift, int64_t n4, int C4, float4 *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * kThreads) {
        const int c = (int)(i % C4);
        const float4 s = scale[c], b = shift[c];
        float4 v = x[i];
        v.x = v.x * s.x + b.x; v.y = v.y * s.y + b.y; v.z = v.z * s.z + b.z; v.w = v.w * s.w + b.w;
        if (RES) {
            const float4 r = res[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        y[i] = v;
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_bwd4c(
    const float4 *__restrict__ dy, const float4 *__restrict__ dy2, const float4 *__restrict__ y,
    const float4 *__restrict__ scale, int64_t n4, int C4, float4 *__restrict__ dx, float4 *__restrict__ dres)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * kThreads) {
        const float4 s = scale[(int)(i % C4)];
        float4 g = dy[i];
        if (dy2) {                               // two consumers of y: their gradients meet here
            const float4 h = dy2[i];
            g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
        }
        if (RELU) {
            const float4 o = y[i];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (RES) dres[i] = g;
        dx[i] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_fwd1(
    const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ scale,
    const float *__restrict__ shift, int64_t n, int C, int64_t inner, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        const int c = (int)((i / inner) % C);
        float v = x[i] * scale[c] + shift[c];
        if (RES) v += res[i];
        if (RELU) v = fmaxf(v, 0.f);
        y[i] = v;
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_bwd4(
    const float4 *__restrict__ dy, const float4 *__restrict__ dy2, const float4 *__restrict__ y,
    const float *__restrict__ scale, int64_t n4, int C, int64_t inner4, float4 *__restrict__ dx,
    float4 *__restrict__ dres)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * kThreads) {
        const int c = (int)((i / inner4) % C);
        const float s = scale[c];
        float4 g = dy[i];
        if (dy2) {
            const float4 h = dy2[i];
            g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
        }
        if (RELU) {
            const float4 o = y[i];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (RES) dres[i] = g;
        dx[i] = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void affine_bwd1(
    const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ y,
    const float *__restrict__ scale, int64_t n, int C, int64_t inner, float *__restrict__ dx,
    float *__restrict__ dres)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        const int c = (int)((i / inner) % C);
        float g = dy[i];
        if (dy2) g += dy2[i];
        if (RELU) g = y[i] > 0.f ? g : 0.f;
        if (RES) dres[i] = g;
        dx[i] = g * scale[c];
    }
}

unsigned grid_for(int64_t items) {
    int64_t b = (items + kThreads - 1) / kThreads;
    return (unsigned)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int datr_affine_act_forward_f32(const float *x, const float *res, const float *scale,
                                const float *shift, int64_t n, int64_t C, int64_t inner, int relu,
                                float *y, void *stream) {
    if (n < 0 || C <= 0 || inner <= 0) return DATR_EINVAL;
    if (n == 0) return DATR_OK;
    if (!x || !scale || !shift || !y) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = n % 4 == 0 && aligned16(x) && aligned16(y) && (!res || aligned16(res));
    const bool vec = al && inner % 4 == 0;
    const bool vecc = al && inner == 1 && C % 4 == 0 && aligned16(scale) && aligned16(shift);
#define DATR_GO(RELU, RES)                                                                        \
    if (vec)                                                                                      \
        hipLaunchKernelGGL((affine_fwd4<RELU, RES>), dim3(grid_for(n / 4)), dim3(kThreads), 0, st, \
                           (const float4 *)x, (const float4 *)res, scale, shift, n / 4, (int)C,   \
                           inner / 4, (float4 *)y);                                               \
    else if (vecc)                                                                                \
        hipLaunchKernelGGL((affine_fwd4c<RELU, RES>), dim3(grid_for(n / 4)), dim3(kThreads), 0, st, \
                           (const float4 *)x, (const float4 *)res, (const float4 *)scale,         \
                           (const float4 *)shift, n / 4, (int)(C / 4), (float4 *)y);              \
    else                                                                                          \
        hipLaunchKernelGGL((affine_fwd1<RELU, RES>), dim3(grid_for(n)), dim3(kThreads), 0, st, x,  \
                           res, scale, shift, n, (int)C, inner, y)
    if (relu && res) { DATR_GO(true, true); }
    else if (relu) { DATR_GO(true, false); }
    else if (res) { DATR_GO(false, true); }
    else { DATR_GO(false, false); }
#undef DATR_GO
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

static int affine_backward(const float *dy, const float *dy2, const float *y, const float *scale, int64_t n,
                           int64_t C, int64_t inner, int relu, float *dx, float *dres, void *stream) {
    if (n < 0 || C <= 0 || inner <= 0) return DATR_EINVAL;
    if (n == 0) return DATR_OK;
    if (!dy || !scale || !dx || (relu && !y)) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = n % 4 == 0 && aligned16(dy) && aligned16(dx) && (!relu || aligned16(y)) &&
                    (!dres || aligned16(dres)) && (!dy2 || aligned16(dy2));
    const bool vec = al && inner % 4 == 0;
    const bool vecc = al && inner == 1 && C % 4 == 0 && aligned16(scale);
#define DATR_GO(RELU, RES)                                                                        \
    if (vec)                                                                                      \
        hipLaunchKernelGGL((affine_bwd4<RELU, RES>), dim3(grid_for(n / 4)), dim3(kThreads), 0, st, \
                           (const float4 *)dy, (const float4 *)dy2, (const float4 *)y, scale, n / 4, \
                           (int)C, inner / 4, (float4 *)dx, (float4 *)dres);                      \
    else if (vecc)                                                                                \
        hipLaunchKernelGGL((affine_bwd4c<RELU, RES>), dim3(grid_for(n / 4)), dim3(kThreads), 0, st, \
                           (const float4 *)dy, (const float4 *)dy2, (const float4 *)y,            \
                           (const float4 *)scale, n / 4, (int)(C / 4), (float4 *)dx, (float4 *)dres); \
    else                                                                                          \
        hipLaunchKernelGGL((affine_bwd1<RELU, RES>), dim3(grid_for(n)), dim3(kThreads), 0, st, dy, \
                           dy2, y, scale, n, (int)C, inner, dx, dres)
    if (relu && dres) { DATR_GO(true, true); }
    else if (relu) { DATR_GO(true, false); }
    else if (dres) { DATR_GO(false, true); }
    else { DATR_GO(false, false); }
#undef DATR_GO
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

int datr_affine_act_backward_f32(const float *dy, const float *y, const float *scale, int64_t n,
                                 int64_t C, int64_t inner, int relu, float *dx, float *dres,
                                 void *stream) {
    return affine_backward(dy, nullptr, y, scale, n, C, inner, relu, dx, dres, stream);
}

int datr_affine_act_backward2_f32(const float *dy, const float *dy2, const float *y, const float *scale,
                                  int64_t n, int64_t C, int64_t inner, int relu, float *dx, float *dres,
                                  void *stream) {
    if (!dy2) return DATR_EINVAL;
    return affine_backward(dy, dy2, y, scale, n, C, inner, relu, dx, dres, stream);
}

}  // extern "C"
