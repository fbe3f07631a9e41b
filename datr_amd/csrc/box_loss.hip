// box_loss.hip -- the box regression losses of SetCriterion for all matched pairs of all
// prediction sets in ONE launch each way.
//
// `loss_boxes` (/root/reference/models/dino/dino.py:553-577): for every matched (prediction,
// ground-truth) pair, L1 between the cxcywh boxes and 1 - GIoU of the xyxy boxes
// (/root/reference/util/box_ops.py:9-63, with its 1e-6 terms), summed per prediction set; plus the
// logging-only xy / hw splits of the L1 (dino.py:571-574).  Written with torch ops this is ~55
// launches forward and ~70 backward per loss family (pairs are few: 140 matching + 1200
// denoising pairs per step), i.e. ~250 launches of 4-8 us per step.
//
// Forward: one workgroup; per-pair terms go to LDS, then each (term, set) sum is accumulated by
// one wave in a fixed order -- deterministic (the torch version scatters with atomics).
// Backward: closed form per pair, following autograd's conventions exactly: |x|' = sign(x),
// clamp(min=0)' = [x >= 0], max/min split the gradient in half on ties.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr float kEps = 1e-6f;

struct Pair {
    float l1[4];
    float x0, y0, x1, y1, X0, Y0, X1, Y1;          // xyxy of prediction / ground truth
    float area_s, area_t, iw_raw, ih_raw, hw_raw, hh_raw, inter, uni, hull;
};

__device__ __forceinline__ Pair eval_pair(const float4 s, const float4 t) {
    Pair p;
    p.l1[0] = fabsf(s.x - t.x); p.l1[1] = fabsf(s.y - t.y);
    p.l1[2] = fabsf(s.z - t.z); p.l1[3] = fabsf(s.w - t.w);
    p.x0 = s.x - 0.5f * s.z; p.y0 = s.y - 0.5f * s.w; p.x1 = s.x + 0.5f * s.z; p.y1 = s.y + 0.5f * s.w;
    p.X0 = t.x - 0.5f * t.z; p.Y0 = t.y - 0.5f * t.w; p.X1 = t.x + 0.5f * t.z; p.Y1 = t.y + 0.5f * t.w;
    p.area_s = (p.x1 - p.x0) * (p.y1 - p.y0);
    p.area_t = (p.X1 - p.X0) * (p.Y1 - p.Y0);
    p.iw_raw = fminf(p.x1, p.X1) - fmaxf(p.x0, p.X0);
    p.ih_raw = fminf(p.y1, p.Y1) - fmaxf(p.y0, p.Y0);
    p.inter = fmaxf(p.iw_raw, 0.f) * fmaxf(p.ih_raw, 0.f);
    p.uni = p.area_s + p.area_t - p.inter;
    p.hw_raw = fmaxf(p.x1, p.X1) - fminf(p.x0, p.X0);
    p.hh_raw = fmaxf(p.y1, p.Y1) - fminf(p.y0, p.Y0);
    p.hull = fmaxf(p.hw_raw, 0.f) * fmaxf(p.hh_raw, 0.f);
    return p;
}

__device__ __forceinline__ float giou_of(const Pair &p) {
    const float iou = p.inter / (p.uni + kEps);
    return iou - (p.hull - p.uni) / (p.hull + kEps);
}

// sums: [4][G] = (sum L1, sum (1 - GIoU), sum L1 of xy, sum L1 of wh) per prediction set
__global__ __launch_bounds__(kThreads) void box_loss_fwd(const float4 *__restrict__ src,
                                                         const float4 *__restrict__ tgt,
                                                         const int64_t *__restrict__ group, int P, int G,
                                                         float *__restrict__ sums)
{
    extern __shared__ float term[];                 // [4][P] floats + [P] set indices
    int *grp = reinterpret_cast<int *>(term + 4 * P);
    for (int i = threadIdx.x; i < P; i += kThreads) {
        const Pair p = eval_pair(src[i], tgt[i]);
        const float xy = p.l1[0] + p.l1[1], hw = p.l1[2] + p.l1[3];
        term[i] = xy + hw;
        term[P + i] = 1.f - giou_of(p);
        term[2 * P + i] = xy;
        term[3 * P + i] = hw;
        grp[i] = (int)group[i];
    }
    __syncthreads();
    // one wave per (term, set): lane-strided partial sums, then a fixed xor tree
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < 4 * G; o += kThreads / 64) {
        const int k = o / G, g = o - k * G;
        float s = 0.f;
        for (int i = lane; i < P; i += 64)
            if (grp[i] == g) s += term[k * P + i];
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) sums[o] = s;
    }
}

// two-way max / min with autograd's tie rule: weight of the FIRST argument's gradient
__device__ __forceinline__ float wmax(float a, float b) { return a > b ? 1.f : a == b ? 0.5f : 0.f; }
__device__ __forceinline__ float wmin(float a, float b) { return a < b ? 1.f : a == b ? 0.5f : 0.f; }

__global__ __launch_bounds__(kThreads) void box_loss_bwd(const float4 *__restrict__ src,
                                                         const float4 *__restrict__ tgt,
                                                         const int64_t *__restrict__ group,
                                                         const float *__restrict__ d_l1,
                                                         const float *__restrict__ d_giou, int P,
                                                         float4 *__restrict__ d_src)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= P) return;
    const float4 s = src[i], t = tgt[i];
    const Pair p = eval_pair(s, t);
    const int g = (int)group[i];
    const float gl = d_l1[g];
    const float gg = -d_giou[g];                    // d loss / d giou  (loss term = 1 - giou)
    auto sgn = [](float v) { return v > 0.f ? 1.f : v < 0.f ? -1.f : 0.f; };
    float dcx = gl * sgn(s.x - t.x), dcy = gl * sgn(s.y - t.y);
    float dw = gl * sgn(s.z - t.z), dh = gl * sgn(s.w - t.w);

    // giou = I / (U + eps) - (H - U) / (H + eps)
    const float Ue = p.uni + kEps, He = p.hull + kEps;
    const float dI0 = gg / Ue;
    const float dU = gg * (-p.inter / (Ue * Ue) + 1.f / He);
    const float dH = gg * (-(He - (p.hull - p.uni)) / (He * He));
    const float dI = dI0 - dU;                      // U = As + At - I
    const float dAs = dU;
    float dx0 = 0.f, dy0 = 0.f, dx1 = 0.f, dy1 = 0.f;
    // I = clamp(iw_raw, 0) * clamp(ih_raw, 0)
    const float iw = fmaxf(p.iw_raw, 0.f), ih = fmaxf(p.ih_raw, 0.f);
    const float diw = p.iw_raw >= 0.f ? dI * ih : 0.f, dih = p.ih_raw >= 0.f ? dI * iw : 0.f;
    dx1 += diw * wmin(p.x1, p.X1);  dx0 -= diw * wmax(p.x0, p.X0);
    dy1 += dih * wmin(p.y1, p.Y1);  dy0 -= dih * wmax(p.y0, p.Y0);
    // H = clamp(hw_raw, 0) * clamp(hh_raw, 0)
    const float hw = fmaxf(p.hw_raw, 0.f), hh = fmaxf(p.hh_raw, 0.f);
    const float dhw = p.hw_raw >= 0.f ? dH * hh : 0.f, dhh = p.hh_raw >= 0.f ? dH * hw : 0.f;
    dx1 += dhw * wmax(p.x1, p.X1);  dx0 -= dhw * wmin(p.x0, p.X0);
    dy1 += dhh * wmax(p.y1, p.Y1);  dy0 -= dhh * wmin(p.y0, p.Y0);
    // As = (x1 - x0) * (y1 - y0)
    const float bw = p.x1 - p.x0, bh = p.y1 - p.y0;
    dx1 += dAs * bh;  dx0 -= dAs * bh;  dy1 += dAs * bw;  dy0 -= dAs * bw;
    // x0 = cx - w / 2, x1 = cx + w / 2, ...
    dcx += dx0 + dx1;  dw += 0.5f * (dx1 - dx0);
    dcy += dy0 + dy1;  dh += 0.5f * (dy1 - dy0);
    d_src[i] = make_float4(dcx, dcy, dw, dh);
}

}  // namespace

extern "C" int datr_box_loss_forward_f32(const float *src, const float *tgt, const int64_t *group,
                                         int64_t P, int64_t G, float *sums, void *stream) {
    if (P < 0 || G <= 0) return DATR_EINVAL;
    if (!sums || (P > 0 && (!src || !tgt || !group))) return DATR_EINVAL;
    if (P > DATR_BOX_LOSS_MAX_PAIRS || G > 64) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(box_loss_fwd, dim3(1), dim3(kThreads), (size_t)(5 * P * sizeof(float)),
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<const float4 *>(tgt), group, (int)P, (int)G, sums);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_box_loss_backward_f32(const float *src, const float *tgt, const int64_t *group,
                                          const float *d_l1, const float *d_giou, int64_t P,
                                          float *d_src, void *stream) {
    if (P < 0) return DATR_EINVAL;
    if (P == 0) return DATR_OK;
    if (!src || !tgt || !group || !d_l1 || !d_giou || !d_src) return DATR_EINVAL;
    if (P > DATR_BOX_LOSS_MAX_PAIRS) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(box_loss_bwd, dim3((unsigned)((P + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<const float4 *>(tgt), group, d_l1, d_giou, (int)P,
                       reinterpret_cast<float4 *>(d_src));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
