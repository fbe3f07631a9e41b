// match_cost.hip -- the Hungarian matcher's cost matrix in one launch, written in the layout
// the device assignment solver reads (targets x queries).
//
// `HungarianMatcher.forward` (/root/reference/models/dino/matcher.py:48-88):
//   C = w_bbox * cdist_1(box_q, box_t) + w_class * (pos(p_{q, c_t}) - neg(p_{q, c_t}))
//       + w_giou * (-GIoU(xyxy(box_q), xyxy(box_t)))
// with p = sigmoid(logit), neg = (1 - alpha) p^2 (-log(1 - p + 1e-8)),
// pos = alpha (1 - p)^2 (-log(p + 1e-8)) and box_ops.py's GIoU (1e-6 terms).  With torch ops
// that is ~73 small launches per step (sigmoid, pow, log, gather, cdist, 2 x cxcywh->xyxy, IoU,
// hull, ...) on [G*B*Q, sum T] elements plus the transposition the solver wants; here one
// thread computes one (query, target) entry with the same operations in the same order.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

__global__ __launch_bounds__(256) void match_cost_kernel(
    const float *__restrict__ logits, const float4 *__restrict__ boxes,
    const int64_t *__restrict__ tgt_ids, const float4 *__restrict__ tgt_boxes, int sets, int nq,
    int T, int C, float w_class, float w_bbox, float w_giou, float alpha,
    float *__restrict__ cost_t, int *__restrict__ boxes_ok)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)sets * T * nq;
    if (i >= total) return;
    const int q = (int)(i % nq);
    const int t = (int)((i / nq) % T);
    const int s = (int)(i / ((long)nq * T));
    const long row = (long)s * nq + q;
    const float4 b = boxes[row], g = tgt_boxes[t];
    const float x = logits[row * C + tgt_ids[t]];

    const float p = 1.f / (1.f + expf(-x));
    const float neg = ((1.f - alpha) * (p * p)) * (-logf(1.f - p + 1e-8f));
    const float pos = (alpha * ((1.f - p) * (1.f - p))) * (-logf(p + 1e-8f));
    const float cost_class = pos - neg;
    const float cost_bbox = fabsf(b.x - g.x) + fabsf(b.y - g.y) + fabsf(b.z - g.z) + fabsf(b.w - g.w);

    const float x0 = b.x - 0.5f * b.z, y0 = b.y - 0.5f * b.w, x1 = b.x + 0.5f * b.z, y1 = b.y + 0.5f * b.w;
    const float X0 = g.x - 0.5f * g.z, Y0 = g.y - 0.5f * g.w, X1 = g.x + 0.5f * g.z, Y1 = g.y + 0.5f * g.w;
    if (!(x1 >= x0 && y1 >= y0 && X1 >= X0 && Y1 >= Y0)) *boxes_ok = 0;     // box_ops.py:52-53
    const float area1 = (x1 - x0) * (y1 - y0), area2 = (X1 - X0) * (Y1 - Y0);
    const float iw = fmaxf(fminf(x1, X1) - fmaxf(x0, X0), 0.f);
    const float ih = fmaxf(fminf(y1, Y1) - fmaxf(y0, Y0), 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / (uni + 1e-6f);
    const float hw = fmaxf(fmaxf(x1, X1) - fminf(x0, X0), 0.f);
    const float hh = fmaxf(fmaxf(y1, Y1) - fminf(y0, Y0), 0.f);
    const float hull = hw * hh;
    const float giou = iou - (hull - uni) / (hull + 1e-6f);

    cost_t[i] = (w_bbox * cost_bbox + w_class * cost_class) + w_giou * (-giou);
}

}  // namespace

extern "C" int datr_match_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids,
                                   const float *tgt_boxes, int64_t sets, int64_t nq, int64_t T,
                                   int64_t C, float w_class, float w_bbox, float w_giou, float alpha,
                                   float *cost_t, int32_t *boxes_ok, void *stream) {
    if (sets < 0 || nq < 0 || T < 0 || C <= 0) return DATR_EINVAL;
    const int64_t total = sets * T * nq;
    if (total == 0) return DATR_OK;
    if (!logits || !boxes || !tgt_ids || !tgt_boxes || !cost_t || !boxes_ok) return DATR_EINVAL;
    if (total > 0x7fffffffLL * 64) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, logits, reinterpret_cast<const float4 *>(boxes), tgt_ids,
                       reinterpret_cast<const float4 *>(tgt_boxes), (int)sets, (int)nq, (int)T, (int)C,
                       w_class, w_bbox, w_giou, alpha, cost_t, boxes_ok);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
