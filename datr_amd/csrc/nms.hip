// nms.hip -- class-aware greedy non-maximum suppression of the teacher's pseudo labels on the
// device (`rescale_pseudo_targets`, /root/reference/models/dino/self_training_utils.py:72-96, which
// calls torchvision.ops.batched_nms(boxes, scores, labels, 0.7)[:100]; torchvision is un-vendored:
// boxes of different classes are moved apart by label * (max coordinate + 1), then plain NMS in
// decreasing score order with IoU = inter / (area_a + area_b - inter) > threshold suppressing).
//
// One workgroup per problem (n <= 4096 boxes; the pseudo-label path has n <= 100): stable rank by
// score (ties: lower index first), sorted boxes / areas in LDS, then the greedy scan -- box i, if
// still alive, is kept and every thread clears the later boxes it suppresses.  All arithmetic is
// spelled with round-to-nearest intrinsics (no FMA contraction), so the kept indices are
// bit-identical to the float32 host formulation (datr_amd/self_training.py::nms_host).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBoxes = 4096;

__global__ __launch_bounds__(kThreads) void nms_kernel(const float *__restrict__ boxes,
                                                       const float *__restrict__ scores,
                                                       const int64_t *__restrict__ labels, int n, float thr,
                                                       int64_t *__restrict__ keep, int32_t *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sb = reinterpret_cast<float4 *>(smem);             // [n] sorted, class-offset boxes
    float *sarea = smem + 4 * n;                               // [n]
    int *sorder = reinterpret_cast<int *>(sarea + n);          // [n] original index of sorted slot
    int *alive = sorder + n;                                   // [n]
    __shared__ float red[kThreads];
    __shared__ int nkept;
    const int tid = threadIdx.x;

    // offset = label * (max over all coordinates + 1)
    float mx = -INFINITY;
    for (int i = tid; i < 4 * n; i += kThreads) mx = fmaxf(mx, boxes[i]);
    red[tid] = mx;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float span = __fadd_rn(red[0], 1.f);

    // stable descending rank by score
    for (int i = tid; i < n; i += kThreads) {
        const float si = scores[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = scores[j];
            rank += (sj > si) || (sj == si && j < i);
        }
        const float off = __fmul_rn((float)labels[i], span);
        const float4 b = reinterpret_cast<const float4 *>(boxes)[i];
        const float4 o = make_float4(__fadd_rn(b.x, off), __fadd_rn(b.y, off), __fadd_rn(b.z, off),
                                     __fadd_rn(b.w, off));
        sb[rank] = o;
        sarea[rank] = __fmul_rn(__fsub_rn(o.z, o.x), __fsub_rn(o.w, o.y));
        sorder[rank] = i;
        alive[rank] = 1;
    }
    if (tid == 0) nkept = 0;
    __syncthreads();

    for (int i = 0; i < n; ++i) {
        if (alive[i]) {                                        // uniform: read after the barrier below
            const float4 a = sb[i];
            const float aa = sarea[i];
            for (int j = i + 1 + tid; j < n; j += kThreads) {
                const float4 b = sb[j];
                const float w = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
                const float h = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
                const float inter = __fmul_rn(w, h);
                const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa, sarea[j]), inter));
                if (iou > thr) alive[j] = 0;
            }
            if (tid == 0) keep[nkept++] = sorder[i];
        }
        __syncthreads();
    }
    if (tid == 0) *count = nkept;
}

}  // namespace

extern "C" int datr_nms_f32(const float *boxes, const float *scores, const int64_t *labels, int64_t n,
                            float iou_threshold, int64_t *keep, int32_t *count, void *stream) {
    if (n < 0 || !count) return DATR_EINVAL;
    if (n > kMaxBoxes) return DATR_EUNSUPPORTED;
    if (n > 0 && (!boxes || !scores || !labels || !keep)) return DATR_EINVAL;
    const size_t lds = (size_t)n * (4 + 1 + 1 + 1) * sizeof(float);
    // above 64 KiB of dynamic LDS (n > 2340) a launch needs the opt-in; kMaxBoxes * 28 B = 112 KiB
    // fits the 160 KiB of a gfx950 CU
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kMaxBoxes * 7 * (int)sizeof(float)) == hipSuccess;
    if (!attr_ok && lds > 64 * 1024) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(kThreads), lds, (hipStream_t)stream, boxes, scores, labels,
                       (int)n, iou_threshold, keep, count);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
