// msda_bwd_owner.hip -- MSDA backward for SMALL query sets (the decoder's cross-attention: a few
// thousand queries, reference points anywhere in the image), D == 32.
//
// The query-tiled backward (msda_bwd_tiled.hip) keeps grad_value windows in LDS per tile of
// QUERIES; that works when neighbouring queries hit neighbouring pixels (the encoder).  Decoder
// queries are spatially unordered: a tile's samples cover the whole image, nearly every
// contribution leaves the window and becomes a global float atomic, and trained models pile
// many queries on the same object -- atomics on the same rows serialise (183 us per call in the
// training step for ~2 k queries).
//
// This kernel turns the ownership around: a workgroup owns a RANGE OF VALUE ROWS (<= 416
// consecutive pixels of one level, one head, one image) in LDS and scans ALL samples of that
// (image, head, level) -- a few thousand, 12 bytes each -- keeping those that touch its range:
//   * grad_value: every corner that falls in the range is accumulated into the LDS window in
//     32-bit fixed point (two channels per ds_add_u64, order-independent), and at the end the
//     whole window is written with plain coalesced stores: no global atomics, no contention,
//     and no zero-fill pass over grad_value (every row is written by exactly one workgroup);
//   * grad_loc / grad_attn of a sample are computed by the workgroup that owns the sample's
//     (clamped) top-left pixel, gathering the four corner rows from global memory.
// Scanning is cheap (geometry only: ~40 VALU operations per sample) compared with the
// ~10x more expensive contributions it filters.  Fixed-point scale per (image, head, level) =
// 2^30 / (max|grad_out| * sum|attn| over its samples), worked out by a small kernel in front (owner_bounds).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"
#include "msda_tiled.h"

namespace {

#ifndef OWNER_THREADS
#define OWNER_THREADS 768
#endif
#ifndef OWNER_ROWS
#define OWNER_ROWS 416
#endif
constexpr int kThreads = OWNER_THREADS; // 2 workgroups per CU = 6 waves per SIMD (76 KB of LDS each)
constexpr int kWaves = kThreads / 64;
constexpr int kLPR = 8;                 // lanes per 32-channel row (float4 each)
constexpr int kGroups = kThreads / kLPR;
constexpr int kRows = OWNER_ROWS;            // value rows owned by a workgroup (52 KB of accumulators)
constexpr unsigned kOutOfRange = 0x80000000u;

struct Entry {               // 32 B: a sample that concerns this workgroup
    int yx;                  // (y0 << 16) | (x0 & 0xffff)
    unsigned flags;          // bits 0..3 corner valid AND in range, bit 4 owner, bits 8..11 valid corners, 16.. p
    float lh, lw, a, aW, aH;
    int q;
};

__device__ __forceinline__ float4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128");
    return __builtin_bit_cast(float4, r);
}

__device__ __forceinline__ float row_sum8(float v) {
    v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    return v;
}

struct OwnerMeta {
    int L;
    int H[DATR_TILED_MAX_LEVELS], W[DATR_TILED_MAX_LEVELS], start[DATR_TILED_MAX_LEVELS];
    int range_base[DATR_TILED_MAX_LEVELS + 1];    // first range of each level; [L] = total
};

// Bound of the fixed-point scale, once per (image, head, level) instead of once per workgroup (54 ranges
// of a head re-read all of its grad_out rows: 24 of the kernel's 149 us):
//   max |grad_out[n, :, m, :]| * sum |attn[n, :, m, l, :]|
// as kBoundParts partial results per (image, head) -- the queries in kBoundParts slices, one workgroup each:
// parts[((n * M + m) * kBoundParts + c) * (1 + L) + {0: max, 1 + l: sum}] -- which every consumer combines
// in the same fixed order (the scale, and with it grad_value, stays bitwise reproducible).
constexpr int kBoundParts = 16;
__global__ __launch_bounds__(256) void owner_bounds(const float *__restrict__ grad_out, const float *__restrict__ attn,
                                                    int Lq, int M, int L, int P, float *__restrict__ bounds)
{
    constexpr int D = 32, kT = 256;
    __shared__ float red[(1 + DATR_TILED_MAX_LEVELS) * (kT / 64)];
    const int part = blockIdx.x % kBoundParts, nm = blockIdx.x / kBoundParts;
    const int n = nm / M, m = nm % M, tid = threadIdx.x, K = L * P;
    const int q_lo = (int)((long)Lq * part / kBoundParts), q_hi = (int)((long)Lq * (part + 1) / kBoundParts);
    float mx = 0.f, asum[DATR_TILED_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < DATR_TILED_MAX_LEVELS; ++l) asum[l] = 0.f;
    for (int i = q_lo * (D / 4) + tid; i < q_hi * (D / 4); i += kT) {
        const int q = i >> 3, j = i & 7;
        const float4 v = reinterpret_cast<const float4 *>(grad_out + (((size_t)n * Lq + q) * M + m) * D)[j];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int i = q_lo * K + tid; i < q_hi * K; i += kT) {
        const int q = i / K, k = i - q * K, l = k / P;
        const float a = fabsf(attn[(((size_t)n * Lq + q) * M + m) * K + k]);
#pragma unroll
        for (int ll = 0; ll < DATR_TILED_MAX_LEVELS; ++ll) asum[ll] += ll == l ? a : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
#pragma unroll
        for (int l = 0; l < DATR_TILED_MAX_LEVELS; ++l) asum[l] += __shfl_xor(asum[l], o, 64);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = mx;
#pragma unroll
        for (int l = 0; l < DATR_TILED_MAX_LEVELS; ++l) red[(1 + l) * (kT / 64) + (tid >> 6)] = asum[l];
    }
    __syncthreads();
    if (tid <= L) {                                   // 0: the maximum, 1 + l: level l's sum
        float r = 0.f;
        for (int w = 0; w < kT / 64; ++w) {
            const float v = red[tid * (kT / 64) + w];
            r = tid == 0 ? fmaxf(r, v) : r + v;
        }
        bounds[(size_t)blockIdx.x * (1 + L) + tid] = r;
    }
}

__global__ __launch_bounds__(kThreads) void msda_bwd_owner_d32(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const float *__restrict__ loc, const float *__restrict__ attn, const OwnerMeta meta, int S,
    int M, int P, int Lq, float *__restrict__ grad_value, long gv_row_stride, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, const float *__restrict__ bounds)
{
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Entry *queue = reinterpret_cast<Entry *>(smem);                                // 24 KB
    int *win = reinterpret_cast<int *>(smem + kThreads * sizeof(Entry));           // 52 KB
    int *ctl = reinterpret_cast<int *>(win + kRows * D);                           // [0] queue length

    const int R = meta.range_base[meta.L];
    const int bid = blockIdx.x;
    const int m = bid % M;
    const int r = (bid / M) % R;
    const int n = bid / (M * R);
    int l = 0;
    while (l + 1 < meta.L && r >= meta.range_base[l + 1]) ++l;
    const int H = meta.H[l], W = meta.W[l], start = meta.start[l];
    const int p0 = (r - meta.range_base[l]) * kRows;              // first owned pixel (level-local)
    const int nrows = min(kRows, H * W - p0);
    const int K = meta.L * P;

    const int tid = threadIdx.x, g = tid / kLPR, j = tid % kLPR;
    const unsigned row_bytes = (unsigned)(M * D) * 4u;
    const size_t item = ((size_t)n * S * M + m) * D;
    __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(value + item), 0, (S * M - m) * D * 4, 0x00020000);
    const unsigned chan = (unsigned)j * 16u;
    const int nsamples = Lq * P;

    // ---- the bound of the fixed-point scale (owner_bounds); zero the accumulators ------------------
    for (int i = tid; i < kRows * (D / 4); i += kThreads)
        reinterpret_cast<int4 *>(win)[i] = make_int4(0, 0, 0, 0);
    if (tid == 0) ctl[0] = 0;
    __syncthreads();
    float maxgo = 0.f, asum_all = 0.f;
    {
        const float *pb = bounds + ((size_t)n * M + m) * kBoundParts * (1 + meta.L);
        for (int c = 0; c < kBoundParts; ++c) {
            maxgo = fmaxf(maxgo, pb[c * (1 + meta.L)]);
            asum_all += pb[c * (1 + meta.L) + 1 + l];
        }
    }
    const float bound = maxgo * asum_all;
    const float scale = bound > 0.f ? 1073741824.f / bound : 0.f;      // 2^30 / bound
    const float inv_scale = bound * (1.f / 1073741824.f);

    // ---- main pass: chunks of kThreads samples: filter, then let the row groups work the queue ----
    for (int base = 0; base < nsamples; base += kThreads) {
        const int s = base + tid;
        if (s < nsamples) {
            const int q = s / P, p = s - q * P;
            const size_t k = (((size_t)n * Lq + q) * M + m) * K + l * P + p;
            const float2 xy = reinterpret_cast<const float2 *>(loc)[k];
            const float a = attn[k];
            const float Hf = (float)H, Wf = (float)W;
            const float h_im = xy.y * Hf - 0.5f, w_im = xy.x * Wf - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int y0 = inside ? (int)hf : 0, x0 = inside ? (int)wf : 0;
            const bool top = inside && y0 >= 0, bot = inside && y0 + 1 <= H - 1;
            const bool lef = x0 >= 0, rig = x0 + 1 <= W - 1;
            const unsigned valid = (top && lef ? 1u : 0u) | (top && rig ? 2u : 0u) |
                                   (bot && lef ? 4u : 0u) | (bot && rig ? 8u : 0u);
            const int pix = y0 * W + x0 - p0;                       // corner (0,0), range-local
            unsigned mine = 0;
            if ((valid & 1u) && (unsigned)pix < (unsigned)nrows) mine |= 1u;
            if ((valid & 2u) && (unsigned)(pix + 1) < (unsigned)nrows) mine |= 2u;
            if ((valid & 4u) && (unsigned)(pix + W) < (unsigned)nrows) mine |= 4u;
            if ((valid & 8u) && (unsigned)(pix + W + 1) < (unsigned)nrows) mine |= 8u;
            // owner of the sample's grad_loc / grad_attn: the range holding its clamped corner
            const int oy = min(max(y0, 0), H - 1), ox = min(max(x0, 0), W - 1);
            const bool owner = (unsigned)(oy * W + ox - p0) < (unsigned)nrows;
            if (mine | (owner ? 16u : 0u)) {
                Entry e;
                e.yx = (y0 << 16) | (x0 & 0xffff);
                e.flags = mine | (owner ? 16u : 0u) | (valid << 8) | ((unsigned)p << 16);
                e.lh = inside ? h_im - hf : 0.f;
                e.lw = inside ? w_im - wf : 0.f;
                e.a = a;
                e.aW = a * Wf;
                e.aH = a * Hf;
                e.q = q;
                queue[atomicAdd(&ctl[0], 1)] = e;
            }
        }
        __syncthreads();
        const int nent = ctl[0];
        for (int ei = g; ei < nent; ei += kGroups) {
            const int4 s0 = reinterpret_cast<const int4 *>(queue + ei)[0];
            const float4 s1 = reinterpret_cast<const float4 *>(queue + ei)[1];
            const int y0 = s0.x >> 16, x0 = (int)(short)(s0.x & 0xffff);
            const unsigned flags = (unsigned)s0.y;
            const float lh = __builtin_bit_cast(float, s0.z), lw = __builtin_bit_cast(float, s0.w);
            const float a = s1.x, aW = s1.y, aH = s1.z;
            const int q = __builtin_bit_cast(int, s1.w);
            const size_t qm = ((size_t)n * Lq + q) * M + m;
            const float4 go = reinterpret_cast<const float4 *>(grad_out + qm * D)[j];
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float c0 = hh * hw, c1 = hh * lw, c2 = lh * hw, c3 = lh * lw;
            if (flags & 16u) {
                const unsigned valid = (flags >> 8) & 0xfu;
                const int p = (int)((flags >> 16) & 0xfu);
                const int gp = start + y0 * W + x0;
                const float4 v0 = load_row4(vsrc, ((valid & 1u) ? (unsigned)gp * row_bytes : kOutOfRange) + chan);
                const float4 v1 = load_row4(vsrc, ((valid & 2u) ? (unsigned)(gp + 1) * row_bytes : kOutOfRange) + chan);
                const float4 v2 = load_row4(vsrc, ((valid & 4u) ? (unsigned)(gp + W) * row_bytes : kOutOfRange) + chan);
                const float4 v3 = load_row4(vsrc, ((valid & 8u) ? (unsigned)(gp + W + 1) * row_bytes : kOutOfRange) + chan);
                const float d0 = go.x * v0.x + go.y * v0.y + go.z * v0.z + go.w * v0.w;
                const float d1 = go.x * v1.x + go.y * v1.y + go.z * v1.z + go.w * v1.w;
                const float d2 = go.x * v2.x + go.y * v2.y + go.z * v2.z + go.w * v2.w;
                const float d3 = go.x * v3.x + go.y * v3.y + go.z * v3.z + go.w * v3.w;
                const float pa = row_sum8(c0 * d0 + c1 * d1 + c2 * d2 + c3 * d3);
                const float pw = row_sum8(hh * (d1 - d0) + lh * (d3 - d2)) * aW;
                const float ph = row_sum8(hw * (d2 - d0) + lw * (d3 - d1)) * aH;
                const size_t kk = qm * K + l * P + p;
                if (j == 0) reinterpret_cast<float2 *>(grad_loc)[kk] = make_float2(pw, ph);
                if (j == 1) grad_attn[kk] = pa;
            }
            const float4 gs = make_float4(go.x * a * scale, go.y * a * scale, go.z * a * scale,
                                          go.w * a * scale);
            const int pix = y0 * W + x0 - p0;
#define DATR_CORNER(BIT, OFF, C)                                                                  \
            if (flags & (BIT)) {                                                                  \
                unsigned long long *dst =                                                         \
                    reinterpret_cast<unsigned long long *>(win + (pix + (OFF)) * D + j * 4);      \
                const long long p01 = (long long)__float2int_rn((C) * gs.x) +                     \
                    ((long long)__float2int_rn((C) * gs.y) << 32);                                \
                const long long p23 = (long long)__float2int_rn((C) * gs.z) +                     \
                    ((long long)__float2int_rn((C) * gs.w) << 32);                                \
                atomicAdd(dst + 0, (unsigned long long)p01);                                      \
                atomicAdd(dst + 1, (unsigned long long)p23);                                      \
            }
            DATR_CORNER(1u, 0, c0)
            DATR_CORNER(2u, 1, c1)
            DATR_CORNER(4u, W, c2)
            DATR_CORNER(8u, W + 1, c3)
#undef DATR_CORNER
        }
        __syncthreads();
        if (tid == 0) ctl[0] = 0;
        // the next chunk's pushes come after its own filtering, behind the barrier below
        __syncthreads();
    }

    // ---- every owned row is written once, plain coalesced stores (no zero-fill needed) ----------
    {
        const int lane32 = tid & 31, rsub = tid >> 5;
        // grad_value rows may sit `gv_row_stride` floats apart (a column slice of a wider buffer that
        // several calls share: datr_msda_backward_strided_f32); M * D for the plain layout
        float *gbase = grad_value + (size_t)n * S * gv_row_stride + (size_t)m * D;
        for (int rr = rsub; rr < nrows; rr += kThreads / 32) {
            const long long pk = reinterpret_cast<const long long *>(win + rr * D)[lane32 >> 1];
            const int lo = (int)(unsigned)(pk & 0xffffffffLL);
            const int hi = (int)((pk - (long long)lo) >> 32);
            const float v = (float)((lane32 & 1) ? hi : lo) * inv_scale;
            gbase[(size_t)(start + p0 + rr) * gv_row_stride + lane32] = v;
        }
    }
}

}  // namespace

DATR_INTERNAL int datr_internal_msda_bwd_owner_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const DatrTiledMeta *tm, int64_t N, int64_t S, int64_t M, int64_t P, int64_t Lq,
    float *grad_value, int64_t grad_value_row_stride, float *grad_loc, float *grad_attn, void *stream)
{
    OwnerMeta meta;
    meta.L = tm->L;
    int base = 0;
    for (int l = 0; l < tm->L; ++l) {
        meta.H[l] = tm->lv[l].H; meta.W[l] = tm->lv[l].W; meta.start[l] = tm->lv[l].start;
        meta.range_base[l] = base;
        base += (tm->lv[l].H * tm->lv[l].W + kRows - 1) / kRows;
    }
    meta.range_base[tm->L] = base;
    const int64_t blocks = N * M * base;
    if (blocks <= 0 || blocks > 0x7fffffff || Lq * P > 0x3fffffff) return DATR_EUNSUPPORTED;
    if (tm->L > DATR_TILED_MAX_LEVELS || Lq * tm->L * P > 0x3fffffff) return DATR_EUNSUPPORTED;
    const size_t lds = kThreads * sizeof(Entry) + (size_t)kRows * 32 * 4 + 16 + 2 * kWaves * 4;
    // N * M * kBoundParts * (1 + L) floats of stream-ordered scratch for the bounds: allocated and released on `stream`
    hipStream_t st = (hipStream_t)stream;
    float *bounds = nullptr;
    if (hipMallocAsync(reinterpret_cast<void **>(&bounds), (size_t)(N * M * kBoundParts * (1 + tm->L)) * sizeof(float), st) != hipSuccess)
        return DATR_ELAUNCH;
    hipLaunchKernelGGL(owner_bounds, dim3((unsigned)(N * M * kBoundParts)), dim3(256), 0, st, grad_out, attn, (int)Lq, (int)M,
                       tm->L, (int)P, bounds);
    hipLaunchKernelGGL(msda_bwd_owner_d32, dim3((unsigned)blocks), dim3(kThreads), lds, st, grad_out, value, loc, attn,
                       meta, (int)S, (int)M, (int)P, (int)Lq, grad_value, (long)grad_value_row_stride, grad_loc,
                       grad_attn, bounds);
    const bool launched = hipGetLastError() == hipSuccess;
    const bool freed = hipFreeAsync(bounds, st) == hipSuccess;
    return launched && freed ? DATR_OK : DATR_ELAUNCH;
}
