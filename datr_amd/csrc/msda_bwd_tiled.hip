// msda_bwd_tiled.hip -- query-tiled MSDA backward.  Built for "self-attention over the pyramid"
// calls (Lq == S: the queries ARE the pixels of the multi-level feature map, level-major raster
// order; workgroup tile = 16 x 8 pixels), the encoder calls that make up half of the backward
// launches of a DATR step and >97 % of the backward's work; also used in LINEAR mode (tile =
// 128 consecutive queries) for the decoder calls, where the queries have no spatial order but
// the coarse levels (273 / 1050 rows) still fit the window and are exactly where hundreds of
// de-noising copies of the same ground-truth box collide on the same rows (reference: /root/reference/models/dino/deformable_transformer.py:807-809
// -> ms_deform_im2col_cuda.cuh:301-403).
//
// Why a second kernel.  grad_value is a scatter-add of 728 M floats per encoder call.  On
// MI355X global float atomics retire at <= 327 G/s when perfectly coalesced and collapse
// under contention (the 273-pixel level receives a quarter of all updates): the row kernel in
// msda.hip measures 4.9 ms per call on model-like locations (profiles/r01_probes.md).  LDS
// integer adds run >10x faster (ds_add_u32 3.7 T lane-ops/s vs ds_add_f32 0.2 T), so:
//
//   * a workgroup owns a 16x8 TILE of queries of one level and one head.  Neighbouring queries
//     sample neighbouring pixels, so per target level their corner rows fall in a small window;
//   * per target level the workgroup (A) turns its (query, point) pairs into corner geometry in
//     LDS and finds the bounding box of the corners, (B) accumulates every d(out)/d(value)
//     contribution that falls inside a window of up to kWinRows rows into LDS as 32-bit FIXED
//     POINT with ds_add_u32 -- exact integer adds, so the in-window sum is independent of the
//     order of arrival -- and (C) flushes each touched row once with 128-byte-coalesced float
//     atomics.  Contributions outside the window (far-away offsets) go straight to global
//     atomics, so the result never depends on the window heuristics;
//   * the fixed-point scale is chosen per (workgroup, level) from max|grad_out| over the tile
//     and sum|attn| over the level's pairs, which bounds every in-window sum below 2^30; the
//     quantisation step is ~3e-8 of max|grad_out| (fp32 eps is 6e-8);
//   * grad_loc / grad_attn are computed exactly as in the row kernel (DPP reductions over the
//     8 lanes that share a 32-channel row) from corner rows gathered with raw buffer loads.
//
// Results are identical in meaning to the row kernel; the dispatcher in msda.hip picks this one
// when the caller can also hand over HOST copies of the level geometry (needed for the grid).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "datr_hip.h"
#include "msda_tiled.h"

#if defined(DATR_PROBE) && !defined(DATR_PROBE_NOTICKS)
__device__ unsigned long long datr_phase_cycles[8];
#define DATR_TICK(i)                                                                   \
    do {                                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();                  \
        if (threadIdx.x == 0) atomicAdd(&datr_phase_cycles[i], now_ - tick_);          \
        tick_ = now_;                                                                  \
    } while (0)
#define DATR_TICK_INIT unsigned long long tick_ = __builtin_readcyclecounter()
extern "C" void datr_probe_phase_cycles(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(datr_phase_cycles), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(datr_phase_cycles), z, sizeof(z));
    }
}
#else
#define DATR_TICK(i) do {} while (0)
#define DATR_TICK_INIT do {} while (0)
#endif

namespace {

constexpr int kThreads = 768;            // 12 waves; 2 workgroups per CU (LDS) = 6 waves per SIMD
constexpr int kWaves = kThreads / 64;
static_assert(kWaves <= 16, "fctl holds 16 per-wave partials per quantity");
constexpr int kLPR = 8;                 // lanes per 32-channel row (float4 each)
constexpr int kGroups = kThreads / kLPR;
constexpr int kWinRows = 480;           // rows (x 32 ch x 4 B = 60 KB) of the accumulation window
constexpr int kMaxPairs = 512;          // kTQ * P with P <= 4
constexpr unsigned kOutOfRange = 0x80000000u;

struct PairGeom {            // 32 B per (query, point) pair of the current level
    int yx;                  // (y0 << 16) | (x0 & 0xffff), corner (0,0) pixel, signed 16 bit each
    unsigned okq;            // bits 0..3 corner validity, bits 4..7 point index p
    float lh, lw, a, aW, aH;
    int q;                   // global query index
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

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ABL (probe builds only, -DDATR_PROBE): 1 = no LDS adds, 2 = no flush, 4 = no corner gathers,
// 8 = no grad_loc/grad_attn stores.  0 in product builds.
template <int ABL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void msda_bwd_tiled_d32(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const float *__restrict__ loc, const float *__restrict__ attn, const DatrTiledMeta meta,
    int S, int M, int P, int split_levels, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PairGeom *stage = reinterpret_cast<PairGeom *>(smem);                          // 16 KB
    int *win = reinterpret_cast<int *>(smem + kMaxPairs * sizeof(PairGeom));       // 60 KB
    unsigned char *touched = reinterpret_cast<unsigned char *>(win + kWinRows * D);
    int *ctl = reinterpret_cast<int *>(touched + ((kWinRows + 15) & ~15));         // 16 ints
    float *fctl = reinterpret_cast<float *>(ctl + 16);                             // 32 floats

    const int L = meta.L, K = L * P, Lq = meta.Lq;
    // small launches (the decoder calls: a few thousand queries) give every target level its own
    // workgroup -- 4x the parallelism for a grid that would otherwise not fill the chip
    int bid = blockIdx.x, l_begin = 0, l_end = L;
    if (split_levels) {
        l_begin = bid % L;
        l_end = l_begin + 1;
        bid /= L;
    }
    const int m = bid % M;
    const int tile = (bid / M) % meta.total_tiles;
    const int n = bid / (M * meta.total_tiles);
    int lq = 0;
    while (lq + 1 < meta.QL && tile >= meta.qlv[lq + 1].tile_base) ++lq;
    const int tl = tile - meta.qlv[lq].tile_base;
    const int tx = tl % meta.qlv[lq].tiles_x, ty = tl / meta.qlv[lq].tiles_x;
    const int qx0 = tx * meta.tile_w, qy0 = ty * meta.tile_h;
    const int qW = meta.qlv[lq].W;
    const int tw = min(meta.tile_w, qW - qx0), th = min(meta.tile_h, meta.qlv[lq].H - qy0);
    const int nq = tw * th;
    const int q_base = meta.qlv[lq].start + qy0 * qW + qx0;

    const int tid = threadIdx.x, g = tid / kLPR, j = tid % kLPR;
    const int wave = tid >> 6;
    const unsigned row_bytes = (unsigned)(M * D) * 4u;
    const size_t item = ((size_t)n * S * M + m) * D;
    const int records = (S * M - m) * D * 4;
    __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(value + item), 0, records, 0x00020000);
    __amdgpu_buffer_rsrc_t gsrc =
        __builtin_amdgcn_make_buffer_rsrc(grad_value + item, 0, records, 0x00020000);
    const unsigned chan = (unsigned)j * 16u;

    DATR_TICK_INIT;
    // ---- once per workgroup: max |grad_out| over the tile (this head's 32 channels) -----------
    {
        float mx = 0.f;
        for (int qi = g; qi < nq; qi += kGroups) {
            const int q = q_base + (qi / tw) * qW + (qi % tw);
            const float4 v = reinterpret_cast<const float4 *>(
                grad_out + (((size_t)n * Lq + q) * M + m) * D)[j];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        mx = wave_max(mx);
        if ((tid & 63) == 0) fctl[wave] = mx;
        __syncthreads();
    }
    float maxgo = 0.f;
    for (int w = 0; w < kWaves; ++w) maxgo = fmaxf(maxgo, fctl[w]);
    __syncthreads();

    DATR_TICK(0);
    const int npairs = nq * P;
    for (int l = l_begin; l < l_end; ++l) {
        const int H = meta.lv[l].H, W = meta.lv[l].W, start = meta.lv[l].start;
        // ---- phase A: pair geometry, corner bounding box, sum |attn| ---------------------------
        if (tid < 4) ctl[tid] = (tid & 1) ? -(1 << 30) : (1 << 30);   // [minx, maxx, miny, maxy]
        __syncthreads();
        float asum = 0.f;
        int mnx = 1 << 30, mxx = -(1 << 30), mny = 1 << 30, mxy = -(1 << 30);
        for (int pi = tid; pi < npairs; pi += kThreads) {
            const int qi = pi / P, p = pi - qi * P;
            const int q = q_base + (qi / tw) * qW + (qi % tw);
            const size_t k = (((size_t)n * Lq + q) * M + m) * K + l * P + p;
            const float2 xy = reinterpret_cast<const float2 *>(loc)[k];
            const float a = attn[k];
            const float Hf = (float)H, Wf = (float)W;
            const float h_im = xy.y * Hf - 0.5f, w_im = xy.x * Wf - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int y0 = inside ? (int)hf : 0, x0 = inside ? (int)wf : 0;
            unsigned ok = 0;
            if (inside) {
                const bool top = y0 >= 0, bot = y0 + 1 <= H - 1, lef = x0 >= 0, rig = x0 + 1 <= W - 1;
                ok = (top && lef ? 1u : 0u) | (top && rig ? 2u : 0u) | (bot && lef ? 4u : 0u) |
                     (bot && rig ? 8u : 0u);
                const int cx0 = lef ? x0 : x0 + 1, cx1 = rig ? x0 + 1 : x0;
                const int cy0 = top ? y0 : y0 + 1, cy1 = bot ? y0 + 1 : y0;
                if (ok) {
                    mnx = min(mnx, cx0); mxx = max(mxx, cx1);
                    mny = min(mny, cy0); mxy = max(mxy, cy1);
                }
            }
            PairGeom pg;
            pg.yx = (y0 << 16) | (x0 & 0xffff);
            pg.okq = ok | ((unsigned)p << 4);
            pg.lh = inside ? h_im - hf : 0.f;
            pg.lw = inside ? w_im - wf : 0.f;
            pg.a = a;
            pg.aW = a * Wf;
            pg.aH = a * Hf;
            pg.q = q;
            stage[pi] = pg;
            asum += fabsf(a);
        }
        asum = wave_sum(asum);
        // bounding box: reduce inside the wave first -- 512 same-address LDS atomics serialise
        for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64));
            mny = min(mny, __shfl_xor(mny, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
        }
        if ((tid & 63) == 0) {
            fctl[16 + wave] = asum;
            atomicMin(&ctl[0], mnx); atomicMax(&ctl[1], mxx);
            atomicMin(&ctl[2], mny); atomicMax(&ctl[3], mxy);
        }
        __syncthreads();
        DATR_TICK(1);
        // ---- window: the corner bounding box, centred and clipped to kWinRows rows -------------
        const int bx0 = ctl[0], bx1 = ctl[1], by0 = ctl[2], by1 = ctl[3];
        const bool any = bx1 >= bx0 && by1 >= by0;
        int ww = 0, wh = 0, wx0 = 0, wy0 = 0;
        if (any) {
            const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            ww = min(bw, 48);
            wh = min(bh, kWinRows / ww);
            wx0 = bx0 + (bw - ww) / 2;
            wy0 = by0 + (bh - wh) / 2;
        }
        const int wrows = ww * wh;
        float asum_all = 0.f;
        for (int w = 0; w < kWaves; ++w) asum_all += fctl[16 + w];
        const float bound = maxgo * asum_all;
        const float scale = bound > 0.f ? 1073741824.f / bound : 0.f;      // 2^30 / bound
        const float inv_scale = bound * (1.f / 1073741824.f);
        for (int i = tid; i < wrows * (D / 4); i += kThreads)
            reinterpret_cast<int4 *>(win)[i] = make_int4(0, 0, 0, 0);
        for (int i = tid; i < wrows; i += kThreads) touched[i] = 0;
        __syncthreads();
        DATR_TICK(2);

        // ---- phase B: gather corners, reduce grad_loc/grad_attn, accumulate grad_value ---------
        // (kUnroll = 1 since the kernel runs at 6 waves per SIMD -- 80 VGPRs, amdgpu_waves_per_eu --
        // where occupancy hides the latency better than a second pair in flight did at 4 waves:
        // 564 -> 544 us per N=2 encoder call; the LDS atomic path is issue-bound per wave,
        // profiles/r01_probes.md.)
        // kUnroll pairs per group are in flight at once: all their loads (stage, grad_out row,
        // four corner rows) are issued before the first one is consumed, otherwise the chain
        // LDS read -> global gather -> DPP -> LDS add is pure latency at 16 waves per CU.
        struct Inflight {
            int4 s0; float4 s1; float4 go, v0, v1, v2, v3; unsigned o0, o1, o2, o3;
        };
        auto issue = [&](int pi, Inflight &f) {
            f.s0 = reinterpret_cast<const int4 *>(stage + pi)[0];
            f.s1 = reinterpret_cast<const float4 *>(stage + pi)[1];
            const int y0 = f.s0.x >> 16, x0 = (int)(short)(f.s0.x & 0xffff);
            const unsigned ok = (unsigned)f.s0.y & 0xfu;
            const int q = __builtin_bit_cast(int, f.s1.w);
            f.go = reinterpret_cast<const float4 *>(
                grad_out + (((size_t)n * Lq + q) * M + m) * D)[j];
            const int pix = start + y0 * W + x0;
            f.o0 = (ok & 1u) ? (unsigned)pix * row_bytes : kOutOfRange;
            f.o1 = (ok & 2u) ? (unsigned)(pix + 1) * row_bytes : kOutOfRange;
            f.o2 = (ok & 4u) ? (unsigned)(pix + W) * row_bytes : kOutOfRange;
            f.o3 = (ok & 8u) ? (unsigned)(pix + W + 1) * row_bytes : kOutOfRange;
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            f.v0 = (ABL & 4) ? zero4 : load_row4(vsrc, f.o0 + chan);
            f.v1 = (ABL & 4) ? zero4 : load_row4(vsrc, f.o1 + chan);
            f.v2 = (ABL & 4) ? zero4 : load_row4(vsrc, f.o2 + chan);
            f.v3 = (ABL & 4) ? zero4 : load_row4(vsrc, f.o3 + chan);
        };
        auto consume = [&](const Inflight &f) {
            const int y0 = f.s0.x >> 16, x0 = (int)(short)(f.s0.x & 0xffff);
            const unsigned ok = (unsigned)f.s0.y & 0xfu;
            const int p = (int)(((unsigned)f.s0.y >> 4) & 0xfu);
            const float lh = __builtin_bit_cast(float, f.s0.z), lw = __builtin_bit_cast(float, f.s0.w);
            const float a = f.s1.x, aW = f.s1.y, aH = f.s1.z;
            const int q = __builtin_bit_cast(int, f.s1.w);
            const size_t qm = ((size_t)n * Lq + q) * M + m;
            const float4 go = f.go, v0 = f.v0, v1 = f.v1, v2 = f.v2, v3 = f.v3;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float c0 = hh * hw, c1 = hh * lw, c2 = lh * hw, c3 = lh * lw;
            // per-corner partial dot products <grad_out, v_c> over this lane's 4 channels; the
            // bilinear combinations below are linear in them (16 + 12 flops instead of 60)
            const float d0 = go.x * v0.x + go.y * v0.y + go.z * v0.z + go.w * v0.w;
            const float d1 = go.x * v1.x + go.y * v1.y + go.z * v1.z + go.w * v1.w;
            const float d2 = go.x * v2.x + go.y * v2.y + go.z * v2.z + go.w * v2.w;
            const float d3 = go.x * v3.x + go.y * v3.y + go.z * v3.z + go.w * v3.w;
            float pa = c0 * d0 + c1 * d1 + c2 * d2 + c3 * d3;
            float pw = hh * (d1 - d0) + lh * (d3 - d2);
            float ph = hw * (d2 - d0) + lw * (d3 - d1);
            pa = row_sum8(pa);
            pw = row_sum8(pw) * aW;
            ph = row_sum8(ph) * aH;
            const size_t kk = qm * K + l * P + p;
            if (!(ABL & 8)) {
                if (j == 0) reinterpret_cast<float2 *>(grad_loc)[kk] = make_float2(pw, ph);
                if (j == 1) grad_attn[kk] = pa;
            } else {
                asm volatile("" ::"v"(pw), "v"(ph), "v"(pa));
            }
            // d out / d value: a * corner weight * grad_out, into the window (fixed point) or,
            // for a corner that fell outside it, straight to global memory
            const float4 ga = make_float4(go.x * a, go.y * a, go.z * a, go.w * a);
            // pre-scaled for the fixed-point path: one multiply + one convert per channel and corner
            const float4 gs = make_float4(ga.x * scale, ga.y * scale, ga.z * scale, ga.w * scale);
            const int ry = y0 - wy0, rx = x0 - wx0;
#define DATR_CORNER(BIT, DY, DX, C, OFF)                                                        \
            if (ok & (BIT)) {                                                                   \
                const int cy = ry + (DY), cx = rx + (DX);                                       \
                if ((ABL & 1)) {                                                                \
                    asm volatile("" ::"v"((C) * ga.x), "v"((C) * ga.w));                        \
                } else if ((unsigned)cy < (unsigned)wh && (unsigned)cx < (unsigned)ww) {        \
                    const int r = cy * ww + cx;                                                 \
                    /* two 32-bit fixed-point channels per 64-bit LDS add: hi << 32 + lo as a   \
                       signed sum; a negative lo borrows from hi and phase C undoes it */       \
                    unsigned long long *dst =                                                   \
                        reinterpret_cast<unsigned long long *>(win + r * D + j * 4);            \
                    const long long p01 = (long long)__float2int_rn((C) * gs.x) +               \
                        ((long long)__float2int_rn((C) * gs.y) << 32);                          \
                    const long long p23 = (long long)__float2int_rn((C) * gs.z) +               \
                        ((long long)__float2int_rn((C) * gs.w) << 32);                          \
                    atomicAdd(dst + 0, (unsigned long long)p01);                                \
                    atomicAdd(dst + 1, (unsigned long long)p23);                                \
                    if (j == 0) touched[r] = 1;                                                 \
                } else {                                                                        \
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((C) * ga.x, gsrc, (OFF) + chan, 0, 0);      \
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((C) * ga.y, gsrc, (OFF) + chan + 4, 0, 0);  \
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((C) * ga.z, gsrc, (OFF) + chan + 8, 0, 0);  \
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((C) * ga.w, gsrc, (OFF) + chan + 12, 0, 0); \
                }                                                                               \
            }
            DATR_CORNER(1u, 0, 0, c0, f.o0)
            DATR_CORNER(2u, 0, 1, c1, f.o1)
            DATR_CORNER(4u, 1, 0, c2, f.o2)
            DATR_CORNER(8u, 1, 1, c3, f.o3)
#undef DATR_CORNER
        };
        constexpr int kUnroll = 1;
        for (int base = g; base < npairs; base += kUnroll * kGroups) {
            Inflight f[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (base + u * kGroups < npairs) issue(base + u * kGroups, f[u]);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (base + u * kGroups < npairs) consume(f[u]);
        }
        DATR_TICK(3);
        __syncthreads();
        DATR_TICK(4);

        // ---- phase C: one coalesced float atomic per touched (row, channel) --------------------
        {
            const int lane32 = tid & 31, rsub = tid >> 5;          // kThreads/32 rows per pass
            for (int r = rsub; r < wrows && !(ABL & 2); r += kThreads / 32) {
                if (!touched[r]) continue;
                const int cy = r / ww, cx = r - cy * ww;
                const int pixel = start + (wy0 + cy) * W + (wx0 + cx);
                const long long pk = reinterpret_cast<const long long *>(win + r * D)[lane32 >> 1];
                const int lo = (int)(unsigned)(pk & 0xffffffffLL);
                const int hi = (int)((pk - (long long)lo) >> 32);
                const float v = (float)((lane32 & 1) ? hi : lo) * inv_scale;
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                    v, gsrc, (unsigned)pixel * row_bytes + (unsigned)lane32 * 4u, 0, 0);
            }
        }
        __syncthreads();
        DATR_TICK(5);
    }
}

}  // namespace

DATR_INTERNAL int datr_internal_msda_bwd_tiled_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const DatrTiledMeta *meta, int64_t N, int64_t S, int64_t M, int64_t P,
    float *grad_value, float *grad_loc, float *grad_attn, void *stream)
{
    int64_t blocks = N * M * meta->total_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return DATR_EUNSUPPORTED;
    const int split_levels = blocks < 1024 && meta->L > 1;       // < 2 workgroups per CU x 2 rounds
    if (split_levels) blocks *= meta->L;
    const size_t lds = kMaxPairs * sizeof(PairGeom) + (size_t)kWinRows * 32 * 4 +
                       ((kWinRows + 15) & ~15) + 48 * 4;
#define DATR_LAUNCH_TILED(A)                                                                     \
    hipLaunchKernelGGL(msda_bwd_tiled_d32<A>, dim3((unsigned)blocks), dim3(kThreads), lds,         \
                       (hipStream_t)stream, grad_out, value, loc, attn, *meta, (int)S, (int)M,     \
                       (int)P, split_levels, grad_value, grad_loc, grad_attn)
#ifdef DATR_PROBE
    static const int abl = getenv("DATR_MSDA_ABLATE") ? atoi(getenv("DATR_MSDA_ABLATE")) : 0;
    switch (abl) {
        case 1: DATR_LAUNCH_TILED(1); break;
        case 2: DATR_LAUNCH_TILED(2); break;
        case 3: DATR_LAUNCH_TILED(3); break;
        case 4: DATR_LAUNCH_TILED(4); break;
        case 7: DATR_LAUNCH_TILED(7); break;
        case 15: DATR_LAUNCH_TILED(15); break;
        default: DATR_LAUNCH_TILED(0); break;
    }
#else
    DATR_LAUNCH_TILED(0);
#endif
#undef DATR_LAUNCH_TILED
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
