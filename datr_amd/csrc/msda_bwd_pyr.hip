// msda_bwd_pyr.hip -- MSDA backward for the encoder calls (Lq == S, D == 32, L == P == 4),
// pyramid-region decomposition (msda_pyr.h), grad_value by SORTED SCATTER instead of LDS atomics.
//
// Reference behaviour: /root/reference/models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403
// (col2im: grad_value += w * a * grad_out at the 4 corners, grad_attn = <grad_out, bilinear(value)>,
// grad_loc = a * {W, H} * <grad_out, d bilinear / d (w, h)>), host :87-159.
//
// Why.  The query-tiled kernel (msda_bwd_tiled.hip) accumulates d out / d value in LDS with
// fixed-point ds_add_u64: 728 M lane-adds per N=4 encoder call at 2.45 lane-ops/clk/CU are half
// of its 0.97 ms (profiles/r01_probes.md), and its 16 x 8 tiles flush 14.6 window rows per query
// with float atomics (2.8-3.8x the algorithmic write traffic).  Here:
//   * a workgroup (256 threads / <= 128 queries, or 512 / <= 256 where the samples reach far: BCfg
//     below) owns one (image, region, head): all queries of a region in all four levels;
//   * per level the 16 corner contributions of every query are written as 8-byte records
//     {weight = corner weight x attention, query slot} into LDS, COUNTING-SORTED by destination
//     row of the level's window: one 32-bit LDS atomic per record for the histogram and one for
//     the cursor (2 per corner instead of 32 channel adds);
//   * a row's gradient is then a GATHER: 8 lanes walk the row's contiguous records and
//     accumulate weight x grad_out[slot] in registers from the workgroup's LDS copy of its
//     queries' grad_out rows (ds_read_b128, no atomics), and add the finished row to grad_value
//     with one 128-byte-coalesced float atomic pass (windows of neighbouring regions overlap);
//   * grad_attn / grad_loc -- the value-dependent half -- come from msda_fwd_pyr2.hip's LDS-window
//     kernel (msda_bwd_dots_pyr2_d32) where the forward's plan covers the shape, and this kernel reads
//     no value rows (kDots = false); otherwise (kDots = true) from the corner rows gathered here with
//     zero-filling buffer loads: 4 lanes share a query, lane p works out point p's geometry, a quad
//     covers a 128-B row with two 64-B halves, dot products are completed with two DPP steps.
// A sample whose corners fall outside the level's window (offsets beyond the halo) bypasses the
// sort: its contributions go to grad_value as plain float atomics -- results never depend on
// the window heuristic.  grad_value must arrive zero-filled (the C entry point memsets it).
// The order of a row's records depends on LDS atomic arrival, so grad_value is reproducible to
// fp32 rounding, not bitwise (the reference's atomicAdd is no different).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "datr_hip.h"
#include "msda_tiled.h"
#include "msda_pyr.h"

#ifdef PYR_PROBE
// per-phase cycle counters of lane 0 of every wave (development builds only)
__device__ unsigned long long pyr_bwd_phase_cycles[1024][8];
#define TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
                     ticks_[i] += now_ - tick_; tick_ = now_; } while (0)
extern "C" void datr_probe_pyr_bwd_phase_cycles(unsigned long long *out, int reset) {
    static unsigned long long all[1024][8];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(pyr_bwd_phase_cycles), sizeof(all));
    for (int i = 0; i < 8; ++i) {
        out[i] = 0;
        for (int b = 0; b < 1024; ++b) out[i] += all[b][i];
    }
    if (reset) {
        static unsigned long long z[1024][8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pyr_bwd_phase_cycles), z, sizeof(z));
    }
}
#else
#define TICK(i) do {} while (0)
#endif

namespace {

#ifndef PYRB_GO_AUX
#define PYRB_GO_AUX 0      // cache policy of the grad_out row loads (2 = non-temporal: each row is read by one workgroup)
#endif
#ifndef PYRB_INFLIGHT
#define PYRB_INFLIGHT 8
#endif
#ifndef PYRB_WAVES_PER_SIMD
#define PYRB_WAVES_PER_SIMD 1      // minimum 256-thread workgroups per CU the register allocation must allow (development)
#endif
#ifndef PYRB_BANDS
#define PYRB_BANDS 0
#endif
// Development only (results are wrong with a bit set; tools/probes/bwd_scatter_ablate.sh):
// 1 = no float-atomic flush, 2 = no reduce loop, 4 = no records (pass B), 8 = no histogram atomics (pass A)
#ifndef PYRB_ABLATE
#define PYRB_ABLATE 0
#endif
#ifndef PYRB_REC_DPP
#define PYRB_REC_DPP (PYRB_INFLIGHT == 8)
#endif

constexpr int kMaxRows = 1024;             // rows of a level's window
constexpr unsigned kOutOfRange = 0x80000000u;
constexpr int kRowBytes = 128;
constexpr int kInFlight = PYRB_INFLIGHT;    // records of a row in flight in the reduce phase

// Launch shape: threads per workgroup and the most queries a region may hold.  256 / 128 since the value
// gathers left the kernel (tools/probes/bwd_scatter_cfgs.sh: 372 us per N = 4 encoder call at the model's
// offsets; 512 / 256: 400, 768 / 384: 603, 1024 / 512: 449 us); 512 / 256 where the sample reach makes
// the windows of 128-query regions too large (offsets ~ N(0, 2.5 px): 718 against 943 us).
template <int THREADS, int MAXQ>
struct BCfg {
    static constexpr int kThreads = THREADS, kWaves = THREADS / 64, kMaxQ = MAXQ;
    static constexpr int kMaxTasks = (kMaxQ / 16 + kWaves - 1) / kWaves;      // 16-query tasks per wave
    // LDS map (bytes)
    static constexpr int kGoOff = 0;                                  // grad_out rows by query slot
    static constexpr int kRecOff = kGoOff + kMaxQ * kRowBytes;        // records of the current level
    static constexpr int kHistOff = kRecOff + kMaxQ * 16 * 8;         // counts -> exclusive offsets [rows + 1]
    static constexpr int kCurOff = kHistOff + (2 * kThreads) * 4;     // cursors (pass B); in the reduce phase: flush scratch
    static constexpr int kFlushPerWave = 8 * kRowBytes + 8 * 4;       // 8 finished rows + their global offsets
    static constexpr int kCurBytes = (2 * kThreads) * 4 > kWaves * kFlushPerWave ? (2 * kThreads) * 4 : kWaves * kFlushPerWave;
    static constexpr int kTabOff = kCurOff + ((kCurBytes + 15) & ~15);   // query slot -> pyramid index
    static constexpr int kScanOff = kTabOff + kMaxQ * 4;              // per-wave totals of the scan
    static constexpr int kLdsBytes = kScanOff + 64;
    static_assert(kLdsBytes <= 160 * 1024, "LDS");
};
#ifdef PYRB_THREADS
typedef BCfg<PYRB_THREADS, PYRB_MAXQ> CfgSmall;                     // development: one forced shape
typedef BCfg<PYRB_THREADS, PYRB_MAXQ> CfgLarge;
#else
typedef BCfg<256, 128> CfgSmall;
typedef BCfg<512, 256> CfgLarge;
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ f4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128");
    return __builtin_bit_cast(f4, r);
}
template <int SRC>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, quad_bcast<SRC>(__builtin_bit_cast(int, v)));
}
// sum over the 4 lanes of a quad; every lane ends with the total
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float dot4(const f4 a, const f4 b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// geometry of one sample, kept in registers between the counting and the filling pass
struct Sample {
    int yx;            // (iy << 16) | (ix & 0xffff): top-left corner pixel
    float lh, lw, a;
    int flags;         // bits 0..3 corner validity (tl, tr, bl, br), bit 4: all corners in the window
    int row;           // window row index of the top-left corner (when bit 4)
};

// kDots = false: grad_loc / grad_attn come from msda_fwd_pyr2.hip's LDS-window kernel
// (msda_bwd_dots_pyr2_d32); this kernel then reads no value rows at all.
template <class C, bool kDots>
__global__ __launch_bounds__(C::kThreads, PYRB_WAVES_PER_SIMD * C::kThreads / 256 > 0 ? PYRB_WAVES_PER_SIMD * C::kThreads / 256 : 1) void msda_bwd_pyr_d32(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const float *__restrict__ loc, const float *__restrict__ attn, const PyrMeta pm, int S, int M,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr int kThreads = C::kThreads, kWaves = C::kWaves, kMaxTasks = C::kMaxTasks, kGoOff = C::kGoOff,
                  kRecOff = C::kRecOff, kHistOff = C::kHistOff, kCurOff = C::kCurOff, kFlushPerWave = C::kFlushPerWave,
                  kTabOff = C::kTabOff, kScanOff = C::kScanOff;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned *hist = reinterpret_cast<unsigned *>(lds + kHistOff);
    unsigned *cur = reinterpret_cast<unsigned *>(lds + kCurOff);
    int *qtab = reinterpret_cast<int *>(lds + kTabOff);
    unsigned *scan = reinterpret_cast<unsigned *>(lds + kScanOff);
    unsigned long long *recs = reinterpret_cast<unsigned long long *>(lds + kRecOff);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PYR_PROBE
    unsigned long long ticks_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = __builtin_readcyclecounter();
#endif
    const int bid = blockIdx.x;
#if PYRB_BANDS
    // XCD = bid % 8 owns a BAND of regions with all heads (dense lines in its L2; a grad_value line is
    // touched by one XCD's workgroups only, close together in time): bid = ((j * M + m) * 8 + band)
    const int nreg_ = pm.nRy * pm.nRx, per_band = (nreg_ + 7) / 8;
    const int n = bid / (8 * per_band * M);
    const int rem_ = bid - n * (8 * per_band * M);
    const int band = rem_ % 8, m = (rem_ / 8) % M, jj = rem_ / (8 * M);
    const int reg = band * per_band + jj;
    if (reg >= nreg_) return;
    const int ry = reg / pm.nRx, rx = reg % pm.nRx;
#else
    const int m = bid % M;
    const int reg = (bid / M) % (pm.nRy * pm.nRx);
    const int n = bid / (M * pm.nRy * pm.nRx);
    const int ry = reg / pm.nRx, rx = reg % pm.nRx;
#endif
    const unsigned row_stride = (unsigned)M * kRowBytes;
    const size_t Lq = (size_t)S;

    const size_t item = ((size_t)n * S * M + m) * 32;
    const int records = (S * M - m) * kRowBytes;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(value + item), 0, records, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc =
        __builtin_amdgcn_make_buffer_rsrc(grad_value + item, 0, records, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(grad_out + item), 0, records, 0x00020000);

    // ---- the region's queries; their grad_out rows (this head) into LDS by LDS-DMA -------------
    int pre[5];
    pre[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l)
        pre[l + 1] = pre[l] + (pm.yb[l][ry + 1] - pm.yb[l][ry]) * (pm.xb[l][rx + 1] - pm.xb[l][rx]);
    const int nq = pre[4];
    auto decode = [&](int qi) {
        const int lq = (qi >= pre[1]) + (qi >= pre[2]) + (qi >= pre[3]);
        const int li = qi - (lq == 0 ? 0 : lq == 1 ? pre[1] : lq == 2 ? pre[2] : pre[3]);
        const int oy = pm.yb[lq][ry], ox = pm.xb[lq][rx];
        const int rw = pm.xb[lq][rx + 1] - ox;
        const int r_ = (int)(((float)li + 0.5f) / (float)rw);
        return pm.start[lq] + (oy + r_) * pm.W[lq] + ox + (li - r_ * rw);
    };
    for (int qi = tid; qi < nq; qi += kThreads) qtab[qi] = decode(qi);
    for (int s0 = wave * 8; s0 < nq; s0 += kWaves * 8) {           // 8 query rows per instruction
        const int slot_ = s0 + (lane >> 3);
        const unsigned off = slot_ < nq ? (unsigned)decode(slot_) * row_stride + (unsigned)(lane & 7) * 16u
                                        : kOutOfRange;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(osrc, reinterpret_cast<lds_void *>(kGoOff + s0 * kRowBytes),
                                                 16, (int)off, 0, 0, PYRB_GO_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TICK(0);                                   // set-up: query table, grad_out rows

    // lane roles (as in msda_fwd_pyr.hip): 4 lanes per query, lane j = point j's geometry and the
    // 16-B pieces 16 j / 16 j + 64 of every row
    const int slot = lane >> 2, j = lane & 3;
    const int chan = 16 * j + 64 * (slot & 1), chan2 = chan ^ 64;
    const int ntasks = (nq + 15) >> 4;

    // sampling locations / weights of the lane's samples, requested one level ahead: their latency
    // passes behind the previous level's scan, records and reduce phases
    f2 xy_nx[kMaxTasks];
    float a_nx[kMaxTasks];
    auto request_level = [&](int l_) {
#pragma unroll
        for (int k = 0; k < kMaxTasks; ++k) {
            const int t = wave + k * kWaves;
            const int qi = t * 16 + slot;
            xy_nx[k] = f2{0.f, 0.f};
            a_nx[k] = 0.f;
            if (t < ntasks && qi < nq) {
                const size_t idx = (((size_t)n * Lq + qtab[qi]) * M + m) * 16 + l_ * 4 + j;
                xy_nx[k] = reinterpret_cast<const f2 *>(loc)[idx];
                a_nx[k] = attn[idx];
            }
        }
    };
    request_level(0);

#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const int Hl = pm.H[l], Wl = pm.W[l], stl = pm.start[l];
        // the level's window, trimmed to this head's reach (envelope plans; zero trims otherwise)
        const int WW = pm.WW[l] - pm.cut_x[m & 7][l], WH = pm.WH[l] - pm.cut_y[m & 7][l], rows = WW * WH;
        const int wy0 = pm.wy0[l][ry] + pm.cut_y0[m & 7][l], wx0 = pm.wx0[l][rx] + pm.cut_x0[m & 7][l];
        const float Hf = (float)Hl, Wf = (float)Wl;

        for (int i = tid; i <= rows; i += kThreads) hist[i] = 0;
        __syncthreads();

        // ---- pass A: geometry of point j of this lane's queries; count records per window row ---
        Sample smp[kMaxTasks];
#pragma unroll
        for (int k = 0; k < kMaxTasks; ++k) {
            const int t = wave + k * kWaves;
            Sample s;
            s.yx = 0; s.lh = 0.f; s.lw = 0.f; s.a = 0.f; s.flags = 0; s.row = 0;
            const int qi = t * 16 + slot;
            if (t < ntasks && qi < nq) {
                const f2 xy = xy_nx[k];
                s.a = a_nx[k];
                const float h_im = xy.y * Hf - 0.5f, w_im = xy.x * Wf - 0.5f;
                const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
                if (inside) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int iy = (int)hf, ix = (int)wf;
                    s.lh = h_im - hf;
                    s.lw = w_im - wf;
                    s.yx = (iy << 16) | (ix & 0xffff);
                    const bool top = iy >= 0, bot = iy + 1 <= Hl - 1, lef = ix >= 0, rig = ix + 1 <= Wl - 1;
                    s.flags = (top && lef ? 1 : 0) | (top && rig ? 2 : 0) | (bot && lef ? 4 : 0) |
                              (bot && rig ? 8 : 0);
                    const int wy = iy - wy0, wx = ix - wx0;
                    if ((unsigned)wy <= (unsigned)(WH - 2) && (unsigned)wx <= (unsigned)(WW - 2)) {
                        s.flags |= 16;
                        s.row = wy * WW + wx;
                        if (!(PYRB_ABLATE & 8)) {
                        if (s.flags & 1) atomicAdd(&hist[s.row], 1u);
                        if (s.flags & 2) atomicAdd(&hist[s.row + 1], 1u);
                        if (s.flags & 4) atomicAdd(&hist[s.row + WW], 1u);
                        if (s.flags & 8) atomicAdd(&hist[s.row + WW + 1], 1u);
                        }
                    }
                }
            }
            smp[k] = s;
        }
        if (l + 1 < 4) request_level(l + 1);
        TICK(1);                               // pass A: geometry + counts
        __syncthreads();
        TICK(2);                               // barriers + scan

        // ---- exclusive scan of the counts (two entries per thread) ------------------------------
        {
            const int i0 = 2 * tid, i1 = i0 + 1;
            const unsigned c0 = i0 <= rows ? hist[i0] : 0u, c1 = i1 <= rows ? hist[i1] : 0u;
            unsigned incl = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (lane == 63) scan[wave] = incl;
            __syncthreads();
            unsigned base = 0;
            for (int w = 0; w < wave; ++w) base += scan[w];
            const unsigned excl = base + incl - (c0 + c1);
            if (i0 <= rows) { hist[i0] = excl; cur[i0] = excl; }
            if (i1 <= rows) { hist[i1] = excl + c0; cur[i1] = excl + c0; }
        }
        __syncthreads();
        TICK(2);

        // ---- pass B: records, grad_attn / grad_loc ----------------------------------------------
#pragma unroll
        for (int k = 0; k < kMaxTasks; ++k) {
            const int t = wave + k * kWaves;
            if (t >= ntasks) break;                                   // wave-uniform
            const Sample s = smp[k];
            const int qi = t * 16 + slot;
            const bool live = qi < nq;
            const int qslot = live ? qi : nq - 1;
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const float c0 = hh * hw, c1 = hh * s.lw, c2 = s.lh * hw, c3 = s.lh * s.lw;
            const int iy = s.yx >> 16, ix = (int)(short)(s.yx & 0xffff);
            const unsigned pix = (unsigned)(stl + iy * Wl + ix) * row_stride;
            if (live && (s.flags & 16) && !(PYRB_ABLATE & 4)) {
                const unsigned lo = (unsigned)qslot;
#define DATR_REC(BIT, ROW, C)                                                                    \
                if (s.flags & (BIT)) {                                                           \
                    const unsigned pos = atomicAdd(&cur[ROW], 1u);                               \
                    recs[pos] = ((unsigned long long)__float_as_uint((C) * s.a) << 32) | lo;      \
                }
                DATR_REC(1, s.row, c0)
                DATR_REC(2, s.row + 1, c1)
                DATR_REC(4, s.row + WW, c2)
                DATR_REC(8, s.row + WW + 1, c3)
#undef DATR_REC
            } else if (live && (s.flags & 15)) {
                // outside the window: this lane adds its sample's contributions itself
                const f4 *gq = reinterpret_cast<const f4 *>(lds + kGoOff + qslot * kRowBytes);
#define DATR_DIRECT(BIT, OFF, C)                                                                 \
                if (s.flags & (BIT)) {                                                           \
                    const float w_ = (C) * s.a;                                                  \
                    for (int c = 0; c < 8; ++c) {                                                \
                        const f4 g_ = gq[c];                                                     \
                        const unsigned o_ = pix + (OFF) + (unsigned)c * 16u;                     \
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w_ * g_.x, gsrc, o_, 0, 0);       \
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w_ * g_.y, gsrc, o_ + 4, 0, 0);   \
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w_ * g_.z, gsrc, o_ + 8, 0, 0);   \
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w_ * g_.w, gsrc, o_ + 12, 0, 0);  \
                    }                                                                            \
                }
                DATR_DIRECT(1, 0u, c0)
                DATR_DIRECT(2, row_stride, c1)
                DATR_DIRECT(4, (unsigned)Wl * row_stride, c2)
                DATR_DIRECT(8, (unsigned)(Wl + 1) * row_stride, c3)
#undef DATR_DIRECT
            }

            if constexpr (kDots) {
            // corner offsets of this lane's sample (out-of-image corners read zeros)
            int g[4];
            g[0] = (int)((s.flags & 1) ? pix : kOutOfRange);
            g[1] = (int)((s.flags & 2) ? pix + row_stride : kOutOfRange);
            g[2] = (int)((s.flags & 4) ? pix + (unsigned)Wl * row_stride : kOutOfRange);
            g[3] = (int)((s.flags & 8) ? pix + (unsigned)(Wl + 1) * row_stride : kOutOfRange);
            // this quad's grad_out row: the lane's two 16-B pieces
            const f4 go0 = *reinterpret_cast<const f4 *>(lds + kGoOff + qslot * kRowBytes + chan);
            const f4 go1 = *reinterpret_cast<const f4 *>(lds + kGoOff + qslot * kRowBytes + chan2);

            float pa = 0.f, pw = 0.f, ph = 0.f;          // results of THIS lane's sample
#define DATR_POINT(SRC)                                                                          \
            {                                                                                    \
                float d[4];                                                                      \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                  \
                    const int o1 = quad_bcast<SRC>(g[c]) + chan, o2 = quad_bcast<SRC>(g[c]) + chan2; \
                    d[c] = dot4(go0, load_row4(vsrc, (unsigned)o1)) + dot4(go1, load_row4(vsrc, (unsigned)o2)); \
                }                                                                                \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) d[c] = quad_sum(d[c]);             \
                if (j == (SRC)) {                                                                \
                    pa = c0 * d[0] + c1 * d[1] + c2 * d[2] + c3 * d[3];                           \
                    pw = hh * (d[1] - d[0]) + s.lh * (d[3] - d[2]);                               \
                    ph = hw * (d[2] - d[0]) + s.lw * (d[3] - d[1]);                               \
                }                                                                                \
            }
            DATR_POINT(0)
            DATR_POINT(1)
            DATR_POINT(2)
            DATR_POINT(3)
#undef DATR_POINT
            if (live) {
                const size_t idx = (((size_t)n * Lq + qtab[qslot]) * M + m) * 16 + l * 4 + j;
                grad_attn[idx] = pa;
                reinterpret_cast<f2 *>(grad_loc)[idx] = f2{pw * s.a * Wf, ph * s.a * Hf};
            }
            }
        }
        TICK(3);                               // pass B: records, gathers, dots
        __syncthreads();
        TICK(2);

        // ---- reduce: a window row's gradient is a gather over its records -----------------------
        // 8 lanes x float4 accumulate a row; the finished rows of a wave (8 at a time) are then
        // TRANSPOSED through a 1 KB LDS scratch so that 32 consecutive lanes add 32 consecutive
        // dwords: float atomics retire per (instruction, 64-byte segment) -- ~20 G such requests/s
        // chip-wide (profiles/r01_probes.md) -- and the float4-per-lane pattern (lane stride 16 B, one
        // component per instruction) made 8 requests per row where this makes 2.  At ~1.65 M touched
        // rows per N = 4 call that was 13 M requests = 0.66 ms of the kernel's 0.75 ms.
        {
            const int grp = tid >> 3, c8 = tid & 7, g8 = lane >> 3;
            char *fl = lds + kCurOff + wave * kFlushPerWave;
            unsigned *fl_off = reinterpret_cast<unsigned *>(fl + 8 * kRowBytes);
            const int rounds = (rows + kThreads / 8 - 1) / (kThreads / 8);
            for (int it = 0; it < rounds; ++it) {
                const int r = grp + it * (kThreads / 8);
                unsigned beg = 0, end = 0;
                if (r < rows) { beg = hist[r]; end = hist[r + 1]; }
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                // kInFlight records in flight: record read -> grad_out row read -> FMA is a chain of
                // two LDS latencies, so the loop is unrolled to keep several chains going.  Slots past the
                // row's end re-read its last record with weight 0: guarding the reads instead (exec-masked
                // ds_read per slot) serialises them -- 497 against 371 us per N = 4 call.
                for (unsigned e = beg; e < ((PYRB_ABLATE & 2) ? beg : end); e += kInFlight) {
                    unsigned long long rec[kInFlight];
                    f4 g_[kInFlight];
                    float w_[kInFlight];
#if PYRB_REC_DPP
                    // ONE record per lane (lane c of the group reads record e + c) handed round the group's 8
                    // lanes by DPP: a quad broadcast gives a quad its own lane u, the half-row mirror the
                    // other quad's -- 8 x 64 B of LDS reads per 8 records become 64 B
                    static_assert(kInFlight == 8, "one record per lane of an 8-lane group");
                    (void)rec;
                    const unsigned long long mine = recs[min(e + (unsigned)c8, end - 1)];
                    const int lo_ = (int)(unsigned)(mine & 0xffffffffull);
                    const int hi_ = e + (unsigned)c8 < end ? (int)(unsigned)(mine >> 32) : 0;    // weight 0.0f past the end
                    const bool lower = c8 < 4;
                    int slot_[8];
                    auto hand_round = [&](auto uu) {
                        constexpr int U = decltype(uu)::value;
                        const int tl = quad_bcast<U>(lo_), th = quad_bcast<U>(hi_);
                        const int ml = __builtin_amdgcn_update_dpp(0, tl, 0x141, 0xF, 0xF, true);
                        const int mh = __builtin_amdgcn_update_dpp(0, th, 0x141, 0xF, 0xF, true);
                        slot_[U] = lower ? tl : ml;      slot_[U + 4] = lower ? ml : tl;
                        w_[U] = __int_as_float(lower ? th : mh);
                        w_[U + 4] = __int_as_float(lower ? mh : th);
                    };
                    hand_round(std::integral_constant<int, 0>{});
                    hand_round(std::integral_constant<int, 1>{});
                    hand_round(std::integral_constant<int, 2>{});
                    hand_round(std::integral_constant<int, 3>{});
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u)
                        g_[u] = *reinterpret_cast<const f4 *>(lds + kGoOff + slot_[u] * kRowBytes + c8 * 16);
#else
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) rec[u] = recs[min(e + u, end - 1)];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        w_[u] = e + u < end ? __uint_as_float((unsigned)(rec[u] >> 32)) : 0.f;
                        g_[u] = *reinterpret_cast<const f4 *>(
                            lds + kGoOff + (int)(rec[u] & 0xffffffffull) * kRowBytes + c8 * 16);
                    }
#endif
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        acc.x = fmaf(w_[u], g_[u].x, acc.x);
                        acc.y = fmaf(w_[u], g_[u].y, acc.y);
                        acc.z = fmaf(w_[u], g_[u].z, acc.z);
                        acc.w = fmaf(w_[u], g_[u].w, acc.w);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(beg != end) == 0) continue;      // wave-uniform: nothing to flush
                *reinterpret_cast<f4 *>(fl + g8 * kRowBytes + c8 * 16) = acc;
                if (c8 == 0) {
                    const int wy = r / WW, wx = r - wy * WW;
                    fl_off[g8] = beg != end ? (unsigned)(stl + (wy0 + wy) * Wl + wx0 + wx) * row_stride : kOutOfRange;
                }
                // same wave wrote and reads: LDS operations of a wave complete in order; the fences keep
                // the compiler from moving the cross-lane reads above the writes
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int g = 2 * k + (lane >> 5), l32 = lane & 31;
                    const unsigned o_ = fl_off[g];
                    const float v = *reinterpret_cast<const float *>(fl + g * kRowBytes + l32 * 4);
                    // an empty row has the out-of-range offset: the buffer atomic is dropped
                    if (!(PYRB_ABLATE & 1) || v == 123.456f)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, gsrc, o_ + (unsigned)l32 * 4u, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();      // the scratch is rewritten by the next round
            }
        }
        TICK(4);                               // reduce + flush
        __syncthreads();
        TICK(2);
    }
#ifdef PYR_PROBE
    if (lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&pyr_bwd_phase_cycles[(blockIdx.x * kWaves + wave) & 1023][i], ticks_[i]);
#endif
}

}  // namespace

// Region plan of the backward kernel; false when the shape is not covered.
// `envelope_host` ([8 heads][4 levels]{oy_lo, oy_hi, ox_lo, ox_hi} in pixels of the sampled level, or
// NULL): the measured reach of the samples (datr_amd/msda.py OffsetMonitor, the forward's window
// envelope).  With it the windows are exactly as wide as the samples reach; without it they are made as
// wide as the 1024-row histogram allows.  Either way a sample outside its window takes the direct-atomic
// path: results never depend on it.
// `small` (out): the plan is for the 256-thread / 128-query launch shape.
static bool bwd_pyr_plan(PyrMeta &pm, const int64_t *shapes_host, const int64_t *level_start_host,
                         int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                         const float *envelope_host = nullptr, bool *small = nullptr) {
    if (D != 32 || L != 4 || P != 4 || Lq != S || M < 1 || N < 1) return false;
    static const float halo = pyr_halo_from_env();
    if (small) *small = false;
    int kThreads = CfgLarge::kThreads, kMaxQ = CfgLarge::kMaxQ;
    const auto fits = [&](const PyrMeta &m_, int most_queries) {
        int rows = 0;
        for (int l = 0; l < 4; ++l) rows = std::max(rows, m_.WH[l] * m_.WW[l]);
        return rows <= kMaxRows && rows + 1 <= 2 * kThreads && most_queries <= kMaxQ;
    };
    if (envelope_host && M <= 8) {
        // Window rows cost histogram zeroing, scan and flush bookkeeping per level and workgroup:
        // 697 -> 632 us per N = 4 call at the model's offsets with 4.5 px instead of the widest halo
        // that fits (tools/probes/bwd_halo.sh; 3.5 px, too tight, sends samples down the slow path: 1835 us).
        // the envelope is a 0.5 % .. 99.5 % range: a margin keeps most of the stragglers off the slow path
        constexpr float kEnvelopeMargin = 0.5f;
        float h4[4];
        bool sane = true;
        for (int l = 0; l < 4; ++l) {
            float r = 0.f;
            for (int m_ = 0; m_ < (int)M; ++m_)
                for (int k = 0; k < 4; ++k) {
                    const float v = envelope_host[(m_ * 4 + l) * 4 + k];
                    if (!(v == v)) sane = false;
                    r = std::max(r, std::fabs(v));
                }
            h4[l] = std::min(std::max(r + kEnvelopeMargin, 1.0f), 12.0f);
        }
        // the small launch shape while the reach is short: its 128-query regions carry the halo of a
        // region twice the size
        static const float small_reach = getenv("DATR_MSDA_PYRB_SMALL_REACH") ? (float)atof(getenv("DATR_MSDA_PYRB_SMALL_REACH")) : 6.0f;
        const float reach = std::max(std::max(h4[0], h4[1]), std::max(h4[2], h4[3]));
        if (sane && reach <= small_reach) {
            kThreads = CfgSmall::kThreads; kMaxQ = CfgSmall::kMaxQ;
            if (small) *small = true;
        }
        if (sane && build_pyr_meta(pm, shapes_host, level_start_host, S, h4, kMaxQ >= 512 ? 12.5 : 10.0,
                                   kMaxQ >= 512 ? 28.0 : 16.7, fits)) {
            // A head looks in one direction (the ring initialisation of ms_deform_attn.py:59-68): trim the
            // symmetric window [lo - h, hi + h] to the head's [lo + e_lo - margin, hi + e_hi + margin].
            // floor(a) + floor(b) <= floor(a + b) keeps the integer trims on the safe side.
            for (int m_ = 0; m_ < (int)M; ++m_)
                for (int l = 0; l < 4; ++l) {
                    const float *e = envelope_host + (m_ * 4 + l) * 4;
                    const int y0 = std::max(0, (int)std::floor(h4[l] + e[0] - kEnvelopeMargin));
                    const int y1 = std::max(0, (int)std::floor(h4[l] - e[1] - kEnvelopeMargin));
                    const int x0 = std::max(0, (int)std::floor(h4[l] + e[2] - kEnvelopeMargin));
                    const int x1 = std::max(0, (int)std::floor(h4[l] - e[3] - kEnvelopeMargin));
                    const int cy = std::min(y0 + y1, pm.WH[l] - 2), cx = std::min(x0 + x1, pm.WW[l] - 2);
                    pm.cut_y0[m_][l] = (short)std::min(y0, cy); pm.cut_y[m_][l] = (short)cy;
                    pm.cut_x0[m_][l] = (short)std::min(x0, cx); pm.cut_x[m_][l] = (short)cx;
                }
            return true;
        }
        kThreads = CfgLarge::kThreads; kMaxQ = CfgLarge::kMaxQ;
        if (small) *small = false;
    }
    if (!build_pyr_meta(pm, shapes_host, level_start_host, S, halo, kMaxQ >= 512 ? 12.5 : 10.0,
                        kMaxQ >= 512 ? 28.0 : 16.7, fits))
        return false;
    // The windows are index spaces here (nothing is staged), so the halo may be as wide as the
    // 1024-row histogram allows: samples beyond it take the slow direct-atomic path.
    static const bool widen = !(getenv("DATR_MSDA_PYRB_WIDEN") && atoi(getenv("DATR_MSDA_PYRB_WIDEN")) == 0);
    if (widen) {
        static const float wide[3][4] = {{6.f, 8.f, 10.f, 12.f}, {5.5f, 7.f, 9.f, 11.f}, {5.f, 6.f, 7.f, 8.f}};
        for (int i = 0; i < 3; ++i) {
            float h4[4];
            for (int l = 0; l < 4; ++l) h4[l] = std::max(halo, wide[i][l]);
            PyrMeta wider;
            if (build_pyr_meta(wider, shapes_host, level_start_host, S, h4, 10.0, 16.7, fits, pm.nRy, pm.nRx)) {
                pm = wider;
                break;
            }
        }
    }
    return true;
}

// info[0..2] = {covered, nRy, nRx} of the backward plan (no launch)
DATR_INTERNAL int datr_internal_msda_bwd_pyr_plan(const int64_t *shapes_host, const int64_t *level_start_host,
                                               int64_t S, int64_t M, int32_t *info) {
    PyrMeta pm;
    const bool ok = bwd_pyr_plan(pm, shapes_host, level_start_host, 1, S, M, 32, 4, S, 4);
    info[0] = ok; info[1] = ok ? pm.nRy : 0; info[2] = ok ? pm.nRx : 0;
    return ok ? DATR_OK : DATR_EUNSUPPORTED;
}

// Internal entry (msda.hip dispatches here): DATR_EUNSUPPORTED when the shape is not covered.
// grad_value must be zero-filled by the caller.
// query_grad != 0: `grad_loc` receives the gradient of the module's merged query projection ([N * Lq, M * 48], see
// msda_fwd_pyr2.hip) in place of grad_loc / grad_attn; DATR_EUNSUPPORTED with nothing launched when the LDS-window
// kernel does not cover the shape.
DATR_INTERNAL int datr_internal_msda_bwd_pyr_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const int64_t *shapes_host, const int64_t *level_start_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, const float *envelope_host, float *grad_value,
    float *grad_loc, float *grad_attn, void *stream, int query_grad)
{
    PyrMeta pm;
    bool small = false;
    if (!bwd_pyr_plan(pm, shapes_host, level_start_host, N, S, M, D, L, Lq, P, envelope_host, &small)) return DATR_EUNSUPPORTED;
    {
#if PYRB_BANDS
        const long blocks_ = (long)N * ((pm.nRy * pm.nRx + 7) / 8) * 8 * M;
#else
        const long blocks_ = (long)N * pm.nRy * pm.nRx * M;
#endif
        if (blocks_ <= 0 || blocks_ >= (1L << 31)) return DATR_EUNSUPPORTED;
    }
    // grad_loc / grad_attn out of LDS windows by the forward's structure where its plan covers the shape
    // (DATR_MSDA_BWD_SPLIT=0: this kernel's own gathers, for A/B measurements)
    static const bool split = !(getenv("DATR_MSDA_BWD_SPLIT") && atoi(getenv("DATR_MSDA_BWD_SPLIT")) == 0);
    bool dots_done = false;
    if (query_grad) {
        const int rc = datr_internal_msda_bwd_dots_pyr2_d32(grad_out, value, loc, attn, shapes_host, level_start_host,
                                                            envelope_host, N, S, M, D, L, Lq, P, grad_loc, nullptr,
                                                            stream, 1);
        if (rc != DATR_OK) return rc;
        dots_done = true;
    } else if (split) {
        const int rc = datr_internal_msda_bwd_dots_pyr2_d32(grad_out, value, loc, attn, shapes_host, level_start_host,
                                                            envelope_host, N, S, M, D, L, Lq, P, grad_loc, grad_attn,
                                                            stream);
        if (rc == DATR_OK) dots_done = true;
        else if (rc != DATR_EUNSUPPORTED) return rc;
    }
#if PYRB_BANDS
    const long blocks = (long)N * ((pm.nRy * pm.nRx + 7) / 8) * 8 * M;
#else
    const long blocks = (long)N * pm.nRy * pm.nRx * M;
#endif
    if (blocks <= 0 || blocks >= (1L << 31)) return DATR_EUNSUPPORTED;
    const auto launch = [&](auto cfg, auto dots) -> int {
        using C = decltype(cfg);
        constexpr bool kDots = decltype(dots)::value;
        static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(msda_bwd_pyr_d32<C, kDots>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        C::kLdsBytes) == hipSuccess;
        if (!attr_ok) return DATR_EUNSUPPORTED;
        hipLaunchKernelGGL((msda_bwd_pyr_d32<C, kDots>), dim3((unsigned)blocks), dim3(C::kThreads), (size_t)C::kLdsBytes,
                           (hipStream_t)stream, grad_out, value, loc, attn, pm, (int)S, (int)M, grad_value,
                           grad_loc, grad_attn);
        return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
    };
    if (small) return dots_done ? launch(CfgSmall{}, std::false_type{}) : launch(CfgSmall{}, std::true_type{});
    return dots_done ? launch(CfgLarge{}, std::false_type{}) : launch(CfgLarge{}, std::true_type{});
}
