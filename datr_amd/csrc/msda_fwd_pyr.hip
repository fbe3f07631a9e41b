// msda_fwd_pyr.hip -- MSDA forward for the encoder calls (Lq == S: the queries ARE the pixels of
// the pyramid; D == 32, L == P == 4), organised around PYRAMID REGIONS.
//
// Reference behaviour: /root/reference/models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299
// (sampling + aggregation; pixel mapping h = y*H - 0.5, in-range test :285-288).
//
// Why a second forward kernel.  The row kernel (msda.hip) fetches 64 corner rows x 128 B per
// output row through the vector-memory path: 5.8 GB per N=4 encoder call against 319 MB of
// algorithmic bytes.  That path tops out at ~24 TB/s chip-wide on gfx950 whether the rows hit
// L1 or L2 (profiles/r01_probes.md), i.e. >= 240 us -- the kernel sits on that floor.  LDS serves
// ds_read_b128 at up to 256 B/clk/CU (~150 TB/s), so the way below the floor is to gather from
// LDS -- for as many of the rows as fit -- while the vector-memory path works on the rest in
// parallel.
//
// Decomposition.  The image plane is cut into nRy x nRx regions (about 17 x 42 level-0 pixels:
// 6 x 4 regions at 1333x800).  A 768-thread workgroup owns one (image, region, head): ALL queries
// whose reference point lies in the region -- the region's pixels of level 0 and the matching
// (4x, 16x, 64x fewer) pixels of levels 1-3.  Queries sample around their reference point in every
// level, so the workgroup stages, for levels 1, 2, 3, the window of value rows its queries can
// reach (region footprint + halo 4.5 px: the module's initial offset ring) in LDS once, by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: no staging registers), and every sample of those three levels --
// 3/4 of all gathered bytes -- is a ds_read_b128.  Level 0 (whose window would not fit beside the
// others) is gathered with buffer loads as in the row kernel: 1/4 of the rows on the
// vector-memory path, in flight while the wave works through the LDS gathers.  Out-of-image window
// pixels are zero-filled, so border corners need no special casing there; a sample that lands
// outside its window (learned offsets larger than the halo) is fetched from global memory by a
// slow path at the end of the task, so results never depend on the window heuristic.
//
// Lane mapping (no LDS staging of per-sample geometry at all).  4 lanes share a query ("quad"),
// 16 queries per wave and task.  Lane j of the quad (a) works out the geometry of POINT j of each
// of the 4 levels -- for level 0 four corner offsets, for a window level one LDS address, plus 4
// bilinear weights each, all in registers -- and (b) owns the two 16-B pieces 16 j and 16 j + 64
// of every 128-B row, i.e. channels 4j..4j+3 and 16+4j..19+4j.  In the gather loop the geometry
// travels inside the quad with DPP quad_perm broadcasts (addresses through one v_add_u32_dpp
// that also adds the lane's channel offset).  One load / store instruction of a quad touches 64
// contiguous bytes; odd quads take the two halves in the opposite order, so the 16-lane groups a
// ds_read_b128 is served in cover different 16-bank quarters whenever their rows differ in parity
// (MI355X_MICROARCH.md, LDS).  The next task's sampling locations are requested before the
// current task's gathers start.
//
// Measured (profiles/r02_msda_fwd_pyr.md): N=4 encoder call 167 us against 201 us for the row
// kernel.  Ablations: VALU alone 71 us, + location loads ~35-58 us, + level-0 gathers ~45 us,
// + stores ~16 us, + fills ~10 us: the vector-memory path (level-0 gathers, location loads, fills)
// is still the busiest unit; see DESIGN.md for what the next step is.
//
// blockIdx -> (image, region, head) with the head fastest: workgroup b runs on XCD b % 8, so an
// XCD's 4 MiB L2 holds one head's slice of the pyramid (2.8 MB at 1333x800) -- affinity only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "datr_hip.h"
#include "msda_tiled.h"
#include "msda_pyr.h"

// Development only (tools/probes/pyr_ablate.sh): compile pieces out to see what they cost.
// 1 = no window fill, 2 = no level-0 (vector-memory) gathers, 4 = no LDS gathers,
// 8 = no loc/attn loads, 16 = no output stores.  Results are wrong with any bit set.
#ifndef PYR_ABLATE
#define PYR_ABLATE 0
#endif
#ifndef PYR_THREADS
#define PYR_THREADS 768
#endif
#ifndef PYR_GIF
#define PYR_GIF 2
#endif
#ifndef PYR_LDS_DEPTH
#define PYR_LDS_DEPTH 1
#endif

#ifdef PYR_PROBE
// per-phase cycle counters of lane 0 of every wave (development; tools/probes/pyr_ablate.sh)
__device__ unsigned long long pyr_phase_cycles[1024][8];   // many sets: one hot address would serialise
#define TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
                     ticks_[i] += now_ - tick_; tick_ = now_; } while (0)
extern "C" void datr_probe_pyr_phase_cycles(unsigned long long *out, int reset) {
    static unsigned long long all[1024][8];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(pyr_phase_cycles), sizeof(all));
    for (int i = 0; i < 8; ++i) {
        out[i] = 0;
        for (int b = 0; b < 1024; ++b) out[i] += all[b][i];
    }
    if (reset) {
        static unsigned long long z[1024][8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pyr_phase_cycles), z, sizeof(z));
    }
}
#else
#define TICK(i) do {} while (0)
#endif

namespace {

constexpr int kThreads = PYR_THREADS;
constexpr int kWaves = kThreads / 64;
constexpr int kGlobalInFlight = PYR_GIF;        // level-0 samples in flight beside the LDS gathers
constexpr unsigned kOutOfRange = 0x80000000u;    // >= num_records of every descriptor built here
constexpr int kRowBytes = 128;                   // D = 32 floats
constexpr int kMaxLds = 156 * 1024;              // windows + query table
constexpr int kMaxQueries = 2048;                // query-table entries (8 KiB)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ f4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    // NB: keep `auto` -- converting the builtin's result to an ext_vector typedef splats lane 0.
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128");
    return __builtin_bit_cast(f4, r);
}

// broadcast lane `SRC` of every quad to the 4 lanes of the quad (DPP quad_perm [s,s,s,s])
template <int SRC>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, quad_bcast<SRC>(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ void fma4(f4 &acc, float w, const f4 v) {
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z);
    acc.w = fmaf(w, v.w, acc.w);
}

// Program-order fence for the staged gathers: the accumulators must hold everything accumulated
// so far, and no memory access moves across.  (`sched_barrier` alone does not do it: the FMAs are
// pure, so instruction selection is free to sink them below the fence.)
#define PIN(a, b) asm volatile("" : "+v"(a), "+v"(b) : : "memory")

struct Rows { f4 a[4], b[4]; };      // the lane's two 16-B pieces of the 4 corner rows of a sample

// Level-0 sample whose 4 corner offsets lane SRC of the quad holds: zero-filling buffer loads.
template <int SRC>
__device__ __forceinline__ void fetch_global(Rows &r, const int (&g)[4], const int chan,
                                             const int chan2, __amdgpu_buffer_rsrc_t rsrc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o1 = quad_bcast<SRC>(g[k]) + chan, o2 = quad_bcast<SRC>(g[k]) + chan2;
        if (PYR_ABLATE & 2) {
            r.a[k] = r.b[k] = f4{__builtin_bit_cast(float, o1 ^ o2), 0.f, 0.f, 0.f};
        } else {
            r.a[k] = load_row4(rsrc, (unsigned)o1);
            r.b[k] = load_row4(rsrc, (unsigned)o2);
        }
    }
}

// Window sample whose top-left LDS address lane SRC holds; row_bytes = window width * 128.
template <int SRC>
__device__ __forceinline__ void fetch_lds(Rows &r, const int base, const int chan, const int chan2,
                                          const int row_bytes) {
    const int t1 = quad_bcast<SRC>(base) + chan, t2 = quad_bcast<SRC>(base) + chan2;
    const int u1 = t1 + row_bytes, u2 = t2 + row_bytes;
    if (PYR_ABLATE & 4) {
        r.a[0] = r.a[1] = r.b[0] = r.b[1] = f4{__builtin_bit_cast(float, t1 ^ t2), 0.f, 0.f, 0.f};
        r.a[2] = r.a[3] = r.b[2] = r.b[3] = f4{__builtin_bit_cast(float, u1 ^ u2), 0.f, 0.f, 0.f};
        return;
    }
    // the addresses ARE LDS addresses: the dynamic LDS block starts at 0 (no static LDS here);
    // the right-hand corners are the next 128-B row: an immediate offset
    r.a[0] = *reinterpret_cast<const lds_f4 *>((unsigned)t1);
    r.b[0] = *reinterpret_cast<const lds_f4 *>((unsigned)t2);
    r.a[1] = *reinterpret_cast<const lds_f4 *>((unsigned)t1 + kRowBytes);
    r.b[1] = *reinterpret_cast<const lds_f4 *>((unsigned)t2 + kRowBytes);
    r.a[2] = *reinterpret_cast<const lds_f4 *>((unsigned)u1);
    r.b[2] = *reinterpret_cast<const lds_f4 *>((unsigned)u2);
    r.a[3] = *reinterpret_cast<const lds_f4 *>((unsigned)u1 + kRowBytes);
    r.b[3] = *reinterpret_cast<const lds_f4 *>((unsigned)u2 + kRowBytes);
}

template <int SRC>
__device__ __forceinline__ void accumulate(f4 &acc0, f4 &acc1, const Rows &r, const float (&w)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = quad_bcast<SRC>(w[k]);
        fma4(acc0, wk, r.a[k]);
        fma4(acc1, wk, r.b[k]);
    }
}

// The four points of one window level (point p's geometry sits in lane p of the quad), two
// samples in flight.
__device__ __forceinline__ void lds_level(f4 &acc0, f4 &acc1, const int base, const float (&w)[4],
                                          const int chan, const int chan2, const int row_bytes) {
#if PYR_LDS_DEPTH == 1
    Rows r;
    fetch_lds<0>(r, base, chan, chan2, row_bytes);
    accumulate<0>(acc0, acc1, r, w);
    PIN(acc0, acc1);
    fetch_lds<1>(r, base, chan, chan2, row_bytes);
    accumulate<1>(acc0, acc1, r, w);
    PIN(acc0, acc1);
    fetch_lds<2>(r, base, chan, chan2, row_bytes);
    accumulate<2>(acc0, acc1, r, w);
    PIN(acc0, acc1);
    fetch_lds<3>(r, base, chan, chan2, row_bytes);
    accumulate<3>(acc0, acc1, r, w);
    return;
#endif
    Rows r0, r1;
    fetch_lds<0>(r0, base, chan, chan2, row_bytes);
    fetch_lds<1>(r1, base, chan, chan2, row_bytes);
    PIN(acc0, acc1);
    accumulate<0>(acc0, acc1, r0, w);
    fetch_lds<2>(r0, base, chan, chan2, row_bytes);
    PIN(acc0, acc1);
    accumulate<1>(acc0, acc1, r1, w);
    fetch_lds<3>(r1, base, chan, chan2, row_bytes);
    PIN(acc0, acc1);
    accumulate<2>(acc0, acc1, r0, w);
    PIN(acc0, acc1);
    accumulate<3>(acc0, acc1, r1, w);
}

// pixel coordinates of a sample: floor, fractions, in-range test of cuh:285-288
struct Pix { int iy, ix; float lh, lw; bool inside; };
__device__ __forceinline__ Pix locate(float x, float y, int H, int W) {
    Pix r;
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = y * Hf - 0.5f, w_im = x * Wf - 0.5f;
    r.inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
    const float hf = floorf(h_im), wf = floorf(w_im);
    r.lh = h_im - hf;
    r.lw = w_im - wf;
    r.iy = r.inside ? (int)hf : 0;
    r.ix = r.inside ? (int)wf : 0;
    return r;
}
__device__ __forceinline__ void corner_weights(float (&w)[4], const Pix &p, float a) {
    const float hh = 1.f - p.lh, hw = 1.f - p.lw;
    const float ah = a * hh, al = a * p.lh;
    w[0] = ah * hw;
    w[1] = ah * p.lw;
    w[2] = al * hw;
    w[3] = al * p.lw;
}

__global__ __launch_bounds__(kThreads) void msda_fwd_pyr_d32(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const PyrMeta pm, int S, int M, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int m = bid % M;
    const int reg = (bid / M) % (pm.nRy * pm.nRx);
    const int n = bid / (M * pm.nRy * pm.nRx);
    const int ry = reg / pm.nRx, rx = reg % pm.nRx;
    const unsigned row_stride = (unsigned)M * kRowBytes;           // bytes between pixels of one head
    const int lane = tid & 63, wave = tid >> 6;
#ifdef PYR_PROBE
    unsigned long long ticks_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = __builtin_readcyclecounter();
#endif

    const float *base = value + ((size_t)n * S * M + m) * 32;
    const int records = (S * M - m) * kRowBytes;                   // bytes from `base` to the end of item n
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, records, 0x00020000);

    // ---- stage the windows of levels 1..3 by LDS-DMA ------------------------------------------
    // One wave instruction moves 64 x 16 B = 8 window pixels straight into LDS (no staging
    // registers, no ds_write pass); lane i of the instruction lands at base + 16 i.  An
    // out-of-image pixel gets an out-of-range offset: the buffer load returns zeros for it.
    if (!(PYR_ABLATE & 1)) {
#pragma unroll
        for (int l = 1; l < 4; ++l) {
            const int WW = pm.WW[l], cnt = pm.WH[l] * WW * 8;
            const int wy0 = pm.wy0[l][ry], wx0 = pm.wx0[l][rx];
            const int Hl = pm.H[l], Wl = pm.W[l], st = pm.start[l];
            const float inv = 1.0f / (float)WW;
            const int lbase = pm.lds_base[l];
            for (int i0 = wave * 64; i0 < cnt; i0 += kThreads) {
                const int i = i0 + lane;
                const int pix = i >> 3, chunk = i & 7;
                const int wr = (int)(((float)pix + 0.5f) * inv);
                const int wc = pix - wr * WW;
                const int y = wy0 + wr, x = wx0 + wc;
                const bool in = i < cnt && (unsigned)y < (unsigned)Hl && (unsigned)x < (unsigned)Wl;
                const unsigned off = in ? (unsigned)(st + y * Wl + x) * row_stride + (unsigned)chunk * 16u
                                        : kOutOfRange;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<lds_void *>(lbase + i0 * 16),
                                                         16, (int)off, 0, 0, 0);
            }
        }
    }

    // ---- the region's queries, level by level, into a table (query slot -> pyramid index) -----
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
    {
        int *qt = reinterpret_cast<int *>(lds + pm.lds_bytes);
        for (int qi = tid; qi < nq; qi += kThreads) qt[qi] = decode(qi);
    }
    const int ntasks = (nq + 15) >> 4;
    const int *qtab = reinterpret_cast<const int *>(lds + pm.lds_bytes);

    // Lane roles.  4 lanes share a query; lane j owns the two 16-B pieces 16 j and 16 j + 64 of
    // every 128-B row (so one load / store instruction of a quad touches 64 contiguous bytes;
    // odd quads take the halves in the opposite order, which spreads a ds_read_b128 lane group
    // over different 16-bank quarters whenever its rows differ in parity), and works out the
    // geometry of POINT j of each of the four levels.
    const int slot = lane >> 2, j = lane & 3;
    const int chan = 16 * j + 64 * (slot & 1), chan2 = chan ^ 64;
    const size_t Lq = (size_t)S;
    const int H0 = pm.H[0], W0 = pm.W[0], start0 = pm.start[0];

    // locations + weights of point j in the four levels of the query task t assigns to this quad
    struct Task { f2 xy[4]; float a[4]; size_t qm; bool live; };
    auto load_query = [&](int q, bool live) {
        Task tk;
        tk.live = live;
        tk.qm = ((size_t)n * Lq + q) * M + m;
        const f2 *lp = reinterpret_cast<const f2 *>(loc) + tk.qm * 16 + j;
        const float *ap = attn + tk.qm * 16 + j;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
#if PYR_ABLATE & 8
            {   // no loads: the query's own reference point (in every window), constant weight
                const int lq_ = (q >= pm.start[1]) + (q >= pm.start[2]) + (q >= pm.start[3]);
                const int li_ = q - pm.start[lq_];
                const int y_ = (int)(((float)li_ + 0.5f) / (float)pm.W[lq_]);
                tk.xy[l] = f2{((float)(li_ - y_ * pm.W[lq_]) + 0.5f) / (float)pm.W[lq_],
                              ((float)y_ + 0.5f) / (float)pm.H[lq_]};
                tk.a[l] = 0.0625f;
            }
#else
            tk.xy[l] = __builtin_nontemporal_load(lp + 4 * l);
            tk.a[l] = __builtin_nontemporal_load(ap + 4 * l);
#endif
        }
        return tk;
    };
    auto load_task = [&](int t) {
        const int qi = t * 16 + slot;
        return load_query(qtab[qi < nq ? qi : nq - 1], qi < nq);
    };

    // the first task's locations are requested before the windows have landed (their query index
    // is worked out directly: the table is not visible yet)
    Task next;
    {
        const int qi = (wave < ntasks ? wave : 0) * 16 + slot;
        next = load_query(decode(qi < nq ? qi : nq - 1), qi < nq);
    }
    TICK(0);                                             // fill issue + query table
    // this wave's LDS-DMA pieces have landed (loads return in order: only the 8 location loads
    // issued after them may still be in flight)
#if PYR_ABLATE & 8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    TICK(1);                                             // own pieces landing
    __syncthreads();
    TICK(2);                                             // barrier
#ifndef PYR_PROBE
    if (wave >= ntasks) return;
#endif

    for (int t = wave; t < ntasks; t += kWaves) {
        const Task cur = next;

        // ---- geometry of point j: level 0 (global offsets), levels 1..3 (window addresses) ------
        int g[4];
        float w0[4];
        {
            const Pix p = locate(cur.xy[0].x, cur.xy[0].y, H0, W0);
            const bool top = p.iy >= 0, bot = p.iy + 1 <= H0 - 1, lef = p.ix >= 0, rig = p.ix + 1 <= W0 - 1;
            const unsigned pix = (unsigned)(start0 + p.iy * W0 + p.ix) * row_stride;
            g[0] = (int)((p.inside && top && lef) ? pix : kOutOfRange);
            g[1] = (int)((p.inside && top && rig) ? pix + row_stride : kOutOfRange);
            g[2] = (int)((p.inside && bot && lef) ? pix + (unsigned)W0 * row_stride : kOutOfRange);
            g[3] = (int)((p.inside && bot && rig) ? pix + (unsigned)(W0 + 1) * row_stride : kOutOfRange);
            corner_weights(w0, p, p.inside ? cur.a[0] : 0.f);
        }
        int wb[3];
        float ww[3][4];
        int missmask = 0;
#pragma unroll
        for (int l = 1; l < 4; ++l) {
            const Pix p = locate(cur.xy[l].x, cur.xy[l].y, pm.H[l], pm.W[l]);
            const int wy = p.iy - pm.wy0[l][ry], wx = p.ix - pm.wx0[l][rx];
            const bool inwin = (unsigned)wy <= (unsigned)(pm.WH[l] - 2) && (unsigned)wx <= (unsigned)(pm.WW[l] - 2);
            const bool use = p.inside && inwin;
            missmask |= (p.inside && !inwin) ? (1 << l) : 0;
            wb[l - 1] = pm.lds_base[l] + ((inwin ? wy : 0) * pm.WW[l] + (inwin ? wx : 0)) * kRowBytes;
            corner_weights(ww[l - 1], p, use ? cur.a[l] : 0.f);
        }

        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#ifdef PYR_PROBE
        asm volatile("" : "+v"(g[0]), "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]) : : "memory");
#endif
        TICK(3);                                         // location wait + geometry

        // the next task's locations travel while this one is gathered
        {
            const int tn = t + kWaves;
            if (tn < ntasks) next = load_task(tn);
        }

        // ---- level 0 through the vector-memory path, levels 1..3 out of LDS ---------------------
        // Level-0 samples are in flight on the vector-memory path while a window level's sixteen
        // corner rows come out of LDS; the fences keep the compiler from hoisting every load of
        // the task to the top (which overflows the register file).
        if (kGlobalInFlight == 4) {
            Rows ga, gb, gc, gd;
            fetch_global<0>(ga, g, chan, chan2, rsrc);
            fetch_global<1>(gb, g, chan, chan2, rsrc);
            fetch_global<2>(gc, g, chan, chan2, rsrc);
            fetch_global<3>(gd, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[0], ww[0], chan, chan2, pm.WW[1] * kRowBytes);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[1], ww[1], chan, chan2, pm.WW[2] * kRowBytes);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[2], ww[2], chan, chan2, pm.WW[3] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<0>(acc0, acc1, ga, w0);
            accumulate<1>(acc0, acc1, gb, w0);
            accumulate<2>(acc0, acc1, gc, w0);
            accumulate<3>(acc0, acc1, gd, w0);
        } else if (kGlobalInFlight == 2) {
            Rows ga, gb;
            fetch_global<0>(ga, g, chan, chan2, rsrc);
            fetch_global<1>(gb, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[0], ww[0], chan, chan2, pm.WW[1] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<0>(acc0, acc1, ga, w0);
            fetch_global<2>(ga, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[1], ww[1], chan, chan2, pm.WW[2] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<1>(acc0, acc1, gb, w0);
            fetch_global<3>(gb, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[2], ww[2], chan, chan2, pm.WW[3] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<2>(acc0, acc1, ga, w0);
            accumulate<3>(acc0, acc1, gb, w0);
        } else {
            Rows ga;
            fetch_global<0>(ga, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[0], ww[0], chan, chan2, pm.WW[1] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<0>(acc0, acc1, ga, w0);
            fetch_global<1>(ga, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[1], ww[1], chan, chan2, pm.WW[2] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<1>(acc0, acc1, ga, w0);
            fetch_global<2>(ga, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            lds_level(acc0, acc1, wb[2], ww[2], chan, chan2, pm.WW[3] * kRowBytes);
            PIN(acc0, acc1);
            accumulate<2>(acc0, acc1, ga, w0);
            fetch_global<3>(ga, g, chan, chan2, rsrc);
            PIN(acc0, acc1);
            accumulate<3>(acc0, acc1, ga, w0);
        }

        TICK(4);                                         // gathers
        // ---- slow path (rare): window samples that fell outside their window come from global
        // memory.  It runs last and re-reads the sample's location, so that nothing it needs stays
        // in registers across the gathers above.
        if (__builtin_amdgcn_ballot_w64(missmask != 0) != 0) {
            for (int l = 1; l < 4; ++l) {
                const int Hl = pm.H[l], Wl = pm.W[l], stl = pm.start[l];
                auto one = [&](int flag, int p) {
                    // only the quads whose sample missed execute this (divergent branch: the
                    // vector-memory path is charged per active lane), so the cost follows the
                    // number of misses, not the number of waves that contain one
                    if (flag) {
                        const f2 xy = reinterpret_cast<const f2 *>(loc)[cur.qm * 16 + l * 4 + p];
                        const float a_ = attn[cur.qm * 16 + l * 4 + p];
                        const Pix px = locate(xy.x, xy.y, Hl, Wl);
                        const bool ok = px.inside;
                        const bool top = px.iy >= 0, bot = px.iy + 1 <= Hl - 1, lef = px.ix >= 0, rig = px.ix + 1 <= Wl - 1;
                        const unsigned pix = (unsigned)(stl + px.iy * Wl + px.ix) * row_stride + (unsigned)chan;
                        const unsigned o0 = (ok && top && lef) ? pix : kOutOfRange;
                        const unsigned o1 = (ok && top && rig) ? pix + row_stride : kOutOfRange;
                        const unsigned o2 = (ok && bot && lef) ? pix + (unsigned)Wl * row_stride : kOutOfRange;
                        const unsigned o3 = (ok && bot && rig) ? pix + (unsigned)(Wl + 1) * row_stride : kOutOfRange;
                        float w_[4];
                        corner_weights(w_, px, ok ? a_ : 0.f);
                        const f4 r0 = load_row4(rsrc, o0), s0 = load_row4(rsrc, o0 ^ 64u);
                        const f4 r1 = load_row4(rsrc, o1), s1 = load_row4(rsrc, o1 ^ 64u);
                        const f4 r2 = load_row4(rsrc, o2), s2 = load_row4(rsrc, o2 ^ 64u);
                        const f4 r3 = load_row4(rsrc, o3), s3 = load_row4(rsrc, o3 ^ 64u);
                        fma4(acc0, w_[0], r0); fma4(acc1, w_[0], s0);
                        fma4(acc0, w_[1], r1); fma4(acc1, w_[1], s1);
                        fma4(acc0, w_[2], r2); fma4(acc1, w_[2], s2);
                        fma4(acc0, w_[3], r3); fma4(acc1, w_[3], s3);
                    }
                };
                one((quad_bcast<0>(missmask) >> l) & 1, 0);
                one((quad_bcast<1>(missmask) >> l) & 1, 1);
                one((quad_bcast<2>(missmask) >> l) & 1, 2);
                one((quad_bcast<3>(missmask) >> l) & 1, 3);
            }
        }

        if (cur.live && (!(PYR_ABLATE & 16) || acc0.x == 123.456f)) {
            float *dst = out + cur.qm * 32;
            __builtin_nontemporal_store(acc0, reinterpret_cast<f4 *>(dst + (chan >> 2)));
            __builtin_nontemporal_store(acc1, reinterpret_cast<f4 *>(dst + (chan2 >> 2)));
        }
        TICK(5);                                         // slow path + stores
    }
#ifdef PYR_PROBE
    __syncthreads();
    TICK(6);                                             // waiting for the workgroup's last wave
    if (lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&pyr_phase_cycles[(blockIdx.x * kWaves + wave) & 1023][i], ticks_[i]);
#endif
}

}  // namespace

// Internal entry (msda.hip dispatches here): DATR_EUNSUPPORTED when the shape is not covered.
DATR_INTERNAL int datr_internal_msda_fwd_pyr_d32(
    const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L,
    int64_t Lq, int64_t P, float *out, void *stream)
{
    if (D != 32 || L != 4 || P != 4 || Lq != S || M < 1 || N < 1) return DATR_EUNSUPPORTED;
    static const float halo = pyr_halo_from_env();
    PyrMeta pm;
    // regions of about 17 x 42 level-0 pixels (6 x 4 at 1333x800), shrunk until the windows of
    // levels 1..3 and the query table fit the LDS
    const auto fits = [](const PyrMeta &m_, int most_queries) {
        int bytes = 0;
        for (int l = 1; l < 4; ++l) bytes += (m_.WH[l] * m_.WW[l] * kRowBytes + 1023) & ~1023;
        return bytes + 4 * kMaxQueries <= kMaxLds && most_queries <= kMaxQueries;
    };
    if (!build_pyr_meta(pm, shapes_host, level_start_host, S, halo, 16.7, 41.75, fits))
        return DATR_EUNSUPPORTED;
    {
        int bytes = 0;
        for (int l = 1; l < 4; ++l) {
            pm.lds_base[l] = bytes;
            // whole 1 KiB LDS-DMA pieces: the zero-filled tail lanes of a level's last piece must
            // not land in the next level's window
            bytes += (pm.WH[l] * pm.WW[l] * kRowBytes + 1023) & ~1023;
        }
        pm.lds_base[0] = 0;
        pm.lds_bytes = bytes;
    }
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_pyr_d32),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kMaxLds) == hipSuccess;
    if (!attr_ok) return DATR_EUNSUPPORTED;
    const long blocks = (long)N * pm.nRy * pm.nRx * M;
    if (blocks <= 0 || blocks >= (1L << 31)) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(msda_fwd_pyr_d32, dim3((unsigned)blocks), dim3(kThreads),
                       (size_t)pm.lds_bytes + 4 * kMaxQueries,
                       (hipStream_t)stream, value, loc, attn, pm, (int)S, (int)M, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
