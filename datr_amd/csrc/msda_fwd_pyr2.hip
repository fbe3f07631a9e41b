// msda_fwd_pyr2.hip -- MSDA forward for the encoder calls (Lq == S, D == 32, L == P == 4), round 3:
// ALL FOUR LEVELS gathered out of LDS, in phases, windows sized by a per-head offset envelope.
//
// Reference behaviour: /root/reference/models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299
// (sampling + aggregation; pixel mapping h = y*H - 0.5, in-range test :285-288).
//
// Why.  Round 2's kernel (msda_fwd_pyr.hip) kept level 0 on the vector-memory path: 928 of its
// 1 490 vector-memory instructions per workgroup were level-0 corner gathers, and the kernel sat
// on that path's issue rate (profiles/r02_msda_fwd_pyr.md: 176 us per N=4 call, 0.227 of the HBM
// line).  LDS serves ds_read_b128 at 256 B/clk/CU -- six times the vector-memory path -- so every
// corner row now comes out of LDS and the vector-memory path only carries what is streamed once:
// the window fills (LDS-DMA), the sampling locations / weights and the output.
//
// Decomposition (host plan: msda_pyr2.h).  The image plane is cut into regions; a 384-thread
// workgroup -- TWO per CU, 80 KB of LDS each: while one fills its windows the other gathers, so the
// vector-memory path and the LDS / VALU pipes overlap without software pipelining -- owns one
// (image, region, head): all queries whose reference point lies in the region, in all four levels.
// It works through PHASES that re-use one window buffer: stage the windows of the phase's levels
// by LDS-DMA (`buffer_load_dwordx4 ... lds`, zero-fill outside the image), barrier, every task
// (16 queries of one wave) gathers its samples of those levels, barrier.  The 16 x 128-B output
// rows of a task stay in registers across the phases (kTPW tasks per wave, statically unrolled).
// Window extents come from a per-(head, level) offset ENVELOPE: a DINO encoder's heads each look
// in one direction, so a head's windows are about half the symmetric ones and level 0 fits.
// A sample outside its window is fetched from global memory by a slow path: results never depend
// on the plan.
//
// Lane mapping as in round 2: 4 lanes share a query; lane j works out the geometry of POINT j of
// the level at hand and owns the 16-B pieces j and j + 4 of every 128-B row; geometry travels by
// DPP quad broadcasts.  Which half a quad reads first alternates with bit 2 of its slot: the four
// quads a ds_read_b128 serves together then hit four different 16-bank quarters for the common
// access patterns (consecutive queries -> consecutive or pairwise-equal rows; exhaustive search of
// the 256 assignments in DESIGN.md 4.1: 1.19 LDS cycles per group against 1.45 for round 2's
// slot-parity rule).  The 4-corner blend is v_pk_fma_f32 on register pairs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "datr_hip.h"
#include "msda_tiled.h"
#include "msda_pyr2.h"

// Development only: compile pieces out to see what they cost (results are wrong with a bit set).
// 1 = no window fill, 4 = no LDS gathers, 8 = no loc/attn loads, 16 = no output stores, 32 = no blend FMAs,
// 64 = no level loop at all (launch + prologue only), 128 = no barriers; backward (kDots) only: 256 = no
// grad_loc / grad_attn stores, 512 = no quad sums
#ifndef PYR2_ABLATE
#define PYR2_ABLATE 0
#endif
#ifndef PYR2_OUT_NT
#define PYR2_OUT_NT 1
#endif
#ifndef PYR2_ROW_LOC
#define PYR2_ROW_LOC 1
#endif
#ifndef PYR2_DOTS_DEPTH
#define PYR2_DOTS_DEPTH 1
#endif
#ifndef PYR2_BANDS
#define PYR2_BANDS 1
#endif


#ifdef PYR2_PROBE
// per-phase cycle counters of lane 0 of every wave (development; tools/probes/pyr2_ablate.sh)
__device__ unsigned long long pyr2_phase_cycles[1024][8];   // many sets: one hot address would serialise
__device__ unsigned long long pyr2_wg_span[8192][4];        // per workgroup: realtime start / end (100 MHz), cycles, XCC|CU id
extern "C" void datr_probe_pyr2_wg_spans(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pyr2_wg_span), sizeof(pyr2_wg_span));
}
#define TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
                     ticks_[i] += now_ - tick_; tick_ = now_; } while (0)
extern "C" void datr_probe_pyr2_phase_cycles(unsigned long long *out, int reset) {
    static unsigned long long all[1024][8];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(pyr2_phase_cycles), sizeof(all));
    for (int i = 0; i < 8; ++i) {
        out[i] = 0;
        for (int b = 0; b < 1024; ++b) out[i] += all[b][i];
    }
    if (reset) {
        static unsigned long long z[1024][8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pyr2_phase_cycles), z, sizeof(z));
    }
}
#else
#define TICK(i) do {} while (0)
#endif

namespace {

constexpr unsigned kOutOfRange = 0x80000000u;    // >= num_records of every descriptor built here
constexpr int kRowBytes = 128;                   // D = 32 floats

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
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

// acc (two channel pairs) += w * v with packed FMAs: 2 instructions per 16-B piece
__device__ __forceinline__ void pk_fma4(f4 &acc, float w, const f4 v) {
    f2 lo = {acc.x, acc.y}, hi = {acc.z, acc.w};
    const f2 ww = {w, w};
    lo = __builtin_elementwise_fma(ww, f2{v.x, v.y}, lo);
    hi = __builtin_elementwise_fma(ww, f2{v.z, v.w}, hi);
    acc = f4{lo.x, lo.y, hi.x, hi.y};
}

#define PIN(a, b) asm volatile("" : "+v"(a), "+v"(b) : : "memory")

struct Rows { f4 a[4], b[4]; };      // the lane's two 16-B pieces of the 4 corner rows of a sample

// Window sample whose top-left LDS address lane SRC holds; row_bytes = window width * 128.
template <int SRC>
__device__ __forceinline__ void fetch_lds(Rows &r, const int base, const int chan, const int row_bytes) {
    const int t1 = quad_bcast<SRC>(base) + chan, t2 = t1 ^ 64;
    const int u1 = t1 + row_bytes, u2 = t2 + row_bytes;
    if (PYR2_ABLATE & 4) {
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

// `first` = the piece the lane's t1 address points at (acc of the lane's FIRST read): the caller
// passes (accA, accB) in read order, so no select is needed here.
template <int SRC>
__device__ __forceinline__ void accumulate(f4 &accA, f4 &accB, const Rows &r, const float (&w)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = quad_bcast<SRC>(w[k]);
        if (PYR2_ABLATE & 32) {                  // no blend: one add keeps the dependency
            accA.x += wk * r.a[k].x;
            accB.x += r.b[k].x;
            continue;
        }
        pk_fma4(accA, wk, r.a[k]);
        pk_fma4(accB, wk, r.b[k]);
    }
}

// The four points of one level (point p's geometry sits in lane p of the quad).
template <int kDepth>                 // samples whose corner rows are in flight out of LDS
__device__ __forceinline__ void lds_level(f4 &accA, f4 &accB, const int base, const float (&w)[4],
                                          const int chan, const int row_bytes) {
  if constexpr (kDepth == 1) {
    Rows r;
    fetch_lds<0>(r, base, chan, row_bytes);
    accumulate<0>(accA, accB, r, w);
    PIN(accA, accB);
    fetch_lds<1>(r, base, chan, row_bytes);
    accumulate<1>(accA, accB, r, w);
    PIN(accA, accB);
    fetch_lds<2>(r, base, chan, row_bytes);
    accumulate<2>(accA, accB, r, w);
    PIN(accA, accB);
    fetch_lds<3>(r, base, chan, row_bytes);
    accumulate<3>(accA, accB, r, w);
  } else {
    Rows r0, r1;
    fetch_lds<0>(r0, base, chan, row_bytes);
    fetch_lds<1>(r1, base, chan, row_bytes);
    PIN(accA, accB);
    accumulate<0>(accA, accB, r0, w);
    fetch_lds<2>(r0, base, chan, row_bytes);
    PIN(accA, accB);
    accumulate<1>(accA, accB, r1, w);
    fetch_lds<3>(r1, base, chan, row_bytes);
    PIN(accA, accB);
    accumulate<2>(accA, accB, r0, w);
    PIN(accA, accB);
    accumulate<3>(accA, accB, r1, w);
  }
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

// The workgroup's program.  kDots = false: the forward (out = sum of a * bilinear(value)).
// kDots = true: the value-dependent half of the BACKWARD (cuh:301-403) with the same windows, phases and
// tasks -- lane j dots the four corner rows of ITS sample (point j of the level) with the query's
// grad_out row and turns the four dot products into grad_attn = <grad_out, bilinear(value)> and
// grad_loc = a * {W, H} * <grad_out, d bilinear / d (w, h)>.
// ---- backward (kDots) ------------------------------------------------------------------------------
// sum over the 4 lanes of a quad; every lane ends with the total
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
// Point SRC of the quad's query: the four corner dot products <grad_out, row> (cuh:301-403 sums
// top_grad * v_k over the channels): packed FMAs over the lane's 8 channels, two DPP steps across the
// quad.  Lane SRC -- the lane that holds point SRC's geometry -- keeps them in d.
template <int SRC>
__device__ __forceinline__ void point_dots(float (&d)[4], const f4 goA, const f4 goB, const Rows &r, const int j) {
    const f2 a_lo = {goA.x, goA.y}, a_hi = {goA.z, goA.w}, b_lo = {goB.x, goB.y}, b_hi = {goB.z, goB.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f2 acc = f2{r.a[k].x, r.a[k].y} * a_lo;
        acc = __builtin_elementwise_fma(f2{r.a[k].z, r.a[k].w}, a_hi, acc);
        acc = __builtin_elementwise_fma(f2{r.b[k].x, r.b[k].y}, b_lo, acc);
        acc = __builtin_elementwise_fma(f2{r.b[k].z, r.b[k].w}, b_hi, acc);
        const float s = (PYR2_ABLATE & 512) ? acc.x + acc.y : quad_sum(acc.x + acc.y);
        d[k] = j == SRC ? s : d[k];
    }
}
// The four points of one level out of the LDS window (the forward's lds_level): lane j leaves with
// the four dot products of point j.
template <int kDepth>
__device__ __forceinline__ void lds_level_dots(float (&d)[4], const f4 goA, const f4 goB, const int base,
                                               const int j, const int chan, const int row_bytes) {
    if constexpr (kDepth == 1) {
        Rows r;
        fetch_lds<0>(r, base, chan, row_bytes);
        point_dots<0>(d, goA, goB, r, j);
        fetch_lds<1>(r, base, chan, row_bytes);
        point_dots<1>(d, goA, goB, r, j);
        fetch_lds<2>(r, base, chan, row_bytes);
        point_dots<2>(d, goA, goB, r, j);
        fetch_lds<3>(r, base, chan, row_bytes);
        point_dots<3>(d, goA, goB, r, j);
    } else {
        Rows r0, r1;
        fetch_lds<0>(r0, base, chan, row_bytes);
        fetch_lds<1>(r1, base, chan, row_bytes);
        point_dots<0>(d, goA, goB, r0, j);
        fetch_lds<2>(r0, base, chan, row_bytes);
        point_dots<1>(d, goA, goB, r1, j);
        fetch_lds<3>(r1, base, chan, row_bytes);
        point_dots<2>(d, goA, goB, r0, j);
        point_dots<3>(d, goA, goB, r1, j);
    }
}
// 4 x 4 transpose across the lanes of a quad: lane j enters with v[i] = M[j][i] and leaves with M[i][j]
__device__ __forceinline__ void quad_transpose(float (&v)[4], const int j) {
    const bool b0 = j & 1, b1 = j & 2;
#pragma unroll
    for (int q = 0; q < 4; q += 2) {                 // exchange with lane ^ 1: register pairs (0, 1), (2, 3)
        const float send = b0 ? v[q] : v[q + 1];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
        if (b0) v[q] = recv; else v[q + 1] = recv;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {                    // exchange with lane ^ 2: register pairs (0, 2), (1, 3)
        const float send = b1 ? v[q] : v[q + 2];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
        if (b1) v[q] = recv; else v[q + 2] = recv;
    }
}
// cuh:301-403 for one sample: pa = <grad_out, bilinear(value)>, pw / ph = its derivatives in lw / lh
__device__ __forceinline__ void combine_dots(float &pa, float &pw, float &ph, const float (&d)[4], const Pix &p) {
    const float hh = 1.f - p.lh, hw = 1.f - p.lw;
    pa = hh * hw * d[0] + hh * p.lw * d[1] + p.lh * hw * d[2] + p.lh * p.lw * d[3];
    pw = hh * (d[1] - d[0]) + p.lh * (d[3] - d[2]);
    ph = hw * (d[2] - d[0]) + p.lw * (d[3] - d[1]);
}

// kQueryGrad (backward): the results leave as the gradient of the module's merged query projection --
// rows of M * 48 floats, [M][4 levels][4 points][2] offset gradients (= grad_loc: the division by (W_l, H_l) is folded
// into the projection's weights, datr_amd/msda.py) then [M][16] logit gradients, the softmax backward
// a_k (g_k - sum_j a_j g_j) of /root/reference/models/dino/ops/modules/ms_deform_attn.py:106 applied here -- in place
// of grad_loc / grad_attn and a separate pass that turns them into that row (csrc/msda_prologue.hip).
template <int kTPW, int kCfg, bool kDots, bool kQueryGrad = false>
__device__ __forceinline__ void pyr2_body(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const Pyr2Meta &pm, int S, int M, int nimg, float *__restrict__ out,
    const float *__restrict__ grad_out, float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int kP2Threads = kP2Configs[kCfg].threads, kP2Waves = kP2Threads / 64;
    // two samples' corner rows in flight where the register budget allows (168 VGPRs at 3 x 256 threads)
    // (the backward holds 12 results per task until its last level: one sample in flight at 128 VGPRs)
    constexpr int kDepth = kDots ? (kCfg == 1 ? 2 : PYR2_DOTS_DEPTH) : (kCfg == 1 || kTPW <= 2) ? 2 : 1;
    const int tid = threadIdx.x;
    const int nreg = pm.nRy * pm.nRx;
    const unsigned row_stride = (unsigned)M * kRowBytes;           // bytes between pixels of one head
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PYR2_PROBE
    unsigned long long ticks_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = __builtin_readcyclecounter();
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), cy0_ = tick_;
#endif
    {
    const int item = blockIdx.x;      // one (image, region, head) item per workgroup, head fastest:
                                      // workgroup b runs on XCD b % 8, an XCD's L2 holds one head's slice
#if PYR2_BANDS
    // XCD = item % 8 owns a BAND of (image, region) pairs with all heads (dense lines in its L2)
    // instead of one head everywhere (every 8th 128-B line): item = (j * M + m) * 8 + band.
    // Bands run over the pairs of the whole launch: 4 x 98 regions are 49 pairs per XCD, where
    // per-image bands of ceil(98 / 8) = 13 regions left the eighth XCD half idle.
    const int npair = nimg * nreg;
    const int band = item % 8, m = (item / 8) % M, jj = item / (8 * M);
    const int pair = band * npair / 8 + jj;                 // band b: pairs [b npair / 8, (b + 1) npair / 8)
    if (pair >= (band + 1) * npair / 8) return;
    const int n = pair / nreg, reg = pair - n * nreg;
    const int ry = reg / pm.nRx, rx = reg % pm.nRx;
#else
    const int m = item % M;
    const int reg = (item / M) % nreg;
    const int n = item / (M * nreg);
    const int ry = reg / pm.nRx, rx = reg % pm.nRx;
#endif

    const float *base = value + ((size_t)n * S * M + m) * 32;
    const int records = (S * M - m) * kRowBytes;                   // bytes from `base` to the end of item n
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, records, 0x00020000);

    // ---- everything the workgroup needs from the plan, fetched ONCE as wide scalar loads --------
    // (wave-uniform indices m, ry, rx; the level loops below are fully unrolled, so these stay in
    // SGPRs and no load or wait sits between the gathers)
    struct Lv { int H, W, start, WH, WW, lbase, wy0, wx0; } lv[4];
    int pre[5], qoy[4], qox[4], qrw[4];
    pre[0] = 0;
    {
        const int4v ya = *reinterpret_cast<const int4v *>(pm.yb[ry]), yb_ = *reinterpret_cast<const int4v *>(pm.yb[ry + 1]);
        const int4v xa = *reinterpret_cast<const int4v *>(pm.xb[rx]), xb_ = *reinterpret_cast<const int4v *>(pm.xb[rx + 1]);
        const int2v wyp = *reinterpret_cast<const int2v *>(pm.wy0[m][ry]);
        const int2v wxp = *reinterpret_cast<const int2v *>(pm.wx0[m][rx]);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int4v h = *reinterpret_cast<const int4v *>(&pm.hl[m][l]);
            lv[l].H = pm.H[l]; lv[l].W = pm.W[l]; lv[l].start = pm.start[l];
            lv[l].WH = h.x; lv[l].WW = h.y; lv[l].lbase = h.z * kRowBytes;
            const int wy2 = l < 2 ? wyp.x : wyp.y, wx2 = l < 2 ? wxp.x : wxp.y;
            lv[l].wy0 = (l & 1) ? (wy2 >> 16) : (int)(short)(wy2 & 0xffff);
            lv[l].wx0 = (l & 1) ? (wx2 >> 16) : (int)(short)(wx2 & 0xffff);
            pre[l + 1] = pre[l] + (yb_[l] - ya[l]) * (xb_[l] - xa[l]);
        }
        // the region's queries: level by level, row by row of the region's footprint
        qoy[0] = ya.x; qoy[1] = ya.y; qoy[2] = ya.z; qoy[3] = ya.w;
        qox[0] = xa.x; qox[1] = xa.y; qox[2] = xa.z; qox[3] = xa.w;
        qrw[0] = xb_.x - xa.x; qrw[1] = xb_.y - xa.y; qrw[2] = xb_.z - xa.z; qrw[3] = xb_.w - xa.w;
    }
    const int nq = pre[4];
    const int ntasks = (nq + 15) >> 4;
    const int nph = pm.nph;
    const int4v phm = *reinterpret_cast<const int4v *>(pm.ph_mask);

    // stage the windows of one phase's levels by LDS-DMA: one wave instruction moves 64 x 16 B = 8
    // window pixels straight into LDS (lane i lands at base + 16 i); an out-of-image pixel gets an
    // out-of-range offset and reads zeros
    auto fill_phase = [&](int mask, int l_first) {
        if (PYR2_ABLATE & 1) return;
#pragma unroll
        for (int lf = 0; lf < 4; ++lf) {
            if (lf < l_first || !((mask >> lf) & 1)) continue;
            const int WW = lv[lf].WW, cnt = lv[lf].WH * WW * 8;
            const int wy0 = lv[lf].wy0, wx0 = lv[lf].wx0;
            const int Hl = lv[lf].H, Wl = lv[lf].W, st = lv[lf].start;
            const float inv = 1.0f / (float)WW;
            const int lbase = lv[lf].lbase;
            for (int i0 = wave * 64; i0 < cnt; i0 += kP2Threads) {
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
    };
    // the first phase's windows are requested BEFORE the query decode and the location loads below:
    // the fill is the longest latency of the prologue
    fill_phase(phm.x, 0);

    // Lane roles.  4 lanes share a query; lane j owns the 16-B pieces j and j + 4 of every row and
    // works out the geometry of POINT j.  Quads whose slot has bit 2 set read the upper half first.
    const int slot = lane >> 2, j = lane & 3;
    const int upper_first = (slot >> 2) & 1;
    const int chan = 16 * j + 64 * upper_first;                    // byte offset of the FIRST read

    // sampling locations / attention weights through buffer descriptors with 32-bit offsets
    // (host-checked: N * Lq * M * 16 samples * 8 B < 2^31)
    const __amdgpu_buffer_rsrc_t loc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(loc), 0, nimg * S * M * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t attn_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(attn), 0, nimg * S * M * 64, 0x00020000);

    f4 accA[kTPW], accB[kTPW];                                     // read order: A = first read
    int soff[kTPW];                                                // first sample index of the task's (query, head)
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
        accA[t] = accB[t] = f4{0.f, 0.f, 0.f, 0.f};                 // kDots: the grad_out pieces (below)
        int qi = (wave + t * kP2Waves) * 16 + slot;
        qi = qi < nq ? qi : nq - 1;
        const int lq = (qi >= pre[1]) + (qi >= pre[2]) + (qi >= pre[3]);
        const int li = qi - (lq == 0 ? 0 : lq == 1 ? pre[1] : lq == 2 ? pre[2] : pre[3]);
        const int oy = lq == 0 ? qoy[0] : lq == 1 ? qoy[1] : lq == 2 ? qoy[2] : qoy[3];
        const int ox = lq == 0 ? qox[0] : lq == 1 ? qox[1] : lq == 2 ? qox[2] : qox[3];
        const int rw = lq == 0 ? qrw[0] : lq == 1 ? qrw[1] : lq == 2 ? qrw[2] : qrw[3];
        const int Wq = lq == 0 ? lv[0].W : lq == 1 ? lv[1].W : lq == 2 ? lv[2].W : lv[3].W;
        const int sq = lq == 0 ? lv[0].start : lq == 1 ? lv[1].start : lq == 2 ? lv[2].start : lv[3].start;
        const int r_ = (int)(((float)li + 0.5f) / (float)rw);
        const int q = sq + (oy + r_) * Wq + ox + (li - r_ * rw);
        soff[t] = ((n * S + q) * M + m) * 16;
        if constexpr (kDots) {                                     // the query's grad_out row (this head)
            const float *src = grad_out + (size_t)soff[t] * 2;
            accA[t] = *reinterpret_cast<const f4 *>(src + (chan >> 2));
            accB[t] = *reinterpret_cast<const f4 *>(src + ((chan ^ 64) >> 2));
        }
    }
#ifndef PYR2_RECORDS
#define PYR2_RECORDS 0      // PROBE (round 6): `loc` holds packed per-sample records {pixel (iy << 16 | ix), lh, lw, a}, 16 B
#endif
    struct LocW { f2 xy; float a; int pix; };
    auto load_sample = [&](int so, int l) {
        LocW r;
        r.pix = 0;
#if PYR2_RECORDS
        {
            const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(loc), 0, nimg * S * M * 256, 0x00020000);
            const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (so + 4 * l + j) * 16, 0, 0));
            r.pix = __builtin_bit_cast(int, v.x);
            r.xy = f2{v.z, v.y};          // (lw, lh)
            r.a = v.w;
            return r;
        }
#endif
#if PYR2_ABLATE & 8
        // no loads: the centre of the level's window (always inside it), constant weight
        r.xy = f2{((float)(lv[l].wx0 + lv[l].WW / 2) + 0.5f + 1e-9f * (float)so) / (float)lv[l].W,
                  ((float)(lv[l].wy0 + lv[l].WH / 2) + 0.5f) / (float)lv[l].H};
        r.a = 0.0625f;
#else
        // default cache policy, NOT nt: the four levels' pieces of a query's 128-B location line are
        // read at four different times; with the non-temporal hint every one of them went back to
        // HBM (FETCH_SIZE 283 -> 147 MB per launch, 155 -> 133 us without it)
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(loc_rsrc, (so + 4 * l + j) * 8, 0, 0);
        r.xy = __builtin_bit_cast(f2, v);
        r.a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(attn_rsrc, (so + 4 * l + j) * 4, 0, 0));
#endif
        return r;
    };
    // Memory-level parallelism: a level's locations / weights of ALL of the wave's tasks are
    // requested one level ahead (4 x 768 B per wave, ~36 KB per CU in flight).
    LocW cur[kTPW], nxt[kTPW];
    float res_a[kDots ? kTPW : 1][4], res_w[kDots ? kTPW : 1][4], res_h[kDots ? kTPW : 1][4];   // kDots: [task][level]
    auto load_level = [&](LocW (&dst)[kTPW], int l) {
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
            if (wave + t * kP2Waves < ntasks) dst[t] = load_sample(soff[t], l);
    };
    // kRowLoc (the backward at two tasks per wave: 137 -> 131 us; the forward's 128 registers leave no room for
    // the 12 values per task -- 95 -> 99 us --, three tasks per wave spill).  Whole rows instead: lane j loads LEVEL j's four points of its query -- 32 B of locations, 16 B of
    // weights; a quad reads the (query, head)'s 128-B location row and 64-B weight row once, as whole lines
    // -- and a 4 x 4 transpose inside the quad leaves it with POINT j of the four levels.  3 loads per task
    // in place of 8 that each used 32 / 16 B of sixteen different lines.
    constexpr bool kRowLoc = PYR2_ROW_LOC && kDots && kTPW <= 2;
    float lx_[kRowLoc ? kTPW : 1][4], ly_[kRowLoc ? kTPW : 1][4], la_[kRowLoc ? kTPW : 1][4];
    if constexpr (kRowLoc) {
#pragma unroll
    for (int t = 0; t < kTPW; ++t)
        if (wave + t * kP2Waves < ntasks) {
            const f4 l0 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(loc_rsrc, (soff[t] + 4 * j) * 8, 0, 0));
            const f4 l1 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(loc_rsrc, (soff[t] + 4 * j + 2) * 8, 0, 0));
            const f4 a4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(attn_rsrc, (soff[t] + 4 * j) * 4, 0, 0));
            lx_[t][0] = l0.x; ly_[t][0] = l0.y; lx_[t][1] = l0.z; ly_[t][1] = l0.w;
            lx_[t][2] = l1.x; ly_[t][2] = l1.y; lx_[t][3] = l1.z; ly_[t][3] = l1.w;
            la_[t][0] = a4.x; la_[t][1] = a4.y; la_[t][2] = a4.z; la_[t][3] = a4.w;
            quad_transpose(lx_[t], j);
            quad_transpose(ly_[t], j);
            quad_transpose(la_[t], j);
        }
    } else {
        load_level(nxt, 0);
    }
    TICK(0);                                             // prologue: plan loads, query decode

    int next_phase = 0;
#pragma unroll
    for (int l = 0; l < ((PYR2_ABLATE & 64) ? 0 : 4); ++l) {
        // ---- a new phase starts with level l: stage the windows of its levels by LDS-DMA --------
        // One wave instruction moves 64 x 16 B = 8 window pixels straight into LDS; lane i lands at
        // base + 16 i.  An out-of-image pixel gets an out-of-range offset: the load returns zeros.
        const int mask = next_phase == 0 ? phm.x : next_phase == 1 ? phm.y : next_phase == 2 ? phm.z : phm.w;
        if (next_phase < nph && (mask & ((1 << l) - 1)) == 0 && ((mask >> l) & 1)) {
            if (next_phase > 0 && !(PYR2_ABLATE & 128)) __syncthreads();   // everyone is done with the old windows
            TICK(1);                                     // waiting for the workgroup before a re-fill
            ++next_phase;
            if (next_phase > 1) fill_phase(mask, l);       // phase 0 was issued in the prologue
            TICK(2);                                     // fill issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TICK(3);                                     // own fill pieces (and pending locations) landing
            if (!(PYR2_ABLATE & 128)) __syncthreads();
            TICK(4);                                     // waiting for the workgroup's pieces
        }

        const int Hl = lv[l].H, Wl = lv[l].W;
        const int WW = lv[l].WW, WH = lv[l].WH;
        const int wy0 = lv[l].wy0, wx0 = lv[l].wx0;
        const int lbase = lv[l].lbase;
        const int row_bytes = WW * kRowBytes;
#pragma unroll
        for (int t = 0; t < kTPW; ++t) {
            if constexpr (kRowLoc) {
                cur[t].xy = f2{lx_[t][l], ly_[t][l]};
                cur[t].a = la_[t][l];
            } else {
                cur[t] = nxt[t];
            }
        }
#ifdef PYR2_PROBE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TICK(5);                                         // locations of this level landing
        if constexpr (!kRowLoc)
            if (l + 1 < 4) load_level(nxt, l + 1);
#pragma unroll
        for (int t = 0; t < kTPW; ++t) {
            const int ti = wave + t * kP2Waves;
            if (ti >= ntasks) break;
#if PYR2_RECORDS
            Pix p;
            p.inside = cur[t].pix != 0x7fffffff;
            p.iy = p.inside ? (cur[t].pix >> 16) : 0;
            p.ix = p.inside ? (int)(short)(cur[t].pix & 0xffff) : 0;
            p.lh = cur[t].xy.y;
            p.lw = cur[t].xy.x;
#else
            const Pix p = locate(cur[t].xy.x, cur[t].xy.y, Hl, Wl);
#endif
            const int wy = p.iy - wy0, wx = p.ix - wx0;
            const bool inwin = (unsigned)wy <= (unsigned)(WH - 2) && (unsigned)wx <= (unsigned)(WW - 2);
            const bool use = p.inside && inwin;
            const bool miss = p.inside && !inwin;
            const int wb = lbase + ((inwin ? wy : 0) * WW + (inwin ? wx : 0)) * kRowBytes;
            float w[4];
            if constexpr (kDots) {
                float d[4] = {0.f, 0.f, 0.f, 0.f};                 // the corner dot products of THIS lane's sample
                lds_level_dots<kDepth>(d, accA[t], accB[t], wb, j, chan, row_bytes);
                if (__builtin_amdgcn_ballot_w64(miss) != 0) {      // rare: rows from global memory
                    const int stl = lv[l].start;
                    const bool top = p.iy >= 0, bot = p.iy + 1 <= Hl - 1, lef = p.ix >= 0, rig = p.ix + 1 <= Wl - 1;
                    const unsigned pix = (unsigned)(stl + p.iy * Wl + p.ix) * row_stride;
                    int g[4];
                    g[0] = (int)((miss && top && lef) ? pix : kOutOfRange);
                    g[1] = (int)((miss && top && rig) ? pix + row_stride : kOutOfRange);
                    g[2] = (int)((miss && bot && lef) ? pix + (unsigned)Wl * row_stride : kOutOfRange);
                    g[3] = (int)((miss && bot && rig) ? pix + (unsigned)(Wl + 1) * row_stride : kOutOfRange);
                    auto one = [&](int flag, auto src) {
                        constexpr int SRC = decltype(src)::value;
                        if (flag) {                                // the quads whose sample SRC missed
                            Rows r;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned o = (unsigned)quad_bcast<SRC>(g[k]);
                                r.a[k] = load_row4(rsrc, o + (unsigned)chan);
                                r.b[k] = load_row4(rsrc, o + (unsigned)(chan ^ 64));
                            }
                            point_dots<SRC>(d, accA[t], accB[t], r, j);
                        }
                    };
                    const int mi = miss ? 1 : 0;
                    one(quad_bcast<0>(mi), std::integral_constant<int, 0>{});
                    one(quad_bcast<1>(mi), std::integral_constant<int, 1>{});
                    one(quad_bcast<2>(mi), std::integral_constant<int, 2>{});
                    one(quad_bcast<3>(mi), std::integral_constant<int, 3>{});
                }
                float pa = 0.f, pw = 0.f, ph = 0.f;
                if (p.inside) combine_dots(pa, pw, ph, d, p);
                // kept until all four levels are done: written per level, the 16-B / 32-B pieces of a
                // (query, head)'s 64-B / 128-B gradient rows cost 72 of the kernel's 194 us
                res_a[t][l] = pa;
                res_w[t][l] = pw * cur[t].a * (float)Wl;
                res_h[t][l] = ph * cur[t].a * (float)Hl;
                PIN(accA[t], accB[t]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            } else {
                corner_weights(w, p, use ? cur[t].a : 0.f);
                lds_level<kDepth>(accA[t], accB[t], wb, w, chan, row_bytes);
            }

            // ---- slow path (rare): samples outside their window come from global memory ----------
            if (__builtin_amdgcn_ballot_w64(miss) != 0) {
                const int stl = lv[l].start;
                const bool top = p.iy >= 0, bot = p.iy + 1 <= Hl - 1, lef = p.ix >= 0, rig = p.ix + 1 <= Wl - 1;
                const unsigned pix = (unsigned)(stl + p.iy * Wl + p.ix) * row_stride;
                int g[4];
                g[0] = (int)((miss && top && lef) ? pix : kOutOfRange);
                g[1] = (int)((miss && top && rig) ? pix + row_stride : kOutOfRange);
                g[2] = (int)((miss && bot && lef) ? pix + (unsigned)Wl * row_stride : kOutOfRange);
                g[3] = (int)((miss && bot && rig) ? pix + (unsigned)(Wl + 1) * row_stride : kOutOfRange);
                float wm[4];
                corner_weights(wm, p, miss ? cur[t].a : 0.f);
                auto one = [&](int flag, auto src) {
                    constexpr int SRC = decltype(src)::value;
                    // only the quads whose sample missed execute this (divergent branch: the
                    // vector-memory path is charged per active lane)
                    if (flag) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned o = (unsigned)quad_bcast<SRC>(g[k]);
                            const float wk = quad_bcast<SRC>(wm[k]);
                            const f4 ra = load_row4(rsrc, o + (unsigned)chan);
                            const f4 rb = load_row4(rsrc, o + (unsigned)(chan ^ 64));
                            pk_fma4(accA[t], wk, ra);
                            pk_fma4(accB[t], wk, rb);
                        }
                    }
                };
                const int mi = miss ? 1 : 0;
                one(quad_bcast<0>(mi), std::integral_constant<int, 0>{});
                one(quad_bcast<1>(mi), std::integral_constant<int, 1>{});
                one(quad_bcast<2>(mi), std::integral_constant<int, 2>{});
                one(quad_bcast<3>(mi), std::integral_constant<int, 3>{});
            }
            // keep the tasks apart: hoisting the next task's gathers above this one's blend
            // overflows the register file
            PIN(accA[t], accB[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        TICK(6);                                         // geometry + gathers + blend of the level
    }

    // ---- output rows ---------------------------------------------------------------------------
    if constexpr (kDots) {
        // lane j holds point j of levels 0..3; a 4 x 4 transpose inside the quad leaves it with the four
        // points of LEVEL j: one 16-B store of grad_attn, two of grad_loc; a quad writes whole rows
#pragma unroll
        for (int t = 0; t < kTPW; ++t) {
            const int ti = wave + t * kP2Waves;
            if (ti >= ntasks) break;
            quad_transpose(res_a[t], j);
            quad_transpose(res_w[t], j);
            quad_transpose(res_h[t], j);
            if constexpr (kQueryGrad) {
                // lane j now holds LEVEL j's four points: their weights again, as one 16-B load (the gather's
                // registers are free here; the line was read a few microseconds ago)
                const f4 a4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(attn_rsrc, (soff[t] + 4 * j) * 4, 0, 0));
                const float dot = quad_sum((a4.x * res_a[t][0] + a4.y * res_a[t][1]) +
                                           (a4.z * res_a[t][2] + a4.w * res_a[t][3]));
                res_a[t][0] = a4.x * (res_a[t][0] - dot);
                res_a[t][1] = a4.y * (res_a[t][1] - dot);
                res_a[t][2] = a4.z * (res_a[t][2] - dot);
                res_a[t][3] = a4.w * (res_a[t][3] - dot);
                if (ti * 16 + slot < nq) {
                    const int row = soff[t] / (16 * M), mh = (soff[t] >> 4) - row * M;
                    float *qrow = grad_loc + (size_t)row * (M * 48);
                    f4 *gl = reinterpret_cast<f4 *>(qrow + (mh * 4 + j) * 8);
                    __builtin_nontemporal_store(f4{res_a[t][0], res_a[t][1], res_a[t][2], res_a[t][3]},
                                                reinterpret_cast<f4 *>(qrow + M * 32 + (mh * 4 + j) * 4));
                    __builtin_nontemporal_store(f4{res_w[t][0], res_h[t][0], res_w[t][1], res_h[t][1]}, gl);
                    __builtin_nontemporal_store(f4{res_w[t][2], res_h[t][2], res_w[t][3], res_h[t][3]}, gl + 1);
                }
                __builtin_amdgcn_sched_barrier(0);       // one task at a time: the three-task variant has no registers to spare
                continue;
            }
            if (ti * 16 + slot < nq && (!(PYR2_ABLATE & 256) || res_a[t][0] == 123.456f)) {
                const int idx = soff[t] + 4 * j;
                f4 *gl = reinterpret_cast<f4 *>(grad_loc + (size_t)idx * 2);
#if PYR2_OUT_NT
                __builtin_nontemporal_store(f4{res_a[t][0], res_a[t][1], res_a[t][2], res_a[t][3]},
                                            reinterpret_cast<f4 *>(grad_attn + idx));
                __builtin_nontemporal_store(f4{res_w[t][0], res_h[t][0], res_w[t][1], res_h[t][1]}, gl);
                __builtin_nontemporal_store(f4{res_w[t][2], res_h[t][2], res_w[t][3], res_h[t][3]}, gl + 1);
#else
                *reinterpret_cast<f4 *>(grad_attn + idx) = f4{res_a[t][0], res_a[t][1], res_a[t][2], res_a[t][3]};
                gl[0] = f4{res_w[t][0], res_h[t][0], res_w[t][1], res_h[t][1]};
                gl[1] = f4{res_w[t][2], res_h[t][2], res_w[t][3], res_h[t][3]};
#endif
            }
        }
    }
    if constexpr (!kDots)
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
        const int ti = wave + t * kP2Waves;
        if (ti >= ntasks) break;
        const bool live = ti * 16 + slot < nq;
        if (live && (!(PYR2_ABLATE & 16) || accA[t].x == 123.456f)) {
            float *dst = out + (size_t)soff[t] * 2;                // 16 samples <-> 32 output floats
#if PYR2_OUT_NT
            __builtin_nontemporal_store(accA[t], reinterpret_cast<f4 *>(dst + (chan >> 2)));
            __builtin_nontemporal_store(accB[t], reinterpret_cast<f4 *>(dst + ((chan ^ 64) >> 2)));
#else
            *reinterpret_cast<f4 *>(dst + (chan >> 2)) = accA[t];
            *reinterpret_cast<f4 *>(dst + ((chan ^ 64) >> 2)) = accB[t];
#endif
        }
    }
    }
#ifdef PYR2_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TICK(7);                                             // output stores
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&pyr2_phase_cycles[(blockIdx.x * kP2Waves + wave) & 1023][i], ticks_[i]);
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pyr2_wg_span[blockIdx.x][0] = rt0_;
        pyr2_wg_span[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        pyr2_wg_span[blockIdx.x][2] = __builtin_readcyclecounter() - cy0_;
        pyr2_wg_span[blockIdx.x][3] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
#endif
}

template <int kTPW, int kCfg>
__global__ __launch_bounds__(kP2Configs[kCfg].threads, kP2Configs[kCfg].wgs_per_cu * kP2Configs[kCfg].threads / 256)
void msda_fwd_pyr2_d32(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const Pyr2Meta pm, int S, int M, int nimg, float *__restrict__ out)
{
    pyr2_body<kTPW, kCfg, false>(value, loc, attn, pm, S, M, nimg, out, nullptr, nullptr, nullptr);
}

// grad_loc / grad_attn of the encoder calls (the value-dependent half of the backward; grad_value is
// msda_bwd_pyr.hip's sorted scatter, which needs no value rows)
template <int kTPW, int kCfg, bool kQueryGrad>
__global__ __launch_bounds__(kP2Configs[kCfg].threads, kP2Configs[kCfg].wgs_per_cu * kP2Configs[kCfg].threads / 256)
void msda_bwd_dots_pyr2_d32(
    const float *__restrict__ grad_out, const float *__restrict__ value, const float *__restrict__ loc,
    const float *__restrict__ attn, const Pyr2Meta pm, int S, int M, int nimg,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    pyr2_body<kTPW, kCfg, true, kQueryGrad>(value, loc, attn, pm, S, M, nimg, nullptr, grad_out, grad_loc, grad_attn);
}

template <int kTPW, int kCfg, bool kQueryGrad = false>
int launch_dots(const float *grad_out, const float *value, const float *loc, const float *attn,
                const Pyr2Meta &pm, int64_t N, int64_t S, int64_t M, float *grad_loc, float *grad_attn,
                hipStream_t stream) {
    constexpr Pyr2Config cfg = kP2Configs[kCfg];
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(msda_bwd_dots_pyr2_d32<kTPW, kCfg, kQueryGrad>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    p2_lds_bytes(cfg)) == hipSuccess;
    if (!attr_ok) return DATR_EUNSUPPORTED;
#if PYR2_BANDS
    const long blocks = ((long)N * pm.nRy * pm.nRx + 7) / 8 * 8 * M;
#else
    const long blocks = (long)N * pm.nRy * pm.nRx * M;
#endif
    if (blocks <= 0 || blocks >= (1L << 31)) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL((msda_bwd_dots_pyr2_d32<kTPW, kCfg, kQueryGrad>), dim3((unsigned)blocks), dim3(cfg.threads),
                       (size_t)p2_window_rows(cfg) * kRowBytes, stream, grad_out, value, loc, attn, pm, (int)S,
                       (int)M, (int)N, grad_loc, grad_attn);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

template <int kTPW, int kCfg>
int launch(const float *value, const float *loc, const float *attn, const Pyr2Meta &pm, int64_t N,
           int64_t S, int64_t M, float *out, hipStream_t stream) {
    constexpr Pyr2Config cfg = kP2Configs[kCfg];
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_pyr2_d32<kTPW, kCfg>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    p2_lds_bytes(cfg)) == hipSuccess;
    if (!attr_ok) return DATR_EUNSUPPORTED;
#if PYR2_BANDS
    const long blocks = ((long)N * pm.nRy * pm.nRx + 7) / 8 * 8 * M;
#else
    const long blocks = (long)N * pm.nRy * pm.nRx * M;
#endif
    if (blocks <= 0 || blocks >= (1L << 31)) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL((msda_fwd_pyr2_d32<kTPW, kCfg>), dim3((unsigned)blocks), dim3(cfg.threads),
                       (size_t)p2_window_rows(cfg) * kRowBytes, stream, value, loc, attn, pm, (int)S,
                       (int)M, (int)N, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

template <int kCfg>
int launch_tpw(const float *value, const float *loc, const float *attn, const Pyr2Meta &pm, int64_t N,
               int64_t S, int64_t M, float *out, hipStream_t st) {
    switch (pm.tpw) {
        case 1: case 2: return launch<2, kCfg>(value, loc, attn, pm, N, S, M, out, st);
        case 3: return launch<3, kCfg>(value, loc, attn, pm, N, S, M, out, st);
        default: return DATR_EUNSUPPORTED;
    }
}

}  // namespace

// Plan only (no launch): what the kernel would do for this geometry / envelope.
// info[0..7] = {covered, nRy, nRx, phases, tasks per wave, workgroups per image, fill KiB per
// workgroup (head 0), largest window rows of a phase (head 0)}.
DATR_INTERNAL int datr_internal_msda_fwd_pyr2_plan(const int64_t *shapes_host, const int64_t *level_start_host,
                                                int64_t S, int64_t M, const float *envelope_host,
                                                Pyr2Meta *pm_out, int32_t *info) {
    Pyr2Envelope env;
    if (envelope_host) memcpy(&env, envelope_host, sizeof(env));
    else p2_symmetric_envelope(env, 4.5f);
    for (int m_ = 0; m_ < kP2Heads; ++m_)
        for (int l = 0; l < 4; ++l)
            for (int k = 0; k < 4; k += 2) {
                float &lo = env.v[m_][l][k], &hi = env.v[m_][l][k + 1];
                if (!(lo == lo) || !(hi == hi) || lo > hi) { lo = -4.5f; hi = 4.5f; }
                // Wider than the symmetric default does not pay in the forward: the windows grow into
                // extra phases while a sample beyond them only costs a global gather (offsets ~ N(0, 2.5 px):
                // 199 us with the measured +-6.4 px envelope, 153 us with +-4.5; tools/probes/bwd_halo.sh)
                static const float clip = getenv("DATR_MSDA_PYR2_ENV_CLIP") ? (float)atof(getenv("DATR_MSDA_PYR2_ENV_CLIP")) : 4.75f;
                lo = std::min(std::max(lo, -clip), clip);
                hi = std::max(std::min(hi, clip), -clip);
            }
    // The grid search costs ~1 ms of host time: plans are cached (geometry + envelope -> plan).
    // The cache is the library's only mutable state; it is guarded and holds plain data.
    struct Entry { int64_t sh[8], S, M; Pyr2Envelope env; Pyr2Meta pm; bool ok; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    Pyr2Meta pm;
    bool ok = false, hit = false;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry &e : cache)
            if (e.S == S && e.M == M && memcmp(e.sh, shapes_host, sizeof(e.sh)) == 0 &&
                memcmp(&e.env, &env, sizeof(env)) == 0) {
                pm = e.pm; ok = e.ok; hit = true;
                break;
            }
    }
    if (!hit) {
        memset(&pm, 0, sizeof(pm));
        static const char *force = getenv("DATR_MSDA_PYR2_REGIONS");
        static const int force_cfg = (getenv("DATR_MSDA_PYR2_CONFIG") && *getenv("DATR_MSDA_PYR2_CONFIG"))
                                         ? atoi(getenv("DATR_MSDA_PYR2_CONFIG")) : -1;
        ok = build_pyr2_meta(pm, shapes_host, level_start_host, S, (int)M, env, force, force_cfg);
        Entry e;
        memcpy(e.sh, shapes_host, sizeof(e.sh));
        e.S = S; e.M = M; e.env = env; e.pm = pm; e.ok = ok;
        std::lock_guard<std::mutex> lock(mu);
        if (cache.size() >= 64) cache.erase(cache.begin());
        cache.push_back(e);
    }
    if (info) {
        memset(info, 0, 13 * sizeof(int32_t));
        if (ok) {
            int fill = 0, big = 0;
            for (int p = 0; p < pm.nph; ++p) {
                int r = 0;
                for (int l = 0; l < 4; ++l)
                    if (pm.ph_mask[p] >> l & 1) r += (pm.hl[0][l].WH * pm.hl[0][l].WW + 7) & ~7;
                fill += r;
                big = std::max(big, r);
            }
            info[0] = 1; info[1] = pm.nRy; info[2] = pm.nRx; info[3] = pm.nph; info[4] = pm.tpw;
            info[5] = pm.nRy * pm.nRx * (int)M; info[6] = fill / 8; info[7] = big;
            info[12 - 8 + 8] = pm.config;        // [12]: launch configuration (index into kP2Configs)
        }
    }
    if (ok && pm_out) *pm_out = pm;
    return ok ? DATR_OK : DATR_EUNSUPPORTED;
}

// Internal entry (msda.hip dispatches here): DATR_EUNSUPPORTED when the shape is not covered.
extern "C" int datr_internal_msda_fwd_pyr2_d32(
    const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, const float *envelope_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream)
{
    if (D != 32 || L != 4 || P != 4 || Lq != S || M < 1 || M > kP2Heads || N < 1) return DATR_EUNSUPPORTED;
    Pyr2Meta pm;
    const int rc = datr_internal_msda_fwd_pyr2_plan(shapes_host, level_start_host, S, M, envelope_host, &pm, nullptr);
    if (rc != DATR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    return pm.config == 1 ? launch_tpw<1>(value, loc, attn, pm, N, S, M, out, st)
                          : launch_tpw<0>(value, loc, attn, pm, N, S, M, out, st);
}

// The value-dependent half of the encoder backward: grad_loc / grad_attn (every element written).
// DATR_EUNSUPPORTED when the forward's plan does not cover the shape (the caller then lets
// msda_bwd_pyr.hip compute them with its own gathers).
// query_grad != 0: `grad_loc` is the [N * Lq, M * 48] gradient of the merged query projection instead (see
// pyr2_body; grad_attn unused).
DATR_INTERNAL int datr_internal_msda_bwd_dots_pyr2_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, const float *envelope_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_loc, float *grad_attn, void *stream, int query_grad)
{
    if (D != 32 || L != 4 || P != 4 || Lq != S || M < 1 || M > kP2Heads || N < 1) return DATR_EUNSUPPORTED;
    if (N * S * M * 128 >= (1LL << 31)) return DATR_EUNSUPPORTED;
    Pyr2Meta pm;
    const int rc = datr_internal_msda_fwd_pyr2_plan(shapes_host, level_start_host, S, M, envelope_host, &pm, nullptr);
    if (rc != DATR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const auto go = [&](auto cfg) -> int {
        constexpr int kCfg = decltype(cfg)::value;
        switch (pm.tpw) {
            case 1: case 2:
                if (query_grad)
                    return launch_dots<2, kCfg, true>(grad_out, value, loc, attn, pm, N, S, M, grad_loc, grad_attn, st);
                return launch_dots<2, kCfg>(grad_out, value, loc, attn, pm, N, S, M, grad_loc, grad_attn, st);
            case 3:
                if (query_grad)
                    return launch_dots<3, kCfg, true>(grad_out, value, loc, attn, pm, N, S, M, grad_loc, grad_attn, st);
                return launch_dots<3, kCfg>(grad_out, value, loc, attn, pm, N, S, M, grad_loc, grad_attn, st);
            default: return DATR_EUNSUPPORTED;
        }
    };
    return pm.config == 1 ? go(std::integral_constant<int, 1>{}) : go(std::integral_constant<int, 0>{});
}
