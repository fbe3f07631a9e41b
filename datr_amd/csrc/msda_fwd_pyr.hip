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
// Decomposition.  The image plane is cut into nRy x nRx regions (about 12 x 28 level-0 pixels).
// A 1024-thread workgroup owns one (image, region, head): ALL queries whose reference point lies
// in the region -- the region's pixels of level 0 and the matching (4x, 16x, 64x fewer) pixels of
// levels 1-3.  Queries sample around their reference point in every level, so the workgroup
// stages, for levels 1, 2, 3, the window of value rows its queries can reach (region footprint
// + halo; ~450 + 270 + 195 rows = 117 KB) in LDS once, and every sample of those three levels
// -- 3/4 of all gathered bytes -- is a ds_read_b128.  Level 0 (whose window would not fit beside
// the others) is gathered with buffer loads as in the row kernel: 1/4 of the rows on the
// vector-memory path, running concurrently with the LDS gathers of the same wave.
// Out-of-image window pixels are zero-filled, so border corners need no special casing; a sample
// that lands outside its window (learned offsets larger than the halo) is fetched from global
// memory by a slow path, so results never depend on the window heuristic.
//
// Lane mapping (no LDS staging of per-sample geometry at all).  4 lanes share a query ("quad"),
// 16 queries per wave.  Lane j of the quad (a) computes the geometry of the 4 points of LEVEL j
// -- 4 corner addresses + 4 weights per point, held in registers -- and (b) owns channels
// 8j..8j+7 of the output row.  In the gather loop the geometry travels inside the quad with DPP
// quad_perm broadcasts (weights as DPP operands of the FMAs, addresses through one
// v_add_u32_dpp that also adds the lane's channel offset).  Even and odd quads read their two
// 16-B pieces of a row in opposite order, so the 16-lane groups a ds_read_b128 is served in see
// four different 16-bank quarters whenever their rows differ in parity (MI355X_MICROARCH.md, LDS).
//
// blockIdx -> (image, region, head) with the head fastest: workgroup b runs on XCD b % 8, so an
// XCD's 4 MiB L2 holds one head's slice of the pyramid (2.8 MB at 1333x800) -- affinity only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxR = 16;                        // regions per axis
constexpr unsigned kOutOfRange = 0x80000000u;    // >= num_records of every descriptor built here
constexpr int kRowBytes = 128;                   // D = 32 floats
constexpr int kMaxLds = 156 * 1024;

struct PyrMeta {
    int H[4], W[4], start[4];
    int nRy, nRx;
    int WH[4], WW[4], lds_base[4];               // windows of levels 1..3 ([0] unused), bytes
    int lds_bytes;
    short yb[4][kMaxR + 1], xb[4][kMaxR + 1];    // query rows / cols of level l in region i: [b[i], b[i+1])
    short wy0[4][kMaxR], wx0[4][kMaxR];          // window origin (may be negative: zero apron)
};

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4 lds_f4;

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

// Corner rows of point P of the level whose geometry lane SRC of the quad holds, the lane's two
// 16-B pieces of each.  GLOBAL: the "addresses" are byte offsets for zero-filling buffer loads;
// otherwise LDS byte addresses.
template <int SRC, bool GLOBAL>
__device__ __forceinline__ void fetch_point(f4 (&va)[4], f4 (&vb)[4], const int (&addr)[4],
                                            const int chan, const char *lds,
                                            __amdgpu_buffer_rsrc_t rsrc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = quad_bcast<SRC>(addr[k]) + chan;
        if (GLOBAL) {
            va[k] = load_row4(rsrc, (unsigned)a);
            vb[k] = load_row4(rsrc, (unsigned)(a ^ 16));
        } else {
            // `a` IS the LDS address (the dynamic LDS block starts at 0 of the workgroup's
            // allocation: no static LDS in this kernel)
            va[k] = *reinterpret_cast<const lds_f4 *>((unsigned)a);
            vb[k] = *reinterpret_cast<const lds_f4 *>((unsigned)(a ^ 16));
        }
    }
}

template <int SRC>
__device__ __forceinline__ void accumulate_point(f4 &acc0, f4 &acc1, const f4 (&va)[4],
                                                 const f4 (&vb)[4], const float (&wgt)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float w = quad_bcast<SRC>(wgt[k]);
        fma4(acc0, w, va[k]);
        fma4(acc1, w, vb[k]);
    }
}

// All four points of one LDS level, two points in flight (the fences stop the compiler from
// issuing all 32 reads up front, which does not fit the register file beside the level-0 loads).
template <int SRC>
__device__ __forceinline__ void lds_level(f4 &acc0, f4 &acc1, const int (&addr)[4][4],
                                          const float (&wgt)[4][4], const int chan, const char *lds,
                                          __amdgpu_buffer_rsrc_t rsrc) {
    f4 va[4], vb[4], wa[4], wb[4];
    fetch_point<SRC, false>(va, vb, addr[0], chan, lds, rsrc);
    fetch_point<SRC, false>(wa, wb, addr[1], chan, lds, rsrc);
    PIN(acc0, acc1);
    accumulate_point<SRC>(acc0, acc1, va, vb, wgt[0]);
    fetch_point<SRC, false>(va, vb, addr[2], chan, lds, rsrc);
    PIN(acc0, acc1);
    accumulate_point<SRC>(acc0, acc1, wa, wb, wgt[1]);
    fetch_point<SRC, false>(wa, wb, addr[3], chan, lds, rsrc);
    PIN(acc0, acc1);
    accumulate_point<SRC>(acc0, acc1, va, vb, wgt[2]);
    PIN(acc0, acc1);
    accumulate_point<SRC>(acc0, acc1, wa, wb, wgt[3]);
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

    const float *base = value + ((size_t)n * S * M + m) * 32;
    const int records = (S * M - m) * kRowBytes;                   // bytes from `base` to the end of item n
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, records, 0x00020000);

    // ---- stage the windows of levels 1..3 (out-of-image pixels read as zeros) ---------------
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        const int WW = pm.WW[l], cnt = pm.WH[l] * WW * 8;
        const int wy0 = pm.wy0[l][ry], wx0 = pm.wx0[l][rx];
        const int Hl = pm.H[l], Wl = pm.W[l], st = pm.start[l];
        const float inv = 1.0f / (float)WW;
        char *dst = lds + pm.lds_base[l];
        for (int i = tid; i < cnt; i += kThreads) {
            const int pix = i >> 3, chunk = i & 7;
            const int wr = (int)(((float)pix + 0.5f) * inv);
            const int wc = pix - wr * WW;
            const int y = wy0 + wr, x = wx0 + wc;
            const bool in = (unsigned)y < (unsigned)Hl && (unsigned)x < (unsigned)Wl;
            const unsigned off = in ? (unsigned)(st + y * Wl + x) * row_stride + (unsigned)chunk * 16u
                                    : kOutOfRange;
            *reinterpret_cast<f4 *>(dst + i * 16) = load_row4(rsrc, off);
        }
    }

    // ---- the region's queries, level by level ------------------------------------------------
    int qy0[4], qx0[4], qw[4], pre[5];
    pre[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        qy0[l] = pm.yb[l][ry];
        qx0[l] = pm.xb[l][rx];
        qw[l] = pm.xb[l][rx + 1] - qx0[l];
        pre[l + 1] = pre[l] + (pm.yb[l][ry + 1] - qy0[l]) * qw[l];
    }
    const int nq = pre[4];
    const int ntasks = (nq + 15) >> 4;

    const int lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 2, j = lane & 3;
    // per-lane constants: lane j works out the geometry of level j
    const int Hj = j == 0 ? pm.H[0] : j == 1 ? pm.H[1] : j == 2 ? pm.H[2] : pm.H[3];
    const int Wj = j == 0 ? pm.W[0] : j == 1 ? pm.W[1] : j == 2 ? pm.W[2] : pm.W[3];
    const int wyj = j == 1 ? pm.wy0[1][ry] : j == 2 ? pm.wy0[2][ry] : pm.wy0[3][ry];
    const int wxj = j == 1 ? pm.wx0[1][rx] : j == 2 ? pm.wx0[2][rx] : pm.wx0[3][rx];
    const int WWj = j == 1 ? pm.WW[1] : j == 2 ? pm.WW[2] : pm.WW[3];
    const int WHj = j == 1 ? pm.WH[1] : j == 2 ? pm.WH[2] : pm.WH[3];
    const int ldsj = j == 1 ? pm.lds_base[1] : j == 2 ? pm.lds_base[2] : pm.lds_base[3];
    const float Hf = (float)Hj, Wf = (float)Wj;
    const int start0 = pm.start[0];
    // the lane's two 16-B pieces of a 128-B row: 32j + {0,16}; odd quads take them in the
    // opposite order (bank spreading, see the header); `chan ^ 16` is the second piece
    const int chan = 32 * j + 16 * (slot & 1);
    const size_t Lq = (size_t)S;

    __syncthreads();

    for (int t = wave; t < ntasks; t += kWaves) {
        const int qi_raw = t * 16 + slot;
        const bool live = qi_raw < nq;
        const int qi = live ? qi_raw : nq - 1;
        const int lq = (qi >= pre[1]) + (qi >= pre[2]) + (qi >= pre[3]);
        const int li = qi - (lq == 0 ? 0 : lq == 1 ? pre[1] : lq == 2 ? pre[2] : pre[3]);
        const int rw = lq == 0 ? qw[0] : lq == 1 ? qw[1] : lq == 2 ? qw[2] : qw[3];
        const int oy = lq == 0 ? qy0[0] : lq == 1 ? qy0[1] : lq == 2 ? qy0[2] : qy0[3];
        const int ox = lq == 0 ? qx0[0] : lq == 1 ? qx0[1] : lq == 2 ? qx0[2] : qx0[3];
        const int qW = lq == 0 ? pm.W[0] : lq == 1 ? pm.W[1] : lq == 2 ? pm.W[2] : pm.W[3];
        const int qs = lq == 0 ? pm.start[0] : lq == 1 ? pm.start[1] : lq == 2 ? pm.start[2] : pm.start[3];
        const int r_ = (int)(((float)li + 0.5f) / (float)rw);
        const int c_ = li - r_ * rw;
        const int q = qs + (oy + r_) * qW + ox + c_;

        // ---- lane j: locations and weights of level j's four points ---------------------------
        const size_t qm = ((size_t)n * Lq + q) * M + m;
        const f4 *lp = reinterpret_cast<const f4 *>(loc + (qm * 4 + j) * 8);
        const f4 xy01 = __builtin_nontemporal_load(lp), xy23 = __builtin_nontemporal_load(lp + 1);
        const f4 a4 = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(attn + (qm * 4 + j) * 4));
        const float xs[4] = {xy01.x, xy01.z, xy23.x, xy23.z};
        const float ys[4] = {xy01.y, xy01.w, xy23.y, xy23.w};
        const float as[4] = {a4.x, a4.y, a4.z, a4.w};

        int addr[4][4];
        float wgt[4][4];
        bool miss[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float h_im = ys[p] * Hf - 0.5f, w_im = xs[p] * Wf - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lh = h_im - hf, lw = w_im - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const int iy = inside ? (int)hf : 0, ix = inside ? (int)wf : 0;
            // level 0: byte offsets of the 4 corner rows, out-of-image corners out of range
            const bool top = iy >= 0, bot = iy + 1 <= Hj - 1, lef = ix >= 0, rig = ix + 1 <= Wj - 1;
            const unsigned pix = (unsigned)(start0 + iy * Wj + ix) * row_stride;
            const unsigned g0 = (inside && top && lef) ? pix : kOutOfRange;
            const unsigned g1 = (inside && top && rig) ? pix + row_stride : kOutOfRange;
            const unsigned g2 = (inside && bot && lef) ? pix + (unsigned)Wj * row_stride : kOutOfRange;
            const unsigned g3 = (inside && bot && rig) ? pix + (unsigned)(Wj + 1) * row_stride : kOutOfRange;
            // levels 1..3: LDS addresses inside the window (zero apron covers the image border)
            const int wy = iy - wyj, wx = ix - wxj;
            const bool inwin = (unsigned)wy <= (unsigned)(WHj - 2) && (unsigned)wx <= (unsigned)(WWj - 2);
            const bool use = j == 0 ? inside : (inside && inwin);
            miss[p] = j != 0 && inside && !inwin;
            const int lb = ldsj + ((inwin ? wy : 0) * WWj + (inwin ? wx : 0)) * kRowBytes;
            addr[p][0] = j == 0 ? (int)g0 : lb;
            addr[p][1] = j == 0 ? (int)g1 : lb + kRowBytes;
            addr[p][2] = j == 0 ? (int)g2 : lb + WWj * kRowBytes;
            addr[p][3] = j == 0 ? (int)g3 : lb + WWj * kRowBytes + kRowBytes;
            const float a = use ? as[p] : 0.f;
            wgt[p][0] = a * (hh * hw);
            wgt[p][1] = a * (hh * lw);
            wgt[p][2] = a * (lh * hw);
            wgt[p][3] = a * (lh * lw);
        }

        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

        // ---- slow path: samples of levels 1..3 outside their window come from global memory ---
        if (__builtin_amdgcn_ballot_w64(miss[0] | miss[1] | miss[2] | miss[3]) != 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                for (int l = 1; l < 4; ++l) {
                    // the quad's level-l lane publishes (x, y, a) of point p if it missed
                    const int src = (lane & ~3) | l;
                    const int flag = __shfl((int)miss[p], src, 64);
                    if (__builtin_amdgcn_ballot_w64(flag != 0) == 0) continue;
                    const float x = __shfl(xs[p], src, 64), y = __shfl(ys[p], src, 64);
                    const float a = flag ? __shfl(as[p], src, 64) : 0.f;
                    const int Hl = pm.H[l], Wl = pm.W[l];
                    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const int iy = flag ? (int)hf : 0, ix = flag ? (int)wf : 0;
                    const bool top = iy >= 0, bot = iy + 1 <= Hl - 1, lef = ix >= 0, rig = ix + 1 <= Wl - 1;
                    const unsigned pix = (unsigned)(pm.start[l] + iy * Wl + ix) * row_stride + (unsigned)chan;
                    const unsigned o0 = (flag && top && lef) ? pix : kOutOfRange;
                    const unsigned o1 = (flag && top && rig) ? pix + row_stride : kOutOfRange;
                    const unsigned o2 = (flag && bot && lef) ? pix + (unsigned)Wl * row_stride : kOutOfRange;
                    const unsigned o3 = (flag && bot && rig) ? pix + (unsigned)(Wl + 1) * row_stride : kOutOfRange;
                    fma4(acc0, a * (hh * hw), load_row4(rsrc, o0));
                    fma4(acc1, a * (hh * hw), load_row4(rsrc, o0 ^ 16u));
                    fma4(acc0, a * (hh * lw), load_row4(rsrc, o1));
                    fma4(acc1, a * (hh * lw), load_row4(rsrc, o1 ^ 16u));
                    fma4(acc0, a * (lh * hw), load_row4(rsrc, o2));
                    fma4(acc1, a * (lh * hw), load_row4(rsrc, o2 ^ 16u));
                    fma4(acc0, a * (lh * lw), load_row4(rsrc, o3));
                    fma4(acc1, a * (lh * lw), load_row4(rsrc, o3 ^ 16u));
                }
            }
        }

        // ---- level 0 through the vector-memory path, levels 1..3 out of LDS ---------------------
        // Two level-0 points are in flight on the vector-memory path while a level's sixteen
        // corner rows come out of LDS; the scheduling fences keep the compiler from hoisting
        // every load of the task to the top (which overflows the register file).
        {
            f4 ga[2][4], gb[2][4];
            fetch_point<0, true>(ga[0], gb[0], addr[0], chan, lds, rsrc);
            fetch_point<0, true>(ga[1], gb[1], addr[1], chan, lds, rsrc);
            PIN(acc0, acc1);
            lds_level<1>(acc0, acc1, addr, wgt, chan, lds, rsrc);
            PIN(acc0, acc1);
            accumulate_point<0>(acc0, acc1, ga[0], gb[0], wgt[0]);
            accumulate_point<0>(acc0, acc1, ga[1], gb[1], wgt[1]);
            fetch_point<0, true>(ga[0], gb[0], addr[2], chan, lds, rsrc);
            fetch_point<0, true>(ga[1], gb[1], addr[3], chan, lds, rsrc);
            PIN(acc0, acc1);
            lds_level<2>(acc0, acc1, addr, wgt, chan, lds, rsrc);
            PIN(acc0, acc1);
            lds_level<3>(acc0, acc1, addr, wgt, chan, lds, rsrc);
            PIN(acc0, acc1);
            accumulate_point<0>(acc0, acc1, ga[0], gb[0], wgt[2]);
            accumulate_point<0>(acc0, acc1, ga[1], gb[1], wgt[3]);
        }

        if (live) {
            float *dst = out + qm * 32;
            __builtin_nontemporal_store(acc0, reinterpret_cast<f4 *>(dst + (chan >> 2)));
            __builtin_nontemporal_store(acc1, reinterpret_cast<f4 *>(dst + ((chan ^ 16) >> 2)));
        }
    }
}

// ceil(a / b) for b > 0 and any a
inline long ceil_div(long a, long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

bool build_pyr_meta(PyrMeta &pm, const int64_t *sh, const int64_t *ls, int64_t S, float halo) {
    long total = 0;
    for (int l = 0; l < 4; ++l) {
        const long H = sh[2 * l], W = sh[2 * l + 1];
        if (H < 1 || W < 1 || H > 4096 || W > 4096 || ls[l] != total) return false;
        pm.H[l] = (int)H; pm.W[l] = (int)W; pm.start[l] = (int)total;
        total += H * W;
    }
    if (total != S || S * 8 * 128 >= (1L << 31)) return false;
    // level l must be the coarser the larger l (windows are sized for a pyramid)
    for (int l = 1; l < 4; ++l)
        if (pm.H[l] > pm.H[l - 1] || pm.W[l] > pm.W[l - 1]) return false;
    const int H0 = pm.H[0], W0 = pm.W[0];
    int nRy = std::min(kMaxR, std::max(1, (int)std::lround(H0 / 12.5)));
    int nRx = std::min(kMaxR, std::max(1, (int)std::lround(W0 / 28.0)));
    for (;;) {
        pm.nRy = nRy; pm.nRx = nRx;
        for (int axis = 0; axis < 2; ++axis) {
            const int nR = axis ? nRx : nRy;
            const int *dim = axis ? pm.W : pm.H;
            short (*qb)[kMaxR + 1] = axis ? pm.xb : pm.yb;
            short (*w0)[kMaxR] = axis ? pm.wx0 : pm.wy0;
            int *wdim = axis ? pm.WW : pm.WH;
            const long D0 = dim[0];
            for (int l = 0; l < 4; ++l) {
                for (int i = 0; i <= nR; ++i) {
                    const long b0 = (long)i * D0 / nR;                 // level-0 boundary
                    // first pixel of level l whose centre (y + 0.5) / D_l >= b0 / D0
                    long y = ceil_div(2 * b0 * dim[l] - D0, 2 * D0);
                    y = std::min<long>(std::max<long>(y, 0), dim[l]);
                    qb[l][i] = (short)(i == nR ? dim[l] : y);
                }
            }
            for (int l = 1; l < 4; ++l) {
                int widest = 2;
                for (int i = 0; i < nR; ++i) {
                    double lo = 1e30, hi = -1e30;
                    for (int lq = 0; lq < 4; ++lq) {
                        if (qb[lq][i + 1] <= qb[lq][i]) continue;
                        lo = std::min(lo, (qb[lq][i] + 0.5) / dim[lq] * dim[l] - 0.5);
                        hi = std::max(hi, (qb[lq][i + 1] - 0.5) / dim[lq] * dim[l] - 0.5);
                    }
                    if (lo > hi) { lo = hi = 0; }
                    const int a = (int)std::floor(lo - halo), b = (int)std::floor(hi + halo) + 1;
                    w0[l][i] = (short)a;
                    widest = std::max(widest, b - a + 1);
                }
                wdim[l] = widest;
            }
        }
        int bytes = 0;
        for (int l = 1; l < 4; ++l) {
            pm.lds_base[l] = bytes;
            bytes += pm.WH[l] * pm.WW[l] * kRowBytes;
        }
        pm.lds_base[0] = 0; pm.WH[0] = pm.WW[0] = 0;
        pm.lds_bytes = bytes;
        if (bytes <= kMaxLds) return true;
        // too large: more, smaller regions along the longer region side
        if ((double)H0 / nRy >= (double)W0 / nRx && nRy < kMaxR) ++nRy;
        else if (nRx < kMaxR) ++nRx;
        else if (nRy < kMaxR) ++nRy;
        else return false;
    }
}

}  // namespace

// Internal entry (msda.hip dispatches here): DATR_EUNSUPPORTED when the shape is not covered.
extern "C" int datr_internal_msda_fwd_pyr_d32(
    const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L,
    int64_t Lq, int64_t P, float *out, void *stream)
{
    if (D != 32 || L != 4 || P != 4 || Lq != S || M < 1 || N < 1) return DATR_EUNSUPPORTED;
    static const float halo = [] {
        const char *e = std::getenv("DATR_MSDA_PYR_HALO");
        const float h = e ? (float)std::atof(e) : 4.5f;
        return h >= 0.5f && h <= 16.f ? h : 4.5f;
    }();
    PyrMeta pm;
    if (!build_pyr_meta(pm, shapes_host, level_start_host, S, halo)) return DATR_EUNSUPPORTED;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_pyr_d32),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kMaxLds) == hipSuccess;
    if (!attr_ok) return DATR_EUNSUPPORTED;
    const long blocks = (long)N * pm.nRy * pm.nRx * M;
    if (blocks <= 0 || blocks >= (1L << 31)) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(msda_fwd_pyr_d32, dim3((unsigned)blocks), dim3(kThreads), (size_t)pm.lds_bytes,
                       (hipStream_t)stream, value, loc, attn, pm, (int)S, (int)M, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
