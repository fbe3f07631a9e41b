// msda.hip -- multi-scale deformable attention (sampling + aggregation) for gfx950.
//
// What it computes (behaviour of the reference's native op, restated for CDNA4):
//   out[n,q,m,:] = sum_{l,p} attn[n,q,m,l,p] * bilinear(value_l[n,:,m,:], loc[n,q,m,l,p])
//   /root/reference/models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (forward),
//   :87-159 + :301-403 (backward), pixel mapping h = y*H - 0.5, w = x*W - 0.5 and the
//   in-range test of :285-288.
//
// How it is laid out for MI355X (see DESIGN.md "MSDA kernels"):
//   * A (pixel, head) row of `value` is D contiguous floats (128 B at D = 32 = one cache
//     line).  The fast path gives each lane a float4 of that row, so LPR = D/4 lanes cover a
//     row and a 64-wide wavefront works on 64/LPR (query, head) outputs at once; a
//     256-thread workgroup covers 256/LPR consecutive queries of ONE head.
//   * blockIdx -> (n, query tile, head) with the head index fastest.  The dispatcher places
//     workgroup b on XCD b % 8, so with M = 8 heads every XCD only ever touches one head's
//     rows: its private 4 MiB L2 holds that head's 2.8 MB slice of a 1333x800 pyramid.
//     (Affinity only -- nothing depends on the placement for correctness.)
//   * Phase 1: the workgroup turns each (query, sample) pair into four byte offsets and four
//     weights ONCE and parks them in LDS; phase 2 re-reads them with broadcast
//     ds_read_b128 instead of having every channel lane redo the address arithmetic
//     (the reference re-loads loc/attn from global memory in each of the D channel threads).
//   * Corner rows are fetched with raw buffer loads: an out-of-image corner gets an offset
//     beyond the descriptor's num_records, which the hardware answers with zeros without
//     touching memory -- no per-corner branches.  The backward's float atomics use the same
//     descriptor trick (out-of-range buffer atomics are dropped).
//   * Backward channel reductions are DPP lane permutes inside the LPR-lane row, not LDS +
//     a serial thread-0 loop (cuh:376-394), and the block is never a half-empty wave.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "datr_hip.h"
#include "msda_tiled.h"

namespace {

constexpr int kThreads = 256;
constexpr unsigned kOutOfRange = 0x80000000u;   // >= num_records of every descriptor we build
constexpr int kMaxFastLdsBytes = 64 * 1024;

struct LevelMeta { int H, W, start, pad; };      // 16 B in LDS

__device__ __forceinline__ float4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    // NB: keep `auto` -- converting the builtin's result to an ext_vector typedef splats lane 0.
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128 load must return 16 bytes");
    return __builtin_bit_cast(float4, r);
}

// Sum over the LPR lanes that share one (query, head) row; every lane ends with the total.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xF, 0xF, false);
    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xF, 0xF, false);
    if (LPR >= 8)   // row_half_mirror: lane j <-> 7-j inside each 8 lanes (the other quad)
        v += __builtin_amdgcn_update_dpp(0.f, v, 0x141, 0xF, 0xF, false);
    if (LPR >= 16)  // row_mirror: lane j <-> 15-j inside each 16 lanes (the other half)
        v += __builtin_amdgcn_update_dpp(0.f, v, 0x140, 0xF, 0xF, false);
    return v;
}

// Geometry of one sampling point, shared by forward and backward phase 1.
struct Corner4 {
    unsigned off[4];   // byte offsets of the 4 corner rows from the (n, head) base, or kOutOfRange
    float lh, lw;      // fractional parts
    bool any;
};

__device__ __forceinline__ Corner4 locate(float x, float y, const LevelMeta lm, unsigned row_bytes,
                                          bool live) {
    Corner4 c;
    const float Hf = (float)lm.H, Wf = (float)lm.W;
    const float h_im = y * Hf - 0.5f;
    const float w_im = x * Wf - 0.5f;
    const bool inside = live && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = inside ? (int)hf : 0, w0 = inside ? (int)wf : 0;
    c.lh = h_im - hf;
    c.lw = w_im - wf;
    const bool top = inside && h0 >= 0, bot = inside && h0 + 1 <= lm.H - 1;
    const bool lef = w0 >= 0, rig = w0 + 1 <= lm.W - 1;
    const int pix = lm.start + h0 * lm.W + w0;
    c.off[0] = (top && lef) ? (unsigned)pix * row_bytes : kOutOfRange;
    c.off[1] = (top && rig) ? (unsigned)(pix + 1) * row_bytes : kOutOfRange;
    c.off[2] = (bot && lef) ? (unsigned)(pix + lm.W) * row_bytes : kOutOfRange;
    c.off[3] = (bot && rig) ? (unsigned)(pix + lm.W + 1) * row_bytes : kOutOfRange;
    c.any = inside;
    return c;
}

__device__ __forceinline__ void load_level_meta(LevelMeta *meta, const int64_t *shapes,
                                                const int64_t *level_start, int L) {
    for (int l = threadIdx.x; l < L; l += kThreads) {
        LevelMeta lm;
        lm.H = (int)shapes[2 * l];
        lm.W = (int)shapes[2 * l + 1];
        lm.start = (int)level_start[l];
        lm.pad = 0;
        meta[l] = lm;
    }
}

// ------------------------------------------------------------------------------------------
// Forward, fast path.  LDS per sample: {off[4], w[4]} = 32 B; slot stride K*32+16 B keeps
// the broadcast ds_read_b128 of neighbouring rows on different banks.
// ------------------------------------------------------------------------------------------
template <int LPR, int KS>   // KS = compile-time L*P (0: run-time)
__global__ __launch_bounds__(kThreads) void msda_fwd_rows(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int q_tiles,
    float *__restrict__ out)
{
    constexpr int D = 4 * LPR;
    constexpr int QPB = kThreads / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = KS ? KS : L * P;
    const int slot_bytes = K * 32 + 16;
    LevelMeta *meta = reinterpret_cast<LevelMeta *>(smem);
    char *stage = smem + ((L * (int)sizeof(LevelMeta) + 15) & ~15);

    const int bid = blockIdx.x;
    const int m = bid % M;
    const int qt = (bid / M) % q_tiles;
    const int n = bid / (M * q_tiles);

    const int row = threadIdx.x / LPR;          // which (query) row of the block
    const int j = threadIdx.x % LPR;            // lane inside the row -> channels 4j..4j+3
    const int q = qt * QPB + row;
    const bool live = q < Lq;
    const int qc = live ? q : Lq - 1;
    const unsigned row_bytes = (unsigned)(M * D) * 4u;

    load_level_meta(meta, shapes, level_start, L);
    __syncthreads();

    // ---- phase 1: per-sample offsets and weights into LDS -------------------------------
    {
        const size_t qm = ((size_t)n * Lq + qc) * M + m;
        const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + qm * K;
        const float *att = attn + qm * K;
        char *slot = stage + row * slot_bytes;
        for (int s = j; s < K; s += LPR) {
            const float2 xy = loc2[s];
            const float a = att[s];
            const LevelMeta lm = meta[s / P];
            const Corner4 c = locate(xy.x, xy.y, lm, row_bytes, live);
            const float hh = 1.f - c.lh, hw = 1.f - c.lw;
            uint4 o = make_uint4(c.off[0], c.off[1], c.off[2], c.off[3]);
            float4 w = make_float4(a * (hh * hw), a * (hh * c.lw), a * (c.lh * hw), a * (c.lh * c.lw));
            if (!c.any) w = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<uint4 *>(slot + s * 32) = o;
            *reinterpret_cast<float4 *>(slot + s * 32 + 16) = w;
        }
    }
    __syncthreads();

    // ---- phase 2: gather + accumulate -----------------------------------------------------
    const float *base = value + ((size_t)n * S * M + m) * D;
    const int records = (S * M - m) * D * 4;      // bytes from `base` to the end of batch item n
    __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, records, 0x00020000);
    const unsigned chan = (unsigned)j * 16u;
    const char *slot = stage + row * slot_bytes;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = 0; s < K; ++s) {
        const uint4 o = *reinterpret_cast<const uint4 *>(slot + s * 32);
        const float4 w = *reinterpret_cast<const float4 *>(slot + s * 32 + 16);
        const float4 v0 = load_row4(rsrc, o.x + chan);
        const float4 v1 = load_row4(rsrc, o.y + chan);
        const float4 v2 = load_row4(rsrc, o.z + chan);
        const float4 v3 = load_row4(rsrc, o.w + chan);
        acc.x += w.x * v0.x + w.y * v1.x + w.z * v2.x + w.w * v3.x;
        acc.y += w.x * v0.y + w.y * v1.y + w.z * v2.y + w.w * v3.y;
        acc.z += w.x * v0.z + w.y * v1.z + w.z * v2.z + w.w * v3.z;
        acc.w += w.x * v0.w + w.y * v1.w + w.z * v2.w + w.w * v3.w;
    }
    if (live) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 *dst = reinterpret_cast<f4 *>(out + (((size_t)n * Lq + q) * M + m) * D) + j;
        const f4 r = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(r, dst);
    }
}

// ------------------------------------------------------------------------------------------
// Backward, fast path.  LDS per sample: {off[4]}, {lh, lw, a*W, a*H}, {a, -, -, -} = 48 B.
//
// Lane mapping differs from the forward on purpose.  Measured on MI355X
// (tools/probes/atomics_probe2.hip, profiles/r01_probes.md): float atomics retire per 64-B
// request -- 80 G atomics/s when the lanes of one row hit dwords 16 B apart (the float4
// mapping), 326 G/s when 16 lanes hit 16 consecutive dwords.  So here 16 lanes share a row
// and lane j owns channels j, j+16, ... (CPL = D/16 of them): every atomic instruction and
// every load instruction of a row covers one contiguous 64-B half line.
// ------------------------------------------------------------------------------------------
template <int CPL, int KS>   // CPL = D/16 channels per lane
__global__ __launch_bounds__(kThreads) void msda_bwd_rows(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq,
    int P, int q_tiles, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn)
{
    constexpr int LPR = 16;
    constexpr int D = LPR * CPL;
    constexpr int QPB = kThreads / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = KS ? KS : L * P;
    const int slot_bytes = K * 48 + 16;
    LevelMeta *meta = reinterpret_cast<LevelMeta *>(smem);
    char *stage = smem + ((L * (int)sizeof(LevelMeta) + 15) & ~15);

    const int bid = blockIdx.x;
    const int m = bid % M;
    const int qt = (bid / M) % q_tiles;
    const int n = bid / (M * q_tiles);

    const int row = threadIdx.x / LPR;
    const int j = threadIdx.x % LPR;
    const int q = qt * QPB + row;
    const bool live = q < Lq;
    const int qc = live ? q : Lq - 1;
    const unsigned row_bytes = (unsigned)(M * D) * 4u;
    const size_t qm = ((size_t)n * Lq + qc) * M + m;

    load_level_meta(meta, shapes, level_start, L);
    __syncthreads();

    {
        const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + qm * K;
        const float *att = attn + qm * K;
        char *slot = stage + row * slot_bytes;
        for (int s = j; s < K; s += LPR) {
            const float2 xy = loc2[s];
            const float a = att[s];
            const LevelMeta lm = meta[s / P];
            const Corner4 c = locate(xy.x, xy.y, lm, row_bytes, live);
            *reinterpret_cast<uint4 *>(slot + s * 48) =
                make_uint4(c.off[0], c.off[1], c.off[2], c.off[3]);
            // a skipped sample keeps finite placeholders so 0-filled loads give exact zeros
            *reinterpret_cast<float4 *>(slot + s * 48 + 16) =
                make_float4(c.any ? c.lh : 0.f, c.any ? c.lw : 0.f, a * (float)lm.W,
                            a * (float)lm.H);
            *reinterpret_cast<float *>(slot + s * 48 + 32) = a;
        }
    }
    __syncthreads();

    const size_t item = ((size_t)n * S * M + m) * D;
    const int records = (S * M - m) * D * 4;
    __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(value + item), 0, records, 0x00020000);
    __amdgpu_buffer_rsrc_t gsrc =
        __builtin_amdgcn_make_buffer_rsrc(grad_value + item, 0, records, 0x00020000);
    const unsigned chan = (unsigned)j * 4u;            // byte offset of this lane's first channel
    const char *slot = stage + row * slot_bytes;

    float go[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) go[c] = live ? grad_out[qm * D + j + c * LPR] : 0.f;

    for (int t = 0; t < K; t += LPR) {
        float keep_w = 0.f, keep_h = 0.f, keep_a = 0.f;
#pragma unroll
        for (int i = 0; i < LPR; ++i) {
            const int s = t + i;
            if (s < K) {
                const uint4 o = *reinterpret_cast<const uint4 *>(slot + s * 48);
                const float4 g = *reinterpret_cast<const float4 *>(slot + s * 48 + 16);
                const float a = *reinterpret_cast<const float *>(slot + s * 48 + 32);
                const float lh = g.x, lw = g.y, aW = g.z, aH = g.w;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float c0 = hh * hw, c1 = hh * lw, c2 = lh * hw, c3 = lh * lw;
                float pa = 0.f, pw = 0.f, ph = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const unsigned co = chan + (unsigned)(c * LPR * 4);
                    const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vsrc, o.x + co, 0, 0));
                    const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vsrc, o.y + co, 0, 0));
                    const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vsrc, o.z + co, 0, 0));
                    const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vsrc, o.w + co, 0, 0));
                    // d out / d value at the four corners (dropped by hardware when out of range)
                    const float ga = go[c] * a;
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c0 * ga, gsrc, o.x + co, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c1 * ga, gsrc, o.y + co, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c2 * ga, gsrc, o.z + co, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(c3 * ga, gsrc, o.w + co, 0, 0);
                    // this lane's share of the three per-sample reductions
                    pa += go[c] * (c0 * v0 + c1 * v1 + c2 * v2 + c3 * v3);
                    pw += go[c] * (hh * (v1 - v0) + lh * (v3 - v2));
                    ph += go[c] * (hw * (v2 - v0) + lw * (v3 - v1));
                }
                pa = row_sum<LPR>(pa);
                pw = row_sum<LPR>(pw) * aW;
                ph = row_sum<LPR>(ph) * aH;
                if (j == i) { keep_a = pa; keep_w = pw; keep_h = ph; }
            }
        }
        const int s = t + j;
        if (live && s < K) {
            reinterpret_cast<float2 *>(grad_loc)[qm * K + s] = make_float2(keep_w, keep_h);
            grad_attn[qm * K + s] = keep_a;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic path: any D, float or double, 64-bit indexing.  One thread per output scalar
// (forward) / one workgroup per (n, q, m) with a block reduction per sample (backward).
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool locate_generic(T x, T y, int64_t H, int64_t W, int64_t &h0,
                                               int64_t &w0, T &lh, T &lw) {
    const T h_im = y * (T)H - (T)0.5;
    const T w_im = x * (T)W - (T)0.5;
    if (!(h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W)) return false;
    const T hf = floor(h_im), wf = floor(w_im);
    h0 = (int64_t)hf;
    w0 = (int64_t)wf;
    lh = h_im - hf;
    lw = w_im - wf;
    return true;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void msda_fwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const T *__restrict__ loc,
    const T *__restrict__ attn, int64_t total, int64_t S, int64_t M, int64_t D, int64_t L,
    int64_t Lq, int64_t P, T *__restrict__ out)
{
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * kThreads) {
        const int64_t c = idx % D;
        const int64_t qm = idx / D;
        const int64_t m = qm % M;
        const int64_t n = qm / (M * Lq);
        const int64_t rowlen = M * D;
        T acc = (T)0;
        for (int64_t l = 0; l < L; ++l) {
            const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
            const T *vb = value + (n * S + level_start[l]) * rowlen + m * D + c;
            for (int64_t p = 0; p < P; ++p) {
                const int64_t k = qm * L * P + l * P + p;
                int64_t h0, w0;
                T lh, lw;
                if (!locate_generic<T>(loc[2 * k], loc[2 * k + 1], H, W, h0, w0, lh, lw)) continue;
                const T hh = (T)1 - lh, hw = (T)1 - lw;
                const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
                const T v0 = (top && lef) ? vb[(h0 * W + w0) * rowlen] : (T)0;
                const T v1 = (top && rig) ? vb[(h0 * W + w0 + 1) * rowlen] : (T)0;
                const T v2 = (bot && lef) ? vb[((h0 + 1) * W + w0) * rowlen] : (T)0;
                const T v3 = (bot && rig) ? vb[((h0 + 1) * W + w0 + 1) * rowlen] : (T)0;
                acc += (hh * hw * v0 + hh * lw * v1 + lh * hw * v2 + lh * lw * v3) * attn[k];
            }
        }
        out[idx] = acc;
    }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch) {
    // wave64 butterfly, then one LDS round across the (<= 4) waves of the block
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    T tot = (T)0;
    for (int w = 0; w < nw; ++w) tot += scratch[w];
    return tot;
}

template <typename T>
__global__ void msda_bwd_generic(
    const T *__restrict__ grad_out, const T *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ attn, int64_t rows, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, T *__restrict__ grad_value,
    T *__restrict__ grad_loc, T *__restrict__ grad_attn)
{
    __shared__ T scratch[4];
    const int64_t rowlen = M * D;
    for (int64_t qm = blockIdx.x; qm < rows; qm += gridDim.x) {
        const int64_t m = qm % M;
        const int64_t n = qm / (M * Lq);
        for (int64_t l = 0; l < L; ++l) {
            const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
            const int64_t lvl = (n * S + level_start[l]) * rowlen + m * D;
            for (int64_t p = 0; p < P; ++p) {
                const int64_t k = qm * L * P + l * P + p;
                int64_t h0 = 0, w0 = 0;
                T lh = 0, lw = 0;
                const bool inside =
                    locate_generic<T>(loc[2 * k], loc[2 * k + 1], H, W, h0, w0, lh, lw);
                T pa = (T)0, pw = (T)0, ph = (T)0;
                if (inside) {
                    const T a = attn[k];
                    const T hh = (T)1 - lh, hw = (T)1 - lw;
                    const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0,
                               rig = w0 + 1 <= W - 1;
                    const int64_t o0 = lvl + (h0 * W + w0) * rowlen, o1 = o0 + rowlen,
                                  o2 = o0 + W * rowlen, o3 = o2 + rowlen;
                    for (int64_t c = threadIdx.x; c < D; c += blockDim.x) {
                        const T g = grad_out[qm * D + c];
                        const T ga = g * a;
                        const T v0 = (top && lef) ? value[o0 + c] : (T)0;
                        const T v1 = (top && rig) ? value[o1 + c] : (T)0;
                        const T v2 = (bot && lef) ? value[o2 + c] : (T)0;
                        const T v3 = (bot && rig) ? value[o3 + c] : (T)0;
                        if (top && lef) atomicAdd(grad_value + o0 + c, hh * hw * ga);
                        if (top && rig) atomicAdd(grad_value + o1 + c, hh * lw * ga);
                        if (bot && lef) atomicAdd(grad_value + o2 + c, lh * hw * ga);
                        if (bot && rig) atomicAdd(grad_value + o3 + c, lh * lw * ga);
                        pa += g * (hh * hw * v0 + hh * lw * v1 + lh * hw * v2 + lh * lw * v3);
                        pw += ga * (T)W * (hh * (v1 - v0) + lh * (v3 - v2));
                        ph += ga * (T)H * (hw * (v2 - v0) + lw * (v3 - v1));
                    }
                }
                pa = block_sum<T>(pa, scratch);
                pw = block_sum<T>(pw, scratch);
                ph = block_sum<T>(ph, scratch);
                if (threadIdx.x == 0) {
                    grad_attn[k] = pa;
                    grad_loc[2 * k] = pw;
                    grad_loc[2 * k + 1] = ph;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------
bool dims_ok(int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P) {
    return N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0;
}

int fast_lpr(int64_t S, int64_t M, int64_t D, int64_t L, int64_t P, int bytes_per_sample) {
    if (D != 16 && D != 32 && D != 64) return 0;
    const int lpr = (int)(D / 4);
    const int64_t K = L * P;
    if (K > 256 || L > 256) return 0;
    if (S * M * D * 4 >= (int64_t)kOutOfRange) return 0;          // 32-bit byte offsets per item
    const int64_t lds = ((L * 16 + 15) & ~15) + (int64_t)(kThreads / lpr) * (K * bytes_per_sample + 16);
    if (lds > kMaxFastLdsBytes) return 0;
    return lpr;
}

int check_launch() { return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH; }

template <int LPR>
int launch_fwd_rows(const float *value, const int64_t *shapes, const int64_t *ls, const float *loc,
                    const float *attn, int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq,
                    int64_t P, float *out, hipStream_t st) {
    const int qpb = kThreads / LPR;
    const int q_tiles = (int)((Lq + qpb - 1) / qpb);
    const int64_t blocks = N * q_tiles * M;
    if (blocks > 0x7fffffff) return DATR_EUNSUPPORTED;
    const int K = (int)(L * P);
    const size_t lds = ((L * 16 + 15) & ~15) + (size_t)qpb * (K * 32 + 16);
    if (K == 16)
        hipLaunchKernelGGL((msda_fwd_rows<LPR, 16>), dim3((unsigned)blocks), dim3(kThreads), lds, st,
                           value, shapes, ls, loc, attn, (int)S, (int)M, (int)L, (int)Lq, (int)P,
                           q_tiles, out);
    else
        hipLaunchKernelGGL((msda_fwd_rows<LPR, 0>), dim3((unsigned)blocks), dim3(kThreads), lds, st,
                           value, shapes, ls, loc, attn, (int)S, (int)M, (int)L, (int)Lq, (int)P,
                           q_tiles, out);
    return check_launch();
}

template <int CPL>
int launch_bwd_rows(const float *go, const float *value, const int64_t *shapes, const int64_t *ls,
                    const float *loc, const float *attn, int64_t N, int64_t S, int64_t M, int64_t L,
                    int64_t Lq, int64_t P, float *gv, float *gl, float *ga, hipStream_t st) {
    const int qpb = kThreads / 16;
    const int q_tiles = (int)((Lq + qpb - 1) / qpb);
    const int64_t blocks = N * q_tiles * M;
    if (blocks > 0x7fffffff) return DATR_EUNSUPPORTED;
    const int K = (int)(L * P);
    const size_t lds = ((L * 16 + 15) & ~15) + (size_t)qpb * (K * 48 + 16);
    if (K == 16)
        hipLaunchKernelGGL((msda_bwd_rows<CPL, 16>), dim3((unsigned)blocks), dim3(kThreads), lds, st,
                           go, value, shapes, ls, loc, attn, (int)S, (int)M, (int)L, (int)Lq, (int)P,
                           q_tiles, gv, gl, ga);
    else
        hipLaunchKernelGGL((msda_bwd_rows<CPL, 0>), dim3((unsigned)blocks), dim3(kThreads), lds, st,
                           go, value, shapes, ls, loc, attn, (int)S, (int)M, (int)L, (int)Lq, (int)P,
                           q_tiles, gv, gl, ga);
    return check_launch();
}

template <typename T>
int forward_generic(const T *value, const int64_t *shapes, const int64_t *ls, const T *loc,
                    const T *attn, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq,
                    int64_t P, T *out, hipStream_t st) {
    const int64_t total = N * Lq * M * D;
    int64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((msda_fwd_generic<T>), dim3((unsigned)blocks), dim3(kThreads), 0, st, value,
                       shapes, ls, loc, attn, total, S, M, D, L, Lq, P, out);
    return check_launch();
}

template <typename T>
int backward_generic(const T *go, const T *value, const int64_t *shapes, const int64_t *ls,
                     const T *loc, const T *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                     int64_t L, int64_t Lq, int64_t P, T *gv, T *gl, T *ga, hipStream_t st) {
    const int64_t rows = N * Lq * M;
    int64_t blocks = rows < 262144 ? rows : 262144;
    int threads = (int)(((D < 256 ? D : 256) + 63) / 64 * 64);
    hipLaunchKernelGGL((msda_bwd_generic<T>), dim3((unsigned)blocks), dim3(threads), 0, st, go, value,
                       shapes, ls, loc, attn, rows, S, M, D, L, Lq, P, gv, gl, ga);
    return check_launch();
}

template <typename T>
int forward_any(const T *value, const int64_t *shapes, const int64_t *ls, const T *loc,
                const T *attn, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq,
                int64_t P, T *out, void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0 || Lq == 0) return DATR_OK;
    if (!value || !shapes || !ls || !loc || !attn || !out) return DATR_EINVAL;
    return forward_generic<T>(value, shapes, ls, loc, attn, N, S, M, D, L, Lq, P, out,
                              (hipStream_t)stream);
}

}  // namespace

extern "C" {

const char *datr_strerror(int code) {
    switch (code) {
        case DATR_OK: return "ok";
        case DATR_EINVAL: return "invalid argument (null pointer or bad dimension)";
        case DATR_EUNSUPPORTED: return "shape not supported by the 32-bit indexed kernels";
        case DATR_ELAUNCH: return "HIP kernel launch failed";
        default: return "unknown datr error";
    }
}

int datr_abi_version(void) { return 2; }

int datr_msda_uses_fast_path(int64_t S, int64_t M, int64_t D, int64_t L, int64_t P) {
    return fast_lpr(S, M, D, L, P, 48) != 0;
}

int datr_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                          const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                          int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0 || Lq == 0) return DATR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !out) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (fast_lpr(S, M, D, L, P, 48)) {
        case 4: return launch_fwd_rows<4>(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, st);
        case 8: return launch_fwd_rows<8>(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, st);
        case 16: return launch_fwd_rows<16>(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, st);
        default: return forward_generic<float>(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out, st);
    }
}

int datr_msda_backward_f32(const float *grad_out, const float *value, const int64_t *shapes,
                           const int64_t *level_start, const float *loc, const float *attn,
                           int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq,
                           int64_t P, float *grad_value, float *grad_loc, float *grad_attn,
                           void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0) return DATR_OK;
    if (!value || !grad_value) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_value, 0, (size_t)(N * S * M * D) * sizeof(float), st) != hipSuccess)
        return DATR_ELAUNCH;
    if (Lq == 0) return DATR_OK;
    if (!grad_out || !shapes || !level_start || !loc || !attn || !grad_loc || !grad_attn)
        return DATR_EINVAL;
    switch (fast_lpr(S, M, D, L, P, 48)) {
        case 4: return launch_bwd_rows<1>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn, st);
        case 8: return launch_bwd_rows<2>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn, st);
        case 16: return launch_bwd_rows<4>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn, st);
        default: return backward_generic<float>(grad_out, value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn, st);
    }
}

// Level geometry + query tiling for the tiled kernels; false when the shape is not covered.
static bool build_tiled_meta(DatrTiledMeta &meta, const int64_t *shapes_host,
                             const int64_t *level_start_host, int64_t N, int64_t S, int64_t M,
                             int64_t D, int64_t L, int64_t Lq, int64_t P, bool allow_linear) {
    if (!(shapes_host && level_start_host && D == 32 && L >= 1 && L <= DATR_TILED_MAX_LEVELS &&
          P >= 1 && P <= 4 && N > 0 && Lq >= 64 && S * M * D * 4 < (int64_t)kOutOfRange &&
          Lq * M * L * P < ((int64_t)1 << 30)))
        return false;
    meta.L = (int)L;
    meta.Lq = (int)Lq;
    int64_t expect = 0;
    for (int l = 0; l < L; ++l) {
        const int64_t H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
        if (H <= 0 || W <= 0 || H > 32000 || W > 32000 || level_start_host[l] != expect) return false;
        meta.lv[l] = DatrTileLevel{(int)H, (int)W, (int)expect, 0, 0, 0};
        expect += H * W;
    }
    if (expect != S) return false;
    int base = 0;
    if (Lq == S) {                       // pyramid mode: 16 x 8 pixel tiles per level
        meta.QL = (int)L;
        meta.tile_w = DATR_TILE_W;
        meta.tile_h = DATR_TILE_H;
        for (int l = 0; l < L; ++l) {
            meta.qlv[l] = meta.lv[l];
            meta.qlv[l].tiles_x = (meta.lv[l].W + DATR_TILE_W - 1) / DATR_TILE_W;
            meta.qlv[l].tiles_y = (meta.lv[l].H + DATR_TILE_H - 1) / DATR_TILE_H;
            meta.qlv[l].tile_base = base;
            base += meta.qlv[l].tiles_x * meta.qlv[l].tiles_y;
        }
    } else {                             // linear mode: 128 consecutive queries per tile
        if (!allow_linear) return false;
        meta.QL = 1;
        meta.tile_w = DATR_TILE_LINEAR;
        meta.tile_h = 1;
        meta.qlv[0] = DatrTileLevel{1, (int)Lq, 0,
                                    (int)((Lq + DATR_TILE_LINEAR - 1) / DATR_TILE_LINEAR), 1, 0};
        base = meta.qlv[0].tiles_x;
    }
    meta.total_tiles = base;
    return true;
}

int datr_msda_forward_tiled_f32(const float *value, const int64_t *shapes,
                                const int64_t *level_start, const int64_t *shapes_host,
                                const int64_t *level_start_host, const float *loc,
                                const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                int64_t L, int64_t Lq, int64_t P, float *out, void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0 || Lq == 0) return DATR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !out) return DATR_EINVAL;
    if (shapes_host && level_start_host) {
        // encoder calls: pyramid-region kernel, levels 1..3 gathered out of LDS (msda_fwd_pyr.hip)
        const int rc = datr_internal_msda_fwd_pyr_d32(value, loc, attn, shapes_host, level_start_host,
                                                      N, S, M, D, L, Lq, P, out, stream);
        if (rc != DATR_EUNSUPPORTED) return rc;
    }
    return datr_msda_forward_f32(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out,
                                 stream);
}

int datr_msda_forward_pyramid_f32(const float *value, const int64_t *shapes,
                                  const int64_t *level_start, const int64_t *shapes_host,
                                  const int64_t *level_start_host, const float *envelope_host,
                                  const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                                  int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0 || Lq == 0) return DATR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !out) return DATR_EINVAL;
    // The phased all-LDS kernel (msda_fwd_pyr2.hip) whenever a window plan exists -- with or without
    // an envelope it is ahead of round 2's kernel (N = 4 call at 1333x800: 111 vs 168 us with the
    // measured envelope of the ring initialisation, 135 vs 168 with the symmetric default, 136 vs 180 /
    // 210 vs 222 for offsets ~ N(0, 1.5 / 2.5 px); profiles/r03_msda_fwd.md).  No plan (envelopes too
    // wide for the LDS): round 2's kernel / the row kernel.  DATR_MSDA_PYR2=0: never (A/B runs).
    static const int pyr2 = getenv("DATR_MSDA_PYR2") ? atoi(getenv("DATR_MSDA_PYR2")) : 1;
    // 32-bit sample offsets inside the kernel: N * Lq * M * 16 samples * 8 B
    if (pyr2 && shapes_host && level_start_host && M <= 8 && N * Lq * M * 128 < ((int64_t)1 << 31)) {
        const int rc = datr_internal_msda_fwd_pyr2_d32(value, loc, attn, shapes_host, level_start_host,
                                                       envelope_host, N, S, M, D, L, Lq, P, out, stream);
        if (rc != DATR_EUNSUPPORTED) return rc;
    }
    return datr_msda_forward_tiled_f32(value, shapes, level_start, shapes_host, level_start_host, loc,
                                       attn, N, S, M, D, L, Lq, P, out, stream);
}

int datr_msda_pyramid_plan(const int64_t *shapes_host, const int64_t *level_start_host, int64_t N,
                           int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                           const float *envelope_host, int32_t *info) {
    if (!info || !shapes_host || !level_start_host) return DATR_EINVAL;
    for (int i = 0; i < 16; ++i) info[i] = 0;
    if (!dims_ok(N, S, M, D, L, Lq, P) || D != 32 || L != 4 || P != 4 || Lq != S) return DATR_OK;
    if (M <= 8 && N * Lq * M * 128 < ((int64_t)1 << 31))
        (void)datr_internal_msda_fwd_pyr2_plan(shapes_host, level_start_host, S, M, envelope_host, nullptr, info);
    const int32_t cfg_ = info[12];
    (void)datr_internal_msda_bwd_pyr_plan(shapes_host, level_start_host, S, M, info + 8);
    info[12] = cfg_;
    // [11]: does datr_msda_forward_pyramid_f32 take the phased kernel for this envelope?
    static const int pyr2 = getenv("DATR_MSDA_PYR2") ? atoi(getenv("DATR_MSDA_PYR2")) : 1;
    info[11] = info[0] && pyr2;
    return DATR_OK;
}

static int backward_tiled_impl(const float *grad_out, const float *value, const int64_t *shapes,
                               const int64_t *level_start, const int64_t *shapes_host,
                               const int64_t *level_start_host, const float *loc,
                               const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                               int64_t L, int64_t Lq, int64_t P, float *grad_value,
                               float *grad_loc, float *grad_attn, void *stream, bool allow_pyr,
                               const float *envelope_host = nullptr) {
    // The tiled kernel needs the level geometry on the host (grid size, window maths); anything
    // it does not cover takes the row kernel.
    DatrTiledMeta meta;
    if (!build_tiled_meta(meta, shapes_host, level_start_host, N, S, M, D, L, Lq, P, true))
        return datr_msda_backward_f32(grad_out, value, shapes, level_start, loc, attn, N, S, M, D, L,
                                      Lq, P, grad_value, grad_loc, grad_attn, stream);
    if (!grad_out || !value || !loc || !attn || !grad_value || !grad_loc || !grad_attn)
        return DATR_EINVAL;
    // few, spatially unordered queries (the decoder's cross-attention): workgroups own ranges of
    // value rows and scan the samples -- no global atomics, no zero-fill (msda_bwd_owner.hip)
    if (Lq != S && Lq <= 4096) {
        const int rc = datr_internal_msda_bwd_owner_d32(grad_out, value, loc, attn, &meta, N, S, M, P,
                                                        Lq, grad_value, M * D, grad_loc, grad_attn, stream);
        if (rc != DATR_EUNSUPPORTED) return rc;
    }
    if (hipMemsetAsync(grad_value, 0, (size_t)(N * S * M * D) * sizeof(float),
                       (hipStream_t)stream) != hipSuccess)
        return DATR_ELAUNCH;
    // encoder calls: pyramid regions, grad_value by sorted scatter (msda_bwd_pyr.hip);
    // DATR_MSDA_PYR_BWD=0 keeps the query-tiled kernel (A/B measurements)
    static const bool pyr_bwd = !(getenv("DATR_MSDA_PYR_BWD") && atoi(getenv("DATR_MSDA_PYR_BWD")) == 0);
    if (Lq == S && pyr_bwd && allow_pyr) {
        const int rc = datr_internal_msda_bwd_pyr_d32(grad_out, value, loc, attn, shapes_host,
                                                      level_start_host, N, S, M, D, L, Lq, P, envelope_host,
                                                      grad_value, grad_loc, grad_attn, stream);
        if (rc != DATR_EUNSUPPORTED) return rc;
    }
    return datr_internal_msda_bwd_tiled_d32(grad_out, value, loc, attn, &meta, N, S, M, P,
                                            grad_value, grad_loc, grad_attn, stream);
}

int datr_msda_backward_tiled_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                 const int64_t *level_start, const int64_t *shapes_host,
                                 const int64_t *level_start_host, const float *loc,
                                 const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                 int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                 float *grad_loc, float *grad_attn, void *stream) {
    return backward_tiled_impl(grad_out, value, shapes, level_start, shapes_host, level_start_host, loc, attn, N,
                               S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn, stream, true);
}

int datr_msda_backward_pyramid_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                   const int64_t *level_start, const int64_t *shapes_host,
                                   const int64_t *level_start_host, const float *envelope_host, const float *loc,
                                   const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                   int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                   float *grad_loc, float *grad_attn, void *stream) {
    return backward_tiled_impl(grad_out, value, shapes, level_start, shapes_host, level_start_host, loc, attn, N,
                               S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn, stream, true, envelope_host);
}

int datr_msda_backward_pyramid_query_f32(const float *grad_out, const float *value, const int64_t *shapes_host,
                                         const int64_t *level_start_host, const float *envelope_host,
                                         const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                                         int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                         float *grad_query, void *stream) {
    if (!grad_out || !value || !loc || !attn || !grad_value || !grad_query || !shapes_host || !level_start_host)
        return DATR_EINVAL;
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (Lq != S || D != 32 || M != 8 || L != 4 || P != 4 || N < 1) return DATR_EUNSUPPORTED;
    // the A/B switches of the two-kernel backward this entry is a variant of
    static const bool off = (getenv("DATR_MSDA_PYR_BWD") && atoi(getenv("DATR_MSDA_PYR_BWD")) == 0) ||
                            (getenv("DATR_MSDA_BWD_SPLIT") && atoi(getenv("DATR_MSDA_BWD_SPLIT")) == 0);
    if (off) return DATR_EUNSUPPORTED;
    if (hipMemsetAsync(grad_value, 0, (size_t)(N * S * M * D) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return DATR_ELAUNCH;
    return datr_internal_msda_bwd_pyr_d32(grad_out, value, loc, attn, shapes_host, level_start_host, N, S, M, D, L,
                                          Lq, P, envelope_host, grad_value, grad_query, nullptr, stream, 1);
}

int datr_msda_backward_strided_f32(const float *grad_out, const float *value, const int64_t *shapes_host,
                                   const int64_t *level_start_host, const float *loc, const float *attn,
                                   int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                                   float *grad_value, int64_t grad_value_row_stride, float *grad_loc,
                                   float *grad_attn, void *stream) {
    if (!grad_out || !value || !loc || !attn || !grad_value || !grad_loc || !grad_attn || !shapes_host ||
        !level_start_host)
        return DATR_EINVAL;
    if (grad_value_row_stride < M * D || grad_value_row_stride % 4) return DATR_EINVAL;
    DatrTiledMeta meta;
    // the owner-computes kernel only: it writes every grad_value row exactly once (no zero fill, no atomics)
    if (Lq == S || Lq > 4096 || !build_tiled_meta(meta, shapes_host, level_start_host, N, S, M, D, L, Lq, P, true))
        return DATR_EUNSUPPORTED;
    return datr_internal_msda_bwd_owner_d32(grad_out, value, loc, attn, &meta, N, S, M, P, Lq, grad_value,
                                            grad_value_row_stride, grad_loc, grad_attn, stream);
}

int datr_msda_backward_query_tiled_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                       const int64_t *level_start, const int64_t *shapes_host,
                                       const int64_t *level_start_host, const float *loc,
                                       const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                       int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                       float *grad_loc, float *grad_attn, void *stream) {
    return backward_tiled_impl(grad_out, value, shapes, level_start, shapes_host, level_start_host, loc, attn, N,
                               S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn, stream, false);
}

int datr_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                          const double *loc, const double *attn, int64_t N, int64_t S, int64_t M,
                          int64_t D, int64_t L, int64_t Lq, int64_t P, double *out, void *stream) {
    return forward_any<double>(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out,
                               stream);
}

int datr_msda_backward_f64(const double *grad_out, const double *value, const int64_t *shapes,
                           const int64_t *level_start, const double *loc, const double *attn,
                           int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq,
                           int64_t P, double *grad_value, double *grad_loc, double *grad_attn,
                           void *stream) {
    if (!dims_ok(N, S, M, D, L, Lq, P)) return DATR_EINVAL;
    if (N == 0) return DATR_OK;
    if (!value || !grad_value) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_value, 0, (size_t)(N * S * M * D) * sizeof(double), st) != hipSuccess)
        return DATR_ELAUNCH;
    if (Lq == 0) return DATR_OK;
    if (!grad_out || !shapes || !level_start || !loc || !attn || !grad_loc || !grad_attn)
        return DATR_EINVAL;
    return backward_generic<double>(grad_out, value, shapes, level_start, loc, attn, N, S, M, D, L,
                                    Lq, P, grad_value, grad_loc, grad_attn, st);
}

}  // extern "C"
