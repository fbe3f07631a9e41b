// wino.hip -- Winograd F(2x2, 3x3) convolution on the MFMA units, exact fp32, NHWC: the 3x3 /
// stride 1 / pad 1 convolutions of DATR's image-level domain discriminator (`FCDiscriminator_img`,
// /root/reference/models/dino/DA_utils.py:61-79, applied to the four pyramid levels of all 2B
// images behind the gradient-reversal layer, /root/reference/models/dino/dino.py:351-359) -- the
// forward with bias + LeakyReLU fused, and the data gradient (the same kernel on dY with the
// transposed / flipped filter) with the LeakyReLU gate of the previous layer and the GRL sign fused.
//
//   Y = A^T [ sum_ci (G g G^T) o (B^T d B) ] A        (Lavin & Gray's minimal filtering, 2.25x fewer
//   multiplies than the direct form; MIOpen's fp32 kernel for these layers is the same algorithm on
//   the VALU, this one runs the 16 per-position channel contractions as v_mfma_f32_32x32x2_f32).
//
// Work split: a workgroup = 8 x 8 tiles (16 x 16 output pixels) x 64 output channels, 8 waves.
// Wave (wp, wa, wb) owns tiles 32 wa .. + 31 and channels 32 wb .. + 31 for the transform rows
// xi = 2 wp, 2 wp + 1 (8 of the 16 positions = 8 accumulator blocks = 128 registers; the two waves
// of a SIMD hold the two halves).  The inverse transform A^T M A is lane-local up to one exchange of
// 32 floats per lane between the two position halves (through LDS, once per workgroup), and a lane's
// 32-lane row stores 128 contiguous bytes of NHWC output.  Input channels go by in chunks of 8,
// software-pipelined with ONE barrier per chunk: while chunk c is multiplied, the raw 18 x 18 patch
// of chunk c + 1 (staged in LDS, zero outside the image) is turned into V = B^T d B -- each thread
// owns (tile, 4 channels, 2 of the 4 transform rows, 2 of the 4 columns), a piece of that work is
// issued between the MFMAs of each position pair -- the transformed filter slab of chunk c + 1
// (32 KB, [pos][k-half][cout][4] so that every ds_read_b128 group is bank-conflict free) arrives by
// LDS-DMA, the raw patch of chunk c + 2 is written to LDS and that of chunk c + 3 is in flight in
// registers.  All levels of the pyramid share the filter, so they are ONE launch (level table in the
// arguments).
//
// Measured (profiles/r02_wino.md): 256 -> 256 on the 100 x 167 level of 4 images 492 us = 160 TF/s
// direct-equivalent (MIOpen's fp32 Winograd: 785 us); the MFMA stream alone (copies, transform,
// operand reads compiled out) runs 352 us -- the rest is VALU / LDS / vector-memory ISSUE competing
// with the MFMAs for the SIMD's issue port, the same with one or two waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "datr_hip.h"

#ifndef WINO_NT_EPILOGUE
#define WINO_NT_EPILOGUE 1  // output / gate streams bypass the cache policy of the re-read patches and filter slabs (-3 % per layer)
#endif
#ifndef WINO_ABLATE
#define WINO_ABLATE 0      // development only (wrong results): 1 no copies, 2 no transform, 4 no operand reads, 8 no loop barrier
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 512;                 // 8 waves: two per SIMD share the 16 transform positions
constexpr int TB = 8;                          // tiles per workgroup edge
constexpr int PW = 2 * TB + 2;                 // patch edge (18)
constexpr int PPIX = PW * PW;                  // 324
constexpr int CK = 8;                          // input channels per chunk
constexpr int BN = 64;                         // output channels per workgroup
constexpr int kPatchF = 2 * PPIX * 4;          // floats: [half][pixel][4]
constexpr int kVF = 16 * 2 * 64 * 4;           // floats: [pos][half][tile][4]
constexpr int kBF = 16 * 2 * BN * 4;           // floats per filter slab: [pos][half][cout][4]
constexpr int kLdsBytes = (2 * kPatchF + 2 * kVF + 2 * kBF) * 4;   // 148.7 KB: everything double-buffered

struct WinoLevel { const float *x; float *y; const float *gate; int H, W, tbx, tby, first; };
struct WinoArgs { WinoLevel lv[DATR_WINO_MAX_LEVELS]; int nlevels, Cin, Cout; float slope, gate_slope, out_scale; };

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

__global__ __launch_bounds__(kThreads) void wino_conv_nhwc(WinoArgs args, const float *__restrict__ U,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                       // [2 buffers][2][324][4]
    float *Vs = patch + 2 * kPatchF;           // [2 buffers][16][2][64][4]
    float *Bs = Vs + 2 * kVF;                  // [2 buffers][16][2][64][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // wave = (wp, wa, wb): transform rows 2 wp, 2 wp + 1 (positions 8 wp .. 8 wp + 7) of tiles
    // 32 wa .. + 31 and output channels 32 wb .. + 31; waves w and w + 4 share a SIMD
    const int wp = wave >> 2, wa = (wave >> 1) & 1, wb = wave & 1;

    // ---- which level / image / tile block ------------------------------------------------------
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < DATR_WINO_MAX_LEVELS; ++i)
        if (i < args.nlevels && (int)blockIdx.x >= args.lv[i].first) lvl = i;
    const WinoLevel L = args.lv[lvl];
    int r = blockIdx.x - L.first;
    const int tbx = r % L.tbx; r /= L.tbx;
    const int tby = r % L.tby;
    const int n = r / L.tby;
    const int H = L.H, W = L.W, Cin = args.Cin, Cout = args.Cout;
    const int y0 = tby * 2 * TB, x0 = tbx * 2 * TB;
    const int co0 = blockIdx.y * BN;
    const float *Xn = L.x + (size_t)n * H * W * Cin;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // filter slab of chunk c -> Bs[buf]: per (pos, half) 64 couts x 16 B = 1 KiB contiguous in
    // U[pos][Cin/8][2][Cout][4]; 32 pieces, 4 per wave, one LDS-DMA instruction each.  (Inline asm,
    // not the builtin: the compiler would make every later ds_read wait for the DMA -- vmcnt(0)
    // right after issue; completion is awaited explicitly before the chunk's closing barrier.)
    const int nchunks = Cin / CK;
    const unsigned bs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)Bs;
    auto dma_b = [&](int c, int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int piece = wave * 4 + u;                    // = pos * 2 + half
            const int pos = piece >> 1, half = piece & 1;
            const float *src = U + ((((size_t)pos * nchunks + c) * 2 + half) * Cout + co0 + lane) * 4;
            const unsigned dst = __builtin_amdgcn_readfirstlane(bs_lds + (buf * kBF + piece * BN * 4) * 4);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory");
        }
    };
    // raw patch of chunk c: 324 pixels x 2 float4 (648 <= 2 x 512), zero outside the image;
    // register-staged so that the loads of chunk c + 3 are in flight while chunk c is multiplied
    // The pixel of each of this thread's two float4 does not change from chunk to chunk: its byte
    // offset is worked out once (0x80000000 outside the image / past the patch: a raw buffer load
    // returns zeros there, no select), the chunk only moves the scalar offset.
    float4 pf[2];
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Xn), 0, H * W * Cin * 4, 0x00020000);
    unsigned poff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + u * kThreads;
        const int pix = f >> 1, h = f & 1;
        const int py = pix / PW, px = pix - py * PW;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool in = f < 2 * PPIX && yy >= 0 && yy < H && xx >= 0 && xx < W;
        poff[u] = in ? (unsigned)(((yy * W + xx) * Cin + h * 4) * 4) : 0x80000000u;
    }
    auto fetch_patch = [&](int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const auto r_ = __builtin_amdgcn_raw_buffer_load_b128(xsrc, poff[u], c * CK * 4, 0);
            pf[u] = __builtin_bit_cast(float4, r_);
        }
    };
    auto store_patch = [&](float *dst) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = tid + u * kThreads;
            const int pix = f >> 1, h = f & 1;
            if (f < 2 * PPIX) *reinterpret_cast<float4 *>(&dst[(h * PPIX + pix) * 4]) = pf[u];
        }
    };

    // transform unit of this thread: (tile, channel half, pair of transform rows, pair of transform
    // columns); everything but the tile is wave-uniform.  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:
    // rows {0, 1} are (d0 - d2, d1 + d2), rows {2, -3} are (d2 - d1, d3 - d1) -- with the LAST row /
    // column negated (the filter transform carries the same sign, wino_weights) both pairs are
    // (A - B, C + s B) over three patch rows (A, B, C) = (0, 2, 1), s = +1 or (2, 1, 3), s = -1: the
    // rows are LOADED in that order, so no register selects and no branches in the loop.
    const int t_tile = tid & 63;
    const int t_half = wave & 1, t_rp = (wave >> 1) & 1, t_cp = wave >> 2;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const int p_base = (t_half * PPIX + 2 * t_ty * PW + 2 * t_tx) * 4;
    const int r_off[3] = {(t_rp ? 2 : 0) * PW * 4, (t_rp ? 1 : 2) * PW * 4, (t_rp ? 3 : 1) * PW * 4};
    const int c_off[3] = {(t_cp ? 2 : 0) * 4, (t_cp ? 1 : 2) * 4, (t_cp ? 3 : 1) * 4};
    const float s_r = t_rp ? -1.f : 1.f, s_c = t_cp ? -1.f : 1.f;
    constexpr int kVPos = 2 * 64 * 4;          // floats between consecutive positions of V / a filter slab
    const int v_off = ((((t_rp * 2) * 4 + t_cp * 2) * 2 + t_half) * 64 + t_tile) * 4;   // (xi, nu) = (2 rp, 2 cp)

    float4 d[3][3], tq[3];
    auto f4fma = [](float s_, float4 b, float4 c) {
        return make_float4(fmaf(s_, b.x, c.x), fmaf(s_, b.y, c.y), fmaf(s_, b.z, c.z), fmaf(s_, b.w, c.w));
    };
    auto tr_load = [&](const float *pb, int rr) {              // one of the rows A, B, C: columns A, B, C
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            d[rr][cc] = *reinterpret_cast<const float4 *>(pb + p_base + r_off[rr] + c_off[cc]);
    };
    auto tr_rows = [&](int q) {                                // t = row 2 rp + q of (+-) B^T d
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            tq[cc] = q == 0 ? f4sub(d[0][cc], d[1][cc]) : f4fma(s_r, d[1][cc], d[2][cc]);
    };
    auto tr_store = [&](float *vbuf, int q) {                  // V[xi][2 cp], V[xi][2 cp + 1]
        float *vb = vbuf + v_off + q * 4 * kVPos;
        *reinterpret_cast<float4 *>(vb) = f4sub(tq[0], tq[1]);
        *reinterpret_cast<float4 *>(vb + kVPos) = f4fma(s_c, tq[1], tq[2]);
    };
    auto tr_piece = [&](const float *pb, float *vbuf, int k) { // 7 pieces
        if (k < 3) tr_load(pb, k);
        else if (k == 3) tr_rows(0);
        else if (k == 4) tr_store(vbuf, 0);
        else if (k == 5) tr_rows(1);
        else if (k == 6) tr_store(vbuf, 1);
    };

    // ---- prologue: patches 0, 1 in LDS, patch 2 in registers, slab 0 on its way, V(0) computed ------
    fetch_patch(0);
    dma_b(0, 0);
    store_patch(patch);
    if (nchunks > 1) { fetch_patch(1); store_patch(patch + kPatchF); }
    if (nchunks > 2) fetch_patch(2);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) tr_piece(patch, Vs, k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int a_off = (wp * 8 * 2 * 64 + lhi * 64 + wa * 32 + l31) * 4;
    const int b_off = (wp * 8 * 2 * BN + lhi * BN + wb * 32 + l31) * 4;
    auto chunk = [&](int c, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const int buf = c & 1;
        // invariant: V[buf], Bs[buf] hold chunk c; patch[buf ^ 1] holds the raw chunk c + 1; the
        // registers hold the raw chunk c + 2; every wave is past its reads of chunk c - 1
        auto copies = [&]() {                  // this wave's share of the chunk's global traffic
            if (c + 2 < nchunks) store_patch(patch + buf * kPatchF);
            if (c + 3 < nchunks) fetch_patch(c + 3);
            if (more) dma_b(c + 1, buf ^ 1);
        };
        // the two waves of a SIMD issue their copies at different points of the chunk, so the
        // issue stall of one (an LDS-DMA piece costs ~100 cycles) hides behind the other's MFMAs
        if (!(WINO_ABLATE & 1) && wp == 0) copies();

        // 8 positions x (32 tiles x 32 couts) += V[pos] U[pos] over the 8 channels of chunk c, with
        // this thread's share of the transform of chunk c + 1 issued in the shadow of the MFMAs.
        // Issue order (pinned with sched_barriers): two positions per slot, their MFMAs ALTERNATE so
        // that back-to-back MFMAs never share an accumulator.
        const float *va = Vs + buf * kVF + a_off;
        const float *vbs = Bs + buf * kBF + b_off;
        const float *pnext = patch + (buf ^ 1) * kPatchF;
        float *vnext = Vs + (buf ^ 1) * kVF;
        float4 a0 = *reinterpret_cast<const float4 *>(va), b0 = *reinterpret_cast<const float4 *>(vbs);
        float4 a1 = *reinterpret_cast<const float4 *>(va + kVPos), b1 = *reinterpret_cast<const float4 *>(vbs + kVPos);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int p0 = 2 * sl, p1 = 2 * sl + 1, nx = (sl < 3 ? 2 * sl + 2 : 0);
            float4 an0, bn0, an1, bn1;
            if (!(WINO_ABLATE & 1) && sl == 2 && wp == 1) copies();
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[p0], 0, 0, 0);
            an0 = (WINO_ABLATE & 4) ? a0 : *reinterpret_cast<const float4 *>(va + nx * kVPos);
            __builtin_amdgcn_sched_barrier(0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[p1], 0, 0, 0);
            bn0 = (WINO_ABLATE & 4) ? b0 : *reinterpret_cast<const float4 *>(vbs + nx * kVPos);
            __builtin_amdgcn_sched_barrier(0);
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[p0], 0, 0, 0);
            an1 = (WINO_ABLATE & 4) ? a1 : *reinterpret_cast<const float4 *>(va + (nx + 1) * kVPos);
            __builtin_amdgcn_sched_barrier(0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[p1], 0, 0, 0);
            bn1 = (WINO_ABLATE & 4) ? b1 : *reinterpret_cast<const float4 *>(vbs + (nx + 1) * kVPos);
            __builtin_amdgcn_sched_barrier(0);
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[p0], 0, 0, 0);
            if (!(WINO_ABLATE & 2) && more) tr_piece(pnext, vnext, 2 * sl);
            __builtin_amdgcn_sched_barrier(0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[p1], 0, 0, 0);
            if (!(WINO_ABLATE & 2) && more) tr_piece(pnext, vnext, 2 * sl + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[p0], 0, 0, 0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[p1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = an0; b0 = bn0; a1 = an1; b1 = bn1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // slab c + 1 (this wave's pieces) landed
        if (!(WINO_ABLATE & 8)) __syncthreads();
    };
    // steady state without a branch per piece; the last chunk has nothing to prepare
    for (int c = 0; c + 1 < nchunks; ++c) chunk(c, std::true_type{});
    chunk(nchunks - 1, std::false_type{});

    // ---- inverse transform: lane = cout, register e = tile.  This wave holds transform rows
    // xi = 2 wp, 2 wp + 1: T[xi][j] = (M A)[xi][j], and its share of Y = A^T T is
    //   wp 0: Y[0][j] += T[0][j] + T[1][j], Y[1][j] += T[1][j];   wp 1: Y[0][j] += T[2][j], Y[1][j] += -T[2][j] - T[3][j].
    // Wave wp finishes output row i = wp of every tile: the partial of the OTHER row goes to the
    // partner wave (same wa, wb) through LDS (the filter slabs' space; 32 floats per lane).
    float mine[2][16], other[2][16];           // [j][e]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float t[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float m0 = acc[q * 4][e], m1 = acc[q * 4 + 1][e], m2 = acc[q * 4 + 2][e], m3 = acc[q * 4 + 3][e];
            t[q][0] = m0 + m1 + m2;
            t[q][1] = m1 - m2 - m3;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (wp == 0) { mine[j][e] = t[0][j] + t[1][j]; other[j][e] = t[1][j]; }
            else         { mine[j][e] = -t[0][j] - t[1][j]; other[j][e] = t[0][j]; }
        }
    }
    float *xch = Bs;                           // [8 waves][32][64]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[(wave * 32 + j * 16 + e) * 64 + lane] = other[j][e];
    __syncthreads();
    const float *got = xch + ((wave ^ 4) * 32) * 64 + lane;

    // ---- epilogue: output row i = wp of the wave's tiles, both columns ------------------------------
    const int co = co0 + wb * 32 + l31;
    const float sc = scale ? scale[co] : 1.f;
    const float sh = shift ? shift[co] : 0.f;
    const float slope = args.slope, gslope = args.gate_slope, oscale = args.out_scale;
    float *Yn = L.y + (size_t)n * H * W * Cout;
    const float *Gn = L.gate ? L.gate + (size_t)n * H * W * Cout : nullptr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tile = wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const int yy = y0 + 2 * (tile >> 3) + wp, ox = x0 + 2 * (tile & 7);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int xx = ox + j;
            if (yy < H && xx < W) {
                const size_t o = ((size_t)yy * W + xx) * Cout + co;
                float v = (mine[j][e] + got[(j * 16 + e) * 64]) * sc + sh;
                v = v > 0.f ? v : v * slope;
#if WINO_NT_EPILOGUE
                if (Gn) v = __builtin_nontemporal_load(&Gn[o]) > 0.f ? v : v * gslope;
                __builtin_nontemporal_store(v * oscale, &Yn[o]);
#else
                if (Gn) v = Gn[o] > 0.f ? v : v * gslope;
                Yn[o] = v * oscale;
#endif
            }
        }
    }
}


// U[pos][Cin/8][2][Cout][4] = G g G^T of every (ci, co) filter; flip = the data-gradient filter
// (taps mirrored; the caller swaps the channel strides)
__global__ void wino_weights(const float *__restrict__ w, long s_co, long s_ci, long s_r, long s_s, int flip,
                             int Cin, int Cout, float *__restrict__ U)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int co = idx % Cout, ci = idx / Cout;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int rr = flip ? 2 - r : r, qq = flip ? 2 - q : q;
            g[r][q] = w[co * s_co + ci * s_ci + rr * s_r + qq * s_s];
        }
    float gg[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        gg[0][q] = g[0][q];
        gg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
        gg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
        gg[3][q] = g[2][q];
    }
    const int nchunks = Cin / CK;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const float u[4] = {gg[xi][0], 0.5f * (gg[xi][0] + gg[xi][1] + gg[xi][2]),
                            0.5f * (gg[xi][0] - gg[xi][1] + gg[xi][2]), gg[xi][2]};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int pos = xi * 4 + nu;
            // row 3 / column 3 of the input transform are computed negated (wino_conv_nhwc): same sign here
            const float sg = ((xi == 3) != (nu == 3)) ? -1.f : 1.f;
            U[((((size_t)pos * nchunks + ci / CK) * 2 + (ci % CK) / 4) * Cout + co) * 4 + (ci & 3)] = sg * u[nu];
        }
    }
}

}  // namespace

extern "C" int datr_wino_weights_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci,
                                     int64_t s_r, int64_t s_s, int flip, float *u, void *stream) {
    if (!w || !u || Cin <= 0 || Cout <= 0) return DATR_EINVAL;
    if (Cin % CK != 0 || Cout % BN != 0 || Cin * Cout > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int total = (int)(Cin * Cout);
    hipLaunchKernelGGL(wino_weights, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (long)s_co,
                       (long)s_ci, (long)s_r, (long)s_s, flip, (int)Cin, (int)Cout, u);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_conv3x3_wino_nhwc_f32(const datr_wino_level *levels, int64_t nlevels, int64_t N, int64_t Cin,
                                          int64_t Cout, const float *u, const float *scale, const float *shift,
                                          float slope, float gate_slope, float out_scale, void *stream) {
    if (!levels || !u || nlevels <= 0 || N <= 0 || Cin <= 0 || Cout <= 0) return DATR_EINVAL;
    if (nlevels > DATR_WINO_MAX_LEVELS || Cin % CK != 0 || Cout % BN != 0) return DATR_EUNSUPPORTED;
    WinoArgs a;
    a.nlevels = (int)nlevels; a.Cin = (int)Cin; a.Cout = (int)Cout;
    a.slope = slope; a.gate_slope = gate_slope; a.out_scale = out_scale;
    long blocks = 0;
    for (int i = 0; i < DATR_WINO_MAX_LEVELS; ++i) {
        const datr_wino_level &s = levels[i < nlevels ? i : 0];
        if (i < nlevels) {
            if (!s.x || !s.y || s.H <= 0 || s.W <= 0) return DATR_EINVAL;
            if (N * s.H * s.W * (Cin > Cout ? Cin : Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
        }
        WinoLevel &d = a.lv[i];
        d.x = s.x; d.y = s.y; d.gate = s.gate; d.H = (int)s.H; d.W = (int)s.W;
        d.tbx = (int)((s.W + 2 * TB - 1) / (2 * TB)); d.tby = (int)((s.H + 2 * TB - 1) / (2 * TB));
        d.first = (int)blocks;
        if (i < nlevels) blocks += (long)N * d.tbx * d.tby;
    }
    if (blocks > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    static int lds_set = 0;                    // idempotent; a race only repeats the call
    if (!lds_set) {
        if (hipFuncSetAttribute((const void *)wino_conv_nhwc, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes) != hipSuccess) return DATR_ELAUNCH;
        lds_set = 1;
    }
    dim3 grid((unsigned)blocks, (unsigned)(Cout / BN));
    hipLaunchKernelGGL(wino_conv_nhwc, grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, a, u, scale, shift);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
