// gemm_f32.hip -- one exact-fp32 MFMA GEMM family with programmable epilogues, for the layers of the
// path whose arithmetic is a plain matrix product but whose NEIGHBOURS are memory passes that a
// library GEMM cannot absorb:
//   * the 1x1 convolutions of the NHWC ResNet-50 bottlenecks and of `input_proj`
//     (/root/reference/models/dino/backbone.py:62-72,109-128 around torchvision's Bottleneck,
//     /root/reference/models/dino/dino.py:111-119): conv3 + frozen BN + residual + ReLU in ONE launch
//     (epilogue scale / shift / residual / ReLU), the data gradients with the ReLU gate of the tensor
//     they produce (and the gradient of the identity branch) applied on the way out, the weight
//     gradients as a deterministic split-K product over the pixels;
//   * the FFN backward (/root/reference/models/dino/deformable_transformer.py:783-787,803-806):
//     dz = (dy W2) * [h > 0] together with the column sums of dz (linear1's bias gradient) as the
//     epilogue of the GEMM that produces dh -- no pass over the rows x 2048 hidden gradient.
//
//   C[M, N] = epi( op(A) op(B) ),  v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, exact fp32)
//   form NT:  A [M, K] row-major,  B [N, K] row-major        (y = x W^T: forward)
//   form NN:  A [M, K] row-major,  B [K, N] row-major        (dx = dy W: data gradient)
//   form TN:  A [K, M] row-major,  B [K, N] row-major        (dW = dy^T x: weight gradient, split-K)
//   epi(v)[m, n] = gate( relu( v * scale[n] + shift[n] + R[m, n] ) ),  gate(v) = G[m, n] > 0 ? v : 0,
//   every term optional; optional column sums of the result (per-workgroup partials added in a fixed
//   order by `gemm_colsum_finish`: deterministic).
//
// Workgroup = 256 threads = 2 x 2 waves, macro tile (64 TM) x (64 TN), wave tile (32 TM) x (32 TN),
// reduction in steps of 32.  Both operand tiles travel HBM/L2 -> LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: no staging registers, rows outside the matrix read zeros through the
// buffer's range check), double-buffered: the tile of step k + 1 is requested right after the barrier
// that publishes step k.  Two workgroups per CU (64 KB of LDS each at 128 x 128) cover each other's
// barrier.
//   * operand with the reduction axis contiguous ("KC", [rows][32]): a row is one 128-B line, eight
//     16-B slots, stored XOR-swizzled (slot ^ ((row >> 1) & 7)) by permuting the SOURCE address of the
//     DMA (its LDS side is lane-linear); an MFMA lane (row i, half h) takes FOUR consecutive k with one
//     ds_read_b128 (slot 2 g + h of k-group g) -- conflict-free in the instruction's four 16-lane
//     groups -- and feeds four MFMAs: the k pairs are (8 g + q, 8 g + 4 + q), q = 0..3;
//   * operand with the reduction axis strided ("NC", [32][cols]): 32 lanes read 32 consecutive
//     columns of k-row 8 g + 4 h + q with ds_read_b32 (the same k pairing).
// Workgroup b runs on XCD b % 8: an XCD gets a contiguous range of tiles, column tile fastest, so the
// workgroups that share an A tile share an L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// EXPERIMENTAL inner product (template parameter SP = 6, off by default: DATR_GEMM_SPLIT_BF16=1): the fp32
// operands are split EXACTLY into three bf16 pieces each -- a = hi + mid + lo, 24 mantissa bits = 3 x 8, every
// residual a - bf16(a) is exact in fp32 -- and the six largest of the nine piece products are accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16 (of the three dropped ones mid x lo and lo x mid are ~2^-26..2^-24 of the product, lo x lo ~2^-32; measured error against
// float64 is BELOW the fp32 MFMA chain's, tools/probes/split_bf16/).  The bf16 pipe does 16x the multiply-adds
// per cycle of v_mfma_f32_32x32x2_f32: six products leave 2.67x the fp32 rate if the split's ~5.5 VALU
// operations per element stay out of the way.
struct Split3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)(r1 - (float)m);
    }
    return s;
}
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) void lds_void;

#ifndef GEMM_EPI_AUX
#define GEMM_EPI_AUX 2          // epilogue streams (gate / residual in, C out) are non-temporal: they are touched once
                                // per launch and must not evict the A / B tiles other workgroups re-read from L2
                                // (measured: -2 % per epilogue launch, -0.25 ms per training step; 0 = default policy)
#endif
constexpr int kThreads = 256;
constexpr int BK = 32;
constexpr unsigned kOutOfRange = 0x80000000u;

struct GemmArgs {
    const float *A, *B;
    float *C;                       // output, or the partial buffer [ksplit][M][N] when ksplit > 1
    const float *scale, *shift, *R, *G;
    float *colpart;                 // [2 ntm][N] partial column sums (one row per row-wave), or null
    float *asum;                    // TN form: [ksplit][M] sums of A over this split's reduction range, or null
    int M, N, K;
    int lda, ldb, ldc, ldr, ldg;
    int relu;
    int ntm, ntn;
    int ksplit, kchunk;             // reduction range of blockIdx.y: [y kchunk, min(K, (y + 1) kchunk))
    unsigned a_bytes, b_bytes;      // buffer extents (range check = zero fill)
};

// AL / BL: 0 = reduction axis contiguous (KC), 1 = reduction axis strided (NC); BKT = reduction step
// (32: 64 KB of LDS at 128 x 128, two workgroups per CU; 16: 32 KB, four per CU)
template <int AL, int BL, int TM, int TN, int BKT, int SP = 0>
__global__ __launch_bounds__(kThreads, BKT == 16 ? 4 : 2) void gemm_f32_kernel(const GemmArgs a)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_BYTES = BM * BKT * 4, B_BYTES = BN * BKT * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / 4, B_INSTR = B_BYTES / 1024 / 4;   // LDS-DMA instructions per wave
    constexpr int SPR = BKT / 4;                   // 16-B slots per KC row
    constexpr int RPI = 64 / SPR;                  // KC rows per LDS-DMA instruction (1 KB)
    constexpr int SWZ = BKT == 32 ? 1 : 2;         // KC swizzle: slot ^ ((row >> SWZ) & (SPR - 1))
    constexpr int NG = BKT / 8;                    // k-groups of 8 per step
    static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for four waves of LDS-DMA");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // tile of this workgroup: XCD x = b % 8 owns the tiles [x per + min(x, rem), ...)
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7;
    const int lin = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
    const int tile_m = lin / a.ntn, tile_n = lin - tile_m * a.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = (int)blockIdx.y * a.kchunk;
    const int k_end = min(a.K, k_begin + a.kchunk);
    const int nk = (k_end - k_begin + BKT - 1) / BKT;

    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.A), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.B), 0, a.b_bytes, 0x00020000);

    // ---- LDS-DMA source offsets (bytes), one per instruction of this wave; advanced every step ------
    unsigned aoff[A_INSTR], boff[B_INSTR];
    unsigned astep, bstep;
    if (AL == 0) {          // [BM rows][BKT]: an instruction = RPI rows, slot = (lane % SPR) ^ swizzle(row)
#pragma unroll
        for (int u = 0; u < A_INSTR; ++u) {
            const int r = (wave * A_INSTR + u) * RPI + lane / SPR;
            const int s = (lane % SPR) ^ ((r >> SWZ) & (SPR - 1));
            aoff[u] = (m0 + r) < a.M ? ((unsigned)(m0 + r) * (unsigned)a.lda + (unsigned)(k_begin + s * 4)) * 4u : kOutOfRange;
        }
        astep = BKT * 4;
    } else {                // [BKT k-rows][BM]: an instruction = 1024 / (4 BM) rows of BM floats
        constexpr int LPR = BM / 4;                 // lanes per k-row
#pragma unroll
        for (int u = 0; u < A_INSTR; ++u) {
            const int e = (wave * A_INSTR + u) * 64 + lane;
            const int kr = e / LPR, col = (e % LPR) * 4;
            aoff[u] = (m0 + col) < a.M ? ((unsigned)(k_begin + kr) * (unsigned)a.lda + (unsigned)(m0 + col)) * 4u : kOutOfRange;
        }
        astep = (unsigned)a.lda * BKT * 4u;
    }
    if (BL == 0) {
#pragma unroll
        for (int u = 0; u < B_INSTR; ++u) {
            const int r = (wave * B_INSTR + u) * RPI + lane / SPR;
            const int s = (lane % SPR) ^ ((r >> SWZ) & (SPR - 1));
            boff[u] = (n0 + r) < a.N ? ((unsigned)(n0 + r) * (unsigned)a.ldb + (unsigned)(k_begin + s * 4)) * 4u : kOutOfRange;
        }
        bstep = BKT * 4;
    } else {
        constexpr int LPR = BN / 4;
#pragma unroll
        for (int u = 0; u < B_INSTR; ++u) {
            const int e = (wave * B_INSTR + u) * 64 + lane;
            const int kr = e / LPR, col = (e % LPR) * 4;
            boff[u] = (n0 + col) < a.N ? ((unsigned)(k_begin + kr) * (unsigned)a.ldb + (unsigned)(n0 + col)) * 4u : kOutOfRange;
        }
        bstep = (unsigned)a.ldb * BKT * 4u;
    }
    // Dynamic LDS starts at address 0 (no static LDS in this kernel); addressed through integers so the
    // compiler does not order every fragment read behind the DMA in flight.
    auto issue = [&](int buf) {
        const unsigned base = (unsigned)buf * STAGE;
#pragma unroll
        for (int u = 0; u < A_INSTR; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, reinterpret_cast<lds_void *>((uintptr_t)(base + (unsigned)(wave * A_INSTR + u) * 1024u)),
                                                     16, (int)aoff[u], 0, 0, 0);
            aoff[u] += astep;
        }
#pragma unroll
        for (int u = 0; u < B_INSTR; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(br, reinterpret_cast<lds_void *>((uintptr_t)(base + A_BYTES + (unsigned)(wave * B_INSTR + u) * 1024u)),
                                                     16, (int)boff[u], 0, 0, 0);
            boff[u] += bstep;
        }
    };

    // ---- fragment read offsets (bytes, relative to the stage) --------------------------------------
    // KC: row (wave base + 32 t + i), slot (2 g + lhi) ^ swizzle(i); NC: k-row 8 g + 4 lhi + q, column base + l31
    unsigned ard[TM], brd[TN];
    unsigned kslot[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) kslot[g] = (unsigned)(((2 * g + lhi) ^ ((l31 >> SWZ) & (SPR - 1))) * 16);
#pragma unroll
    for (int t = 0; t < TM; ++t)
        ard[t] = AL == 0 ? (unsigned)((wm * TM * 32 + t * 32 + l31) * BKT * 4)
                         : (unsigned)((4 * lhi * BM + wm * TM * 32 + t * 32 + l31) * 4);
#pragma unroll
    for (int t = 0; t < TN; ++t)
        brd[t] = (unsigned)A_BYTES + (BL == 0 ? (unsigned)((wn * TN * 32 + t * 32 + l31) * BKT * 4)
                                              : (unsigned)((4 * lhi * BN + wn * TN * 32 + t * 32 + l31) * 4));

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // TN form: the sums of A over the reduction axis (= the bias gradient when A is dy) fall out of the A
    // fragments: a lane holds A[k][m] of its 32 output rows m for half of the k of every step
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;
    const bool want_asum = AL == 1 && a.asum != nullptr && wn == 0 && tile_n == 0;

    // ---- epilogue operands: descriptors, and for small tiles the residual OR gate values themselves ----
    // Everything of the epilogue goes through buffer descriptors whose extent ends behind row M - 1: rows
    // past the matrix drop out by the range check, a column past N by an out-of-range lane offset.
    // All workgroups of a CU start together and run in phase, so an epilogue that WAITS for its loads
    // leaves the matrix pipes idle chip-wide (measured: +75..125 us on the 88 892 x 2048 FFN gradient):
    // with 1 or 2 accumulator tiles per wave the one extra operand is requested HERE, before the main
    // loop, and has landed long before the epilogue needs it (16 / 32 registers).
    const bool raw = a.ksplit > 1;
    const unsigned c_bytes = (unsigned)a.M * (unsigned)a.ldc * 4u;
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(
        a.C + (raw ? (size_t)blockIdx.y * (size_t)a.M * (size_t)a.ldc : 0), 0, c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.R ? a.R : a.A), 0, a.R && !raw ? (unsigned)a.M * (unsigned)a.ldr * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.G ? a.G : a.A), 0, a.G && !raw ? (unsigned)a.M * (unsigned)a.ldg * 4u : 0u, 0x00020000);
    const bool has_r = a.R && !raw, has_g = a.G && !raw, want_cs = a.colpart && !raw;
    unsigned colb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        colb[j] = n < a.N ? (unsigned)n * 4u : kOutOfRange;
    }
    constexpr bool CAN_PRE = TM * TN <= 2;
    const bool pre_r = CAN_PRE && has_r && !has_g, pre_g = CAN_PRE && has_g && !has_r;
    float pv[CAN_PRE ? TM * TN * 16 : 1];
    if (CAN_PRE && (pre_r || pre_g)) {
        const __amdgpu_buffer_rsrc_t pr = pre_r ? rr : gr;
        const unsigned pld = (unsigned)(pre_r ? a.ldr : a.ldg) * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lhi;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pv[CAN_PRE ? (j * TM + i) * 16 + e : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        pr, (int)(colb[j] + (unsigned)(mb + (e & 3) + 8 * (e >> 2)) * pld), 0, GEMM_EPI_AUX));
            }
    }

    if (nk > 0) issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        // the tile of step kt has landed (this wave's share), everybody's share after the barrier; the
        // other buffer's readers (step kt - 1) are past it too.  NOT __syncthreads(): raw barrier.
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) issue((kt + 1) & 1);
        const unsigned sb = (unsigned)(kt & 1) * STAGE;
        if constexpr (SP != 0) {
#pragma unroll
        for (int h2 = 0; h2 < NG / 2; ++h2) {          // one bf16 k-step of 16 = two groups of 8
            Split3 fa[TM], fb[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                float x[8];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * h2 + gg;
                    if (AL == 0) {
                        const f4 v = *reinterpret_cast<const lds_f4 *>((uintptr_t)(sb + ard[t] + kslot[g]));
                        x[4 * gg] = v.x; x[4 * gg + 1] = v.y; x[4 * gg + 2] = v.z; x[4 * gg + 3] = v.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            x[4 * gg + q] = *reinterpret_cast<const lds_f *>((uintptr_t)(sb + ard[t] + (unsigned)((8 * g + q) * BM * 4)));
                    }
                }
                if (AL == 1 && want_asum)
                    asum[t] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                fa[t] = split3(x);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                float x[8];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * h2 + gg;
                    if (BL == 0) {
                        const f4 v = *reinterpret_cast<const lds_f4 *>((uintptr_t)(sb + brd[t] + kslot[g]));
                        x[4 * gg] = v.x; x[4 * gg + 1] = v.y; x[4 * gg + 2] = v.z; x[4 * gg + 3] = v.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            x[4 * gg + q] = *reinterpret_cast<const lds_f *>((uintptr_t)(sb + brd[t] + (unsigned)((8 * g + q) * BN * 4)));
                    }
                }
                fb[t] = split3(x);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {             // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].lo, fb[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].lo, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].mid, fb[j].mid, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].mid, fb[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].mid, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].hi, acc[i][j], 0, 0, 0);
                }
        }
        } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float av[TM][4], bv[TN][4];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (AL == 0) {
                    const f4 v = *reinterpret_cast<const lds_f4 *>((uintptr_t)(sb + ard[t] + kslot[g]));
                    av[t][0] = v.x; av[t][1] = v.y; av[t][2] = v.z; av[t][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        av[t][q] = *reinterpret_cast<const lds_f *>((uintptr_t)(sb + ard[t] + (unsigned)((8 * g + q) * BM * 4)));
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (BL == 0) {
                    const f4 v = *reinterpret_cast<const lds_f4 *>((uintptr_t)(sb + brd[t] + kslot[g]));
                    bv[t][0] = v.x; bv[t][1] = v.y; bv[t][2] = v.z; bv[t][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bv[t][q] = *reinterpret_cast<const lds_f *>((uintptr_t)(sb + brd[t] + (unsigned)((8 * g + q) * BN * 4)));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
            if (AL == 1 && want_asum) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asum[i] += (av[i][0] + av[i][1]) + (av[i][2] + av[i][3]);
            }
        }
        }
    }
    if (AL == 1 && want_asum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float sum = asum[i] + __shfl_xor(asum[i], 32);
            const int m = m0 + wm * TM * 32 + i * 32 + l31;
            if (lhi == 0 && m < a.M) a.asum[(size_t)blockIdx.y * a.M + m] = sum;
        }
    }

    // ---- epilogue: 32 lanes = 32 consecutive columns (128 B) of one row; branch-free ---------------
    // An absent residual / gate reads as 0; the loads of an accumulator tile that were not requested
    // before the main loop are issued back to back, no per-element branches.
    const float relu_floor = (a.relu && !raw) ? 0.f : -__builtin_inff();
    const float gate_thr = (a.G && !raw) ? 0.f : -1.f;          // absent gate reads 0 > -1: pass
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        float sc = 1.f, sh = 0.f;
        if (!raw && n < a.N) {
            if (a.scale) sc = a.scale[n];
            if (a.shift) sh = a.shift[n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lhi;
            // EB elements of the accumulator tile at a time (the four-per-CU variant has 128 registers)
            constexpr int EB = (BKT == 16 && TM * TN == 4) ? 4 : 16;
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += EB) {
                float rv[EB], gv[EB];
                if (has_r && !pre_r) {
#pragma unroll
                    for (int e = 0; e < EB; ++e)
                        rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rr, (int)(colb[j] + (unsigned)(mb + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2)) * (unsigned)a.ldr * 4u), 0, GEMM_EPI_AUX));
                } else {
#pragma unroll
                    for (int e = 0; e < EB; ++e) rv[e] = pre_r ? pv[CAN_PRE ? (j * TM + i) * 16 + e0 + e : 0] : 0.f;
                }
                if (has_g && !pre_g) {
#pragma unroll
                    for (int e = 0; e < EB; ++e)
                        gv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            gr, (int)(colb[j] + (unsigned)(mb + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2)) * (unsigned)a.ldg * 4u), 0, GEMM_EPI_AUX));
                } else {
#pragma unroll
                    for (int e = 0; e < EB; ++e) gv[e] = pre_g ? pv[CAN_PRE ? (j * TM + i) * 16 + e0 + e : 0] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    const int m = mb + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
                    float v = acc[i][j][e0 + e] * sc + sh + rv[e];
                    // NaN travels as through torch.relu / threshold_backward (fmaxf would return the floor for a NaN
                    // accumulator, `gate > thr` would drop the gradient under a NaN gate): the loss guard of the
                    // training loop (engine.py:81-84) must see a diverged run
                    v = v < relu_floor ? relu_floor : v;
                    v = gv[e] <= gate_thr ? 0.f : v;
                    if (want_cs) cs[j] += m < a.M ? v : 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), cr,
                                                          (int)(colb[j] + (unsigned)m * (unsigned)a.ldc * 4u), 0, GEMM_EPI_AUX);
                }
            }
        }
    }
    if (want_cs) {
        // column sums of this wave's rows: the lane halves by one cross-lane add; one partial row per
        // (row tile, row-wave), no barrier -- `gemm_colsum_finish` adds the 2 ntm rows in a fixed order
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float sum = cs[j] + __shfl_xor(cs[j], 32);
            const int n = n0 + wn * TN * 32 + j * 32 + l31;
            if (lhi == 0 && n < a.N) a.colpart[(size_t)(tile_m * 2 + wm) * a.N + n] = sum;
        }
    }
}

// C[m, n] = rowscale[m] * sum_z partial[z][m, n]  (fixed order: deterministic); N % 4 == 0.
// A workgroup = 64 float4 columns x 4 slices of z: a thread adds every fourth partial (loads in
// flight four deep), the four slices meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void gemm_splitk_fold(const float *__restrict__ partial, int ksplit, long mn, int N,
                                                        const float *__restrict__ rowscale, float *__restrict__ C, int ldc,
                                                        int fold_blocks, const float *__restrict__ asum_part, int M,
                                                        float *__restrict__ asum_out)
{
    __shared__ float4 red[3][64];
    if ((int)blockIdx.x >= fold_blocks) {
        // the blocks behind the product's: asum_out[m] = sum_z asum_part[z][m] (the bias gradient), fixed order
        const int m = ((int)blockIdx.x - fold_blocks) * 256 + threadIdx.x;
        if (m < M) {
            float s0 = 0.f, s1 = 0.f;
            int z = 0;
            for (; z + 1 < ksplit; z += 2) { s0 += asum_part[(size_t)z * M + m]; s1 += asum_part[(size_t)(z + 1) * M + m]; }
            if (z < ksplit) s0 += asum_part[(size_t)z * M + m];
            asum_out[m] = s0 + s1;
        }
        return;
    }
    const int c = threadIdx.x & 63, zs = threadIdx.x >> 6;
    const long i4 = (long)blockIdx.x * 64 + c;
    const bool in = i4 * 4 < mn;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        const float *p = partial + i4 * 4;
        int z = zs;
        for (; z + 12 < ksplit; z += 16) {
            const float4 v0 = *reinterpret_cast<const float4 *>(p + (size_t)z * mn);
            const float4 v1 = *reinterpret_cast<const float4 *>(p + (size_t)(z + 4) * mn);
            const float4 v2 = *reinterpret_cast<const float4 *>(p + (size_t)(z + 8) * mn);
            const float4 v3 = *reinterpret_cast<const float4 *>(p + (size_t)(z + 12) * mn);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; z < ksplit; z += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)z * mn);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (zs > 0) red[zs - 1][c] = s;
    __syncthreads();
    if (zs == 0 && in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float4 v = red[k][c]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        const long m = (i4 * 4) / N;
        const int n = (int)((i4 * 4) - m * N);
        if (rowscale) { const float r = rowscale[m]; s.x *= r; s.y *= r; s.z *= r; s.w *= r; }
        *reinterpret_cast<float4 *>(C + (size_t)m * ldc + n) = s;
    }
}

// out[n] = sum_t part[t][n]: a workgroup = 16 float4 columns x 64 row slices, slices met in LDS in a
// fixed order (deterministic)
__global__ __launch_bounds__(1024) void gemm_colsum_finish(const float *__restrict__ part, int ntm, int N, float *__restrict__ out)
{
    __shared__ float4 red[64][16];
    const int c = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int n = ((int)blockIdx.x * 16 + c) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        int t = rs;
        for (; t + 192 < ntm; t += 256) {
            const float4 v0 = *reinterpret_cast<const float4 *>(part + (size_t)t * N + n);
            const float4 v1 = *reinterpret_cast<const float4 *>(part + (size_t)(t + 64) * N + n);
            const float4 v2 = *reinterpret_cast<const float4 *>(part + (size_t)(t + 128) * N + n);
            const float4 v3 = *reinterpret_cast<const float4 *>(part + (size_t)(t + 192) * N + n);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; t < ntm; t += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)t * N + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[rs][c] = s;
    __syncthreads();
    // 64 slices -> 1: threads 0..63 each add 16 of the 64 x 16 x 4 floats column-wise in a fixed order
    if (threadIdx.x < 64) {
        const int cc = threadIdx.x >> 2, k = threadIdx.x & 3;          // float4 column, component
        float acc = 0.f;
        for (int r = 0; r < 64; ++r) acc += reinterpret_cast<const float *>(&red[r][cc])[k];
        const int nn = ((int)blockIdx.x * 16 + cc) * 4 + k;
        if (nn < N) out[nn] = acc;
    }
}

struct Plan { int tm, tn, bk, ksplit, split; };

// Tile and split choice, from the sweeps of tools/sweep_gemm.py / tools/probes/gemm_tn_sweep.py at the
// step's shapes (profiles/r04_gemm_sweep.txt): SMALL tiles with many workgroups per CU win on this
// kernel -- 64 x 128 while that gives >= 4 workgroups per CU, else 64 x 64 (four resident workgroups
// cover each other's prologue, barrier and epilogue; 128 x 128 at two per CU is 5-8 % slower on the
// 88 892-row FFN shapes and 15-30 % slower on the 16 800-pixel maps).  The weight-gradient form splits
// the reduction until the launch has ~2 workgroups per CU (more: the partial-sum traffic shows).
Plan pick_plan(int form, long M, long N, long K)
{
    Plan p{1, 2, 16, 1, 0};
    if (const char *e = getenv("DATR_GEMM_SPLIT_BF16"))        // experimental inner product, off by default (see Split3)
        p.split = atoi(e) != 0 ? 6 : 0;
    auto tiles = [&](const Plan &q) { return ((M + 64 * q.tm - 1) / (64 * q.tm)) * ((N + 64 * q.tn - 1) / (64 * q.tn)); };
    if (form == 2) {
        if (N <= 64) p.tn = 1;
        const long t = tiles(p);
        const long steps = (K + BK - 1) / BK;
        long ks = std::max(1L, 512 / t);
        ks = std::max(1L, std::min(ks, steps / 2 > 0 ? steps / 2 : 1L));
        p.ksplit = (int)std::min(ks, 1024L);
    } else {
        if (N <= 64) { p.tm = 2; p.tn = 1; p.bk = 32; }        // one column tile: tall tiles, full 128-B k-rows
        else if (tiles(p) < 1024) p.tn = 1;
        else {
            // Round quantisation: 64 x 128 tiles run four workgroups per CU = 1 024 slots.  A launch of 2 104 tiles (the
            // 16 800-pixel maps x 1 024 channels) pays a third round for 5 % of its tiles; as 64 x 64 tiles (five per CU,
            // 1 280 slots, 4 208 tiles) it makes 3.3 rounds of half-size tiles: 86 against 102 us (profiles/
            // r04_gemm_sweep.txt, l3.conv3 nn).  Only for a few rounds: beyond, the larger tile's efficiency wins.
            static const bool rounds_rule = !(getenv("DATR_GEMM_ROUNDS_RULE") && atoi(getenv("DATR_GEMM_ROUNDS_RULE")) == 0);
            const double r = (double)tiles(p) / 1024.0, frac = r - (double)(long)r;
            if (rounds_rule && r < 3.5 && frac > 0.0 && frac <= 0.4) p.tn = 1;
        }
    }
    if (p.split) {
        // the split costs VALU per FRAGMENT, the products pay per fragment PAIR: 128 x 128 tiles (two fragments
        // of each operand per wave) where the launch still fills the chip (FFN shapes: 824 -> 730 us)
        const Plan big{2, 2, 16, p.ksplit, p.split};
        if (tiles(big) * (form == 2 ? p.ksplit : 1) >= 1024 && N > 64) { p.tm = 2; p.tn = 2; }
    }
    if (const char *f = getenv("DATR_GEMM_PLAN")) {            // development: "tm,tn,bk,ksplit" (ksplit 0 = keep)
        int tm = 0, tn = 0, bk = 0, ks = 0;
        if (sscanf(f, "%d,%d,%d,%d", &tm, &tn, &bk, &ks) == 4 && (tm == 1 || tm == 2) && (tn == 1 || tn == 2) &&
            (bk == 16 || bk == 32)) {
            p.tm = tm; p.tn = tn; p.bk = bk;
            if (form == 2 && ks >= 1) p.ksplit = ks;
        }
    }
    return p;
}

template <int AL, int BL>
int launch_form(const GemmArgs &a, const Plan &p, hipStream_t st)
{
    dim3 grid((unsigned)(a.ntm * a.ntn), (unsigned)a.ksplit);
    auto go = [&](auto kernel, int tm, int tn, int bk) {
        const size_t lds = (size_t)2 * (64 * tm + 64 * tn) * bk * 4;
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess;
        if (!ok) return false;
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, a);
        return true;
    };
    bool ok;
#define DATR_GO(TM_, TN_, BK_) ok = p.split ? go(gemm_f32_kernel<AL, BL, TM_, TN_, BK_, 6>, TM_, TN_, BK_) \
                                        : go(gemm_f32_kernel<AL, BL, TM_, TN_, BK_, 0>, TM_, TN_, BK_)
    if (p.bk == 32) {
        if (p.tm == 2 && p.tn == 2) DATR_GO(2, 2, 32);
        else if (p.tm == 2) DATR_GO(2, 1, 32);
        else if (p.tn == 2) DATR_GO(1, 2, 32);
        else DATR_GO(1, 1, 32);
    } else {
        if (p.tm == 2 && p.tn == 2) DATR_GO(2, 2, 16);
        else if (p.tm == 2) DATR_GO(2, 1, 16);
        else if (p.tn == 2) DATR_GO(1, 2, 16);
        else DATR_GO(1, 1, 16);
    }
#undef DATR_GO
    return ok ? DATR_OK : DATR_EUNSUPPORTED;
}

}  // namespace

extern "C" int64_t datr_gemm_workspace_floats(int form, int64_t M, int64_t N, int64_t K, int want_colsum)
{
    if (M <= 0 || N <= 0 || K <= 0 || form < 0 || form > 2) return 0;
    const Plan p = pick_plan(form, M, N, K);
    int64_t w = 0;
    if (p.ksplit > 1) w += (int64_t)p.ksplit * M * N;
    if (want_colsum) w += 2 * ((M + 64 * p.tm - 1) / (64 * p.tm)) * N;
    if (form == 2) w += (int64_t)p.ksplit * M;             // sums of A per split (epilogue.rowsum_a)
    return w;
}

extern "C" int datr_gemm_f32(int form, const float *A, int64_t lda, const float *B, int64_t ldb,
                             int64_t M, int64_t N, int64_t K, const datr_gemm_epilogue *epi,
                             float *C, int64_t ldc, float *workspace, int64_t workspace_floats, void *stream)
{
    if (!A || !B || !C || form < 0 || form > 2 || M < 0 || N < 0 || K < 0) return DATR_EINVAL;
    if (M == 0 || N == 0) return DATR_OK;
    if (K == 0) return DATR_EINVAL;
    const bool al = form == 2, bl = form != 0;
    // rows of a KC operand are 128-B lines of 32 floats; LDS-DMA moves 16 B
    if ((!al && (K % BK || lda % 4)) || (al && lda % 4) || (!bl && (K % BK || ldb % 4)) || (bl && ldb % 4) || N % 4 || (al && M % 4)) return DATR_EUNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return DATR_EUNSUPPORTED;
    const int64_t a_rows = al ? K : M, b_rows = bl ? K : N;
    const int64_t a_bytes = ((a_rows - 1) * lda + (al ? M : K)) * 4, b_bytes = ((b_rows - 1) * ldb + (bl ? N : K)) * 4;
    if (a_bytes >= (int64_t)kOutOfRange || b_bytes >= (int64_t)kOutOfRange || M * N >= (int64_t)1 << 29 || M * ldc >= (int64_t)1 << 29) return DATR_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const Plan p = pick_plan(form, M, N, K);
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc;
    a.ntm = (int)((M + 64 * p.tm - 1) / (64 * p.tm)); a.ntn = (int)((N + 64 * p.tn - 1) / (64 * p.tn));
    a.ksplit = p.ksplit;
    const int steps = (int)((K + BK - 1) / BK);
    a.kchunk = ((steps + p.ksplit - 1) / p.ksplit) * BK;
    a.ksplit = (int)((K + a.kchunk - 1) / a.kchunk);
    a.a_bytes = (unsigned)a_bytes; a.b_bytes = (unsigned)b_bytes;
    float *ws = workspace;
    int64_t left = workspace_floats;
    const float *rowscale = nullptr;
    float *colsum = nullptr;
    if (epi) {
        if (form == 2) rowscale = epi->scale;           // weight gradient: per-ROW scale (folded frozen BN)
        else { a.scale = epi->scale; a.shift = epi->shift; }
        a.R = epi->residual; a.ldr = (int)epi->ldr;
        a.G = epi->gate; a.ldg = (int)epi->ldg;
        a.relu = epi->relu;
        colsum = epi->colsum;
        if ((a.R && M * epi->ldr >= (int64_t)1 << 29) || (a.G && M * epi->ldg >= (int64_t)1 << 29)) return DATR_EUNSUPPORTED;
    }
    if (form == 2 && (a.R || a.G || a.relu || colsum || (epi && epi->shift))) return DATR_EUNSUPPORTED;
    float *rowsum_a = epi ? epi->rowsum_a : nullptr;
    if (rowsum_a && form != 2) return DATR_EUNSUPPORTED;
    float *partial = nullptr;
    if (a.ksplit > 1 || rowscale) {
        const int64_t need = (int64_t)a.ksplit * M * N;
        if (!ws || left < need) return DATR_EINVAL;
        partial = ws; ws += need; left -= need;
    }
    if (colsum) {
        const int64_t need = (int64_t)2 * a.ntm * N;
        if (!ws || left < need) return DATR_EINVAL;
        a.colpart = ws; ws += need; left -= need;
    }
    if (rowsum_a) {
        if (a.ksplit > 1) {
            const int64_t need = (int64_t)a.ksplit * M;
            if (!ws || left < need) return DATR_EINVAL;
            a.asum = ws; ws += need; left -= need;
        } else {
            a.asum = rowsum_a;
        }
    }
    auto launch = [&](const GemmArgs &g) {
        return form == 2 ? launch_form<1, 1>(g, p, st) : form == 1 ? launch_form<0, 1>(g, p, st) : launch_form<0, 0>(g, p, st);
    };
    if (partial) {
        // split reduction (raw partial stores, the kernel tests ksplit > 1) or a single range whose
        // product needs the row scale: either way the product goes to the dense [z][M][N] workspace
        // and `gemm_splitk_fold` writes C
        GemmArgs g = a;
        g.C = partial; g.ldc = (int)N;
        const int rc = launch(g);
        if (rc != DATR_OK) return rc;
        const long mn = (long)M * N;
        const int fold_blocks = (int)((mn / 4 + 63) / 64);
        const bool fold_asum = rowsum_a && a.ksplit > 1;
        hipLaunchKernelGGL(gemm_splitk_fold, dim3((unsigned)(fold_blocks + (fold_asum ? (M + 255) / 256 : 0))), dim3(256), 0, st,
                           partial, a.ksplit, mn, (int)N, rowscale, C, (int)ldc, fold_blocks,
                           fold_asum ? a.asum : nullptr, (int)M, rowsum_a);
    } else {
        const int rc = launch(a);
        if (rc != DATR_OK) return rc;
    }
    if (colsum)
        hipLaunchKernelGGL(gemm_colsum_finish, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, st, a.colpart, 2 * a.ntm,
                           (int)N, colsum);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
