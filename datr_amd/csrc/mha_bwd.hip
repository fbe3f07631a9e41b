// mha_bwd.hip -- exact-fp32 MFMA backward of the decoder's self-attention (head_dim 32, additive
// [L, L] mask): the gradient of `nn.MultiheadAttention`'s softmax(q k^T / sqrt(d) + mask) v inside
// DeformableTransformerDecoderLayer (/root/reference/models/dino/deformable_transformer.py:880-884),
// fed with the output and log-sum-exp of mha_fwd.hip.  Replaces PyTorch's memory-efficient
// backward (two Triton-built kernels, 375 us per layer at N = 4, 8 heads, L = 1100).
//
// Per (batch, head), with P = exp(scale q k^T + mask - lse) and D_i = sum_d dO_id O_id:
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - D),   dQ = scale dS K,   dK = scale dS^T Q.
// Two launches, each recomputing the score tile it needs (7 matrix products per tile pair, no
// atomics, every output element written once => bitwise reproducible):
//   * mha_bwd_dq_d32   -- QUERY-stationary, the forward's layout: a wave holds Q^T and dO^T of its
//     32 queries in registers, S^T = K Q^T and dP^T = V dO^T come out of v_mfma_f32_32x32x2_f32 with
//     one query per lane and 16 keys in its registers, so lse / D are per-lane scalars and dS^T is,
//     as it stands, the B operand of dQ^T += K^T dS^T.  It also writes D for the second kernel.
//   * mha_bwd_dkv_d32  -- KEY-stationary mirror: a wave holds K^T (pre-scaled) and V^T of its 32 keys,
//     S = Q K^T and dP = dO V^T have one key per lane and 16 queries in registers; P and dS are the B
//     operands of dV^T += dO^T P and dK^T += Q^T dS.
// As in the forward, a workgroup is ONE 32-row tile and its four waves split the other sequence
// axis with private (barrier-free) LDS staging, next tile in flight in registers; the four partial
// accumulators are added in a fixed order at the end.
// Tensors are addressed as x[l * ld_l + n * ld_n + h * 32 + d].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int TS = 36;                         // tile row stride (floats): conflict-free ds_read_b128
constexpr int kWaveLds = 2 * 32 * TS + 64;     // two staged tiles + 64 per-row scalars, per wave

struct BwdStrides { long q_l, q_n, k_l, k_n, v_l, v_n, o_l, o_n, g_l, g_n, dq_l, dq_n, dk_l, dk_n, dv_l, dv_n; };

// one lane's share of two [32, 32] tiles: 16 floats of row `row` of each (zeros past the end)
__device__ __forceinline__ void fetch2(const float *ab, const float *bb, long a_l, long b_l, int row, int half,
                                       int L, float4 (&an)[4], float4 (&bn)[4]) {
    const int rc = min(row, L - 1);
    const float4 *ap = reinterpret_cast<const float4 *>(ab + (long)rc * a_l + half * 16);
    const float4 *bp = reinterpret_cast<const float4 *>(bb + (long)rc * b_l + half * 16);
    const float keep = row < L ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = ap[i], b = bp[i];
        an[i] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
        bn[i] = make_float4(b.x * keep, b.y * keep, b.z * keep, b.w * keep);
    }
}

__device__ __forceinline__ void stage2(float *As, float *Bs, int srow, int shalf, const float4 (&an)[4],
                                       const float4 (&bn)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4 *>(&As[srow * TS + shalf * 16 + 4 * i]) = an[i];
        *reinterpret_cast<float4 *>(&Bs[srow * TS + shalf * 16 + 4 * i]) = bn[i];
    }
}

// C[i][j] += sum_d A[i][d] * B[d][j] with A rows in LDS (row = this lane's l31) and B^T in registers
// (breg[t] = B[lhi * 16 + t][l31]); two independent accumulator chains are interleaved by the caller.
__device__ __forceinline__ void mm2_rows(const float *As, const float *Bs, int l31, int lhi,
                                         const float (&areg)[16], const float (&breg)[16], f32x16 &c0, f32x16 &c1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a4 = *reinterpret_cast<const float4 *>(&As[l31 * TS + lhi * 16 + 4 * i]);
        const float4 b4 = *reinterpret_cast<const float4 *>(&Bs[l31 * TS + lhi * 16 + 4 * i]);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, areg[4 * i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, breg[4 * i], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, areg[4 * i + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, breg[4 * i + 1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, areg[4 * i + 2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, breg[4 * i + 2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, areg[4 * i + 3], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, breg[4 * i + 3], c1, 0, 0, 0);
    }
}

__device__ __forceinline__ void load16(const float *p, float scale, float (&r)[16]) {
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t4 = p4[i];
        r[4 * i] = t4.x * scale; r[4 * i + 1] = t4.y * scale;
        r[4 * i + 2] = t4.z * scale; r[4 * i + 3] = t4.w * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// dQ (and D): a workgroup owns 32 queries of one (batch, head); wave w walks key tiles w, w + 4, ...
// ------------------------------------------------------------------------------------------------
// (three waves per SIMD: 156 registers without a spill; 1 120 workgroups then run in two rounds of 768 slots instead of
// three of 512: 115 -> 103 us per layer at N = 4)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void mha_bwd_dq_d32(
    const float *__restrict__ gout, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const float *__restrict__ out, const float *__restrict__ lse,
    const float *__restrict__ mask, float *__restrict__ delta, float *__restrict__ dq, int L, int H,
    BwdStrides st, float scale)
{
    __shared__ __attribute__((aligned(16))) float smem[kWaves * kWaveLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.y / H, h = blockIdx.y % H;
    const int qi = blockIdx.x * 32 + l31;                      // this lane's query
    const int qc = min(qi, L - 1);
    const float *kb = k + (long)n * st.k_n + h * 32;
    const float *vb = v + (long)n * st.v_n + h * 32;
    float *Ks = smem + wave * kWaveLds, *Vs = Ks + 32 * TS;

    float qreg[16], greg[16];
    load16(q + (long)qc * st.q_l + (long)n * st.q_n + h * 32 + lhi * 16, scale, qreg);
    load16(gout + (long)qc * st.g_l + (long)n * st.g_n + h * 32 + lhi * 16, 1.f, greg);
    float dsum = 0.f;
    {
        float oreg[16];
        load16(out + (long)qc * st.o_l + (long)n * st.o_n + h * 32 + lhi * 16, 1.f, oreg);
#pragma unroll
        for (int t = 0; t < 16; ++t) dsum = fmaf(greg[t], oreg[t], dsum);
        dsum += __shfl_xor(dsum, 32, 64);
    }
    const float lse_q = lse[((long)n * H + h) * L + qc];
    if (wave == 0 && lhi == 0 && qi < L) delta[((long)n * H + h) * L + qi] = dsum;

    f32x16 dqacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[e] = 0.f;
    const float *mrow = mask ? mask + (long)qc * L : nullptr;
    const bool vec_mask = (L & 3) == 0;

    const int srow = lane >> 1, shalf = lane & 1;
    float4 kn[4], vn[4];
    const int step = 32 * kWaves;
    int j0 = wave * 32;
    if (j0 < L) fetch2(kb, vb, st.k_l, st.v_l, j0 + srow, shalf, L, kn, vn);
    for (; j0 < L; j0 += step) {
        stage2(Ks, Vs, srow, shalf, kn, vn);
        if (j0 + step < L) fetch2(kb, vb, st.k_l, st.v_l, j0 + step + srow, shalf, L, kn, vn);

        // S^T = K Q^T and dP^T = V dO^T : [32 keys] x [32 queries]
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
        mm2_rows(Ks, Vs, l31, lhi, qreg, greg, sacc, pacc);

        // lane = query; register e = key (e&3) + 8 (e>>2) + 4 lhi
        float ds[16];
        const bool full = j0 + 32 <= L;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key = j0 + 8 * g + 4 * lhi;
            float4 mk = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mrow) {
                if (full && vec_mask) {
                    mk = *reinterpret_cast<const float4 *>(mrow + key);
                } else {
                    mk.x = key < L ? mrow[key] : 0.f;         mk.y = key + 1 < L ? mrow[key + 1] : 0.f;
                    mk.z = key + 2 < L ? mrow[key + 2] : 0.f; mk.w = key + 3 < L ? mrow[key + 3] : 0.f;
                }
            }
            const float m4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * g + c;
                const float p = key + c < L ? __expf(sacc[e] + m4[c] - lse_q) : 0.f;
                ds[e] = p * (pacc[e] - dsum);
            }
        }

        // dQ^T += K^T dS^T : step e pairs keys (e&3)+8(e>>2) [lanes 0-31] and +4 [lanes 32-63]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[key * TS + l31], ds[e], dqacc, 0, 0, 0);
        }
    }

    // combine the four key ranges in a fixed order
    __syncthreads();
    float *part = smem + wave * kWaveLds;                      // [16][64]
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e * 64 + lane] = dqacc[e];
    __syncthreads();
    if (wave == 0 && qi < L) {
        float o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] += smem[w * kWaveLds + e * 64 + lane];
        float *op = dq + (long)qi * st.dq_l + (long)n * st.dq_n + h * 32 + 4 * lhi;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(op + 8 * g) = make_float4(o[4 * g] * scale, o[4 * g + 1] * scale,
                                                                  o[4 * g + 2] * scale, o[4 * g + 3] * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: a workgroup owns 32 keys of one (batch, head); wave w walks query tiles w, w + 4, ...
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void mha_bwd_dkv_d32(
    const float *__restrict__ gout, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const float *__restrict__ lse, const float *__restrict__ delta,
    const float *__restrict__ mask, float *__restrict__ dk, float *__restrict__ dv, int L, int H,
    BwdStrides st, float scale)
{
    // + the workgroup's own K^T (pre-scaled) and V^T tiles: every wave multiplies by the same 32 keys, so they sit in LDS
    // once and a wave re-reads its 2 x 16 operand registers per query tile (8 ds_read_b128) instead of holding them
    // through the softmax and the second pair of products -- the 32 registers that let three waves per SIMD fit
    __shared__ __attribute__((aligned(16))) float smem[kWaves * kWaveLds + 2 * 32 * TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.y / H, h = blockIdx.y % H;
    const int ki = blockIdx.x * 32 + l31;                      // this lane's key
    const int kc = min(ki, L - 1);
    const float *qb = q + (long)n * st.q_n + h * 32;
    const float *gb = gout + (long)n * st.g_n + h * 32;
    const float *lb = lse + ((long)n * H + h) * L;
    const float *db = delta + ((long)n * H + h) * L;
    float *Qs = smem + wave * kWaveLds, *Gs = Qs + 32 * TS, *Ls = Gs + 32 * TS, *Ds = Ls + 32;

    float *Kt = smem + kWaves * kWaveLds, *Vt = Kt + 32 * TS;
    if (wave == 0) {
        float kreg0[16], vreg0[16];
        load16(k + (long)kc * st.k_l + (long)n * st.k_n + h * 32 + lhi * 16, scale, kreg0);
        load16(v + (long)kc * st.v_l + (long)n * st.v_n + h * 32 + lhi * 16, 1.f, vreg0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&Kt[l31 * TS + lhi * 16 + 4 * i]) = make_float4(kreg0[4 * i], kreg0[4 * i + 1], kreg0[4 * i + 2], kreg0[4 * i + 3]);
            *reinterpret_cast<float4 *>(&Vt[l31 * TS + lhi * 16 + 4 * i]) = make_float4(vreg0[4 * i], vreg0[4 * i + 1], vreg0[4 * i + 2], vreg0[4 * i + 3]);
        }
    }
    __syncthreads();

    f32x16 dkacc, dvacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[e] = 0.f; dvacc[e] = 0.f; }
    const float *mcol = mask ? mask + kc : nullptr;             // mask[query * L + key]

    const int srow = lane >> 1, shalf = lane & 1;
    float4 qn[4], gn[4];
    float ln = 0.f, dn = 0.f;
    const int step = 32 * kWaves;
    int i0 = wave * 32;
    for (; i0 < L; i0 += step) {
        fetch2(qb, gb, st.q_l, st.g_l, i0 + srow, shalf, L, qn, gn);
        if (lane < 32) { ln = lb[min(i0 + lane, L - 1)]; dn = db[min(i0 + lane, L - 1)]; }
        stage2(Qs, Gs, srow, shalf, qn, gn);
        if (lane < 32) { Ls[lane] = ln; Ds[lane] = dn; }

        // S = Q K^T and dP = dO V^T : [32 queries] x [32 keys]
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
        {
            float kreg[16], vreg[16];
            load16(&Kt[l31 * TS + lhi * 16], 1.f, kreg);
            load16(&Vt[l31 * TS + lhi * 16], 1.f, vreg);
            mm2_rows(Qs, Gs, l31, lhi, kreg, vreg, sacc, pacc);
        }

        // lane = key; register e = query (e&3) + 8 (e>>2) + 4 lhi
        float p[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qo = 8 * g + 4 * lhi;
            const float4 l4 = *reinterpret_cast<const float4 *>(&Ls[qo]);
            const float4 d4 = *reinterpret_cast<const float4 *>(&Ds[qo]);
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * g + c, qrow = i0 + qo + c;
                const float mk = (mcol && qrow < L) ? mcol[(long)qrow * L] : 0.f;
                p[e] = qrow < L ? __expf(sacc[e] + mk - lq[c]) : 0.f;
                ds[e] = p[e] * (pacc[e] - dq4[c]);
            }
        }

        // dV^T += dO^T P and dK^T += Q^T dS : step e pairs queries (e&3)+8(e>>2) and +4
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qrow = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[qrow * TS + l31], p[e], dvacc, 0, 0, 0);
            dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qrow * TS + l31], ds[e], dkacc, 0, 0, 0);
        }
    }

    // combine the four query ranges in a fixed order: [32][64] per wave (dK^T then dV^T registers)
    __syncthreads();
    float *part = smem + wave * kWaveLds;
#pragma unroll
    for (int e = 0; e < 16; ++e) { part[e * 64 + lane] = dkacc[e]; part[(16 + e) * 64 + lane] = dvacc[e]; }
    __syncthreads();
    if (wave < 2 && ki < L) {                                  // wave 0 finishes dK, wave 1 dV
        float o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] += smem[w * kWaveLds + (16 * wave + e) * 64 + lane];
        const float f = wave == 0 ? scale : 1.f;
        float *op = wave == 0 ? dk + (long)ki * st.dk_l + (long)n * st.dk_n + h * 32 + 4 * lhi
                              : dv + (long)ki * st.dv_l + (long)n * st.dv_n + h * 32 + 4 * lhi;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(op + 8 * g) =
                make_float4(o[4 * g] * f, o[4 * g + 1] * f, o[4 * g + 2] * f, o[4 * g + 3] * f);
    }
}

}  // namespace

extern "C" int datr_mha_backward_d32_f32(const float *grad_out, const float *q, const float *k, const float *v,
                                         const float *out, const float *lse, const float *mask, int64_t L,
                                         int64_t N, int64_t H, const int64_t *strides, float scale,
                                         float *delta, float *grad_q, float *grad_k, float *grad_v,
                                         void *stream) {
    if (L <= 0 || N <= 0 || H <= 0) return DATR_EINVAL;
    if (!grad_out || !q || !k || !v || !out || !lse || !delta || !grad_q || !grad_k || !grad_v || !strides)
        return DATR_EINVAL;
    if (L > 0x7fffff || N * H > 65535) return DATR_EUNSUPPORTED;
    for (int i = 0; i < 16; ++i)
        if (strides[i] % 4 != 0) return DATR_EUNSUPPORTED;            // float4 loads / stores
    BwdStrides st{strides[0], strides[1], strides[2],  strides[3],  strides[4],  strides[5],  strides[6],  strides[7],
                  strides[8], strides[9], strides[10], strides[11], strides[12], strides[13], strides[14], strides[15]};
    static_assert(2 * 16 * 64 <= kWaveLds, "partial accumulators must fit the staging space");
    dim3 grid((unsigned)((L + 31) / 32), (unsigned)(N * H));
    hipLaunchKernelGGL(mha_bwd_dq_d32, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, q, k, v, out, lse,
                       mask, delta, grad_q, (int)L, (int)H, st, scale);
    hipLaunchKernelGGL(mha_bwd_dkv_d32, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, q, k, v, lse,
                       delta, mask, grad_k, grad_v, (int)L, (int)H, st, scale);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
