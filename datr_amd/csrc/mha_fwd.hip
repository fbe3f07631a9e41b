// mha_fwd.hip -- exact-fp32 MFMA scaled-dot-product attention forward for head_dim = 32 with an
// additive [L, L] mask: the decoder's self-attention (`nn.MultiheadAttention` inside
// DeformableTransformerDecoderLayer, /root/reference/models/dino/deformable_transformer.py:880-884;
// 900 matching + 200 de-noising queries, 8 heads, the DN attention mask of dn_components.py:117-124).
//
// Forward only: 86 us for the merged source + target decoder call (N = 4, 8 heads, L = 1100;
// 58 TF/s) against 176 us for PyTorch's memory-efficient kernel -- the only stock backend that
// takes fp32 with a mask -- and the output lands directly in the [L, N, E] layout the output
// projection reads.  The backward stays PyTorch's (fed with this kernel's output and
// log-sum-exp, datr_amd/fused.py::_AttentionD32).
//
// Layout trick: a wave owns 32 queries and computes the TRANSPOSED score tile S^T = K Q^T with
// v_mfma_f32_32x32x2_f32 (A = 32 keys x d from LDS, B = Q^T held in registers for the whole
// kernel).  In the MFMA result layout a lane then holds ONE query (its column) and 16 keys (its
// registers, the other 16 sit in lane ^ 32): the row maximum / sum of the online softmax are
// in-lane reductions plus one cross-half exchange, and the probabilities are, as they stand, the
// B operand of O^T += V^T P^T (register e of both halves = the key pair of MFMA step e) -- no
// shuffle, no LDS round trip between the two matrix products; O^T has the query in the lane
// again, so rescaling by exp(m_old - m_new) is a per-lane multiply.
// Tensors are addressed as x[l * ld_l + n * ld_n + h * 32 + d] (sequence-first [L, N, H*32]
// buffers and column slices of merged projections alike).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;                 // 4 waves share ONE tile of 32 queries, keys split 4 ways
constexpr int kWaves = 4;
constexpr int KS = 36;                        // K tile row stride (floats)
constexpr int VS = 40;                        // V tile row stride
constexpr int kWaveLds = 32 * KS + 32 * VS;   // floats of private staging per wave

struct Strides { long q_l, q_n, k_l, k_n, v_l, v_n, o_l, o_n; };

// one lane's share of a K / V tile: 16 floats of row `key` of each (zeros past the end)
__device__ __forceinline__ void fetch_rows(const float *kb, const float *vb, long k_l, long v_l, int key,
                                           int half, int L, float4 (&kn)[4], float4 (&vn)[4]) {
    const int kc = min(key, L - 1);
    const float4 *kp = reinterpret_cast<const float4 *>(kb + (long)kc * k_l + half * 16);
    const float4 *vp = reinterpret_cast<const float4 *>(vb + (long)kc * v_l + half * 16);
    const float keep = key < L ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = kp[i], b = vp[i];
        kn[i] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
        vn[i] = make_float4(b.x * keep, b.y * keep, b.z * keep, b.w * keep);
    }
}

// Work split: a workgroup owns 32 queries of one (batch, head); wave w walks the key tiles
// w, w + 4, ... with PRIVATE LDS staging (wave-synchronous: no barrier in the loop), the next
// tile's K / V rows are in flight in registers while the current tile is multiplied; the four
// partial (max, sum, O^T) triples meet in LDS at the end.  35 x N x H workgroups of 4 waves give
// every SIMD 4-5 waves to interleave (one 128-query workgroup per CU left a lone wave per SIMD
// exposed to every latency: 131 us; this split: 86 us).
// (four waves per SIMD, stated: 124 registers; without the attribute the compiler settles for three: 83.7 -> 79.3 us)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void mha_fwd_d32(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ mask, float *__restrict__ out, float *__restrict__ lse, int L, int H,
    Strides st, float scale)
{
    __shared__ __attribute__((aligned(16))) float smem[kWaves * kWaveLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.y / H, h = blockIdx.y % H;
    const int qi = blockIdx.x * 32 + l31;                      // this lane's query
    const int qc = min(qi, L - 1);
    const float *kb = k + (long)n * st.k_n + h * 32;
    const float *vb = v + (long)n * st.v_n + h * 32;
    float *Ks = smem + wave * kWaveLds, *Vs = Ks + 32 * KS;

    // Q^T operand: lane holds Q[query][lhi * 16 + t], t = 0..15, pre-scaled
    float qreg[16];
    {
        const float4 *qp = reinterpret_cast<const float4 *>(q + (long)qc * st.q_l + (long)n * st.q_n +
                                                            h * 32 + lhi * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t4 = qp[i];
            qreg[4 * i] = t4.x * scale; qreg[4 * i + 1] = t4.y * scale;
            qreg[4 * i + 2] = t4.z * scale; qreg[4 * i + 3] = t4.w * scale;
        }
    }
    f32x16 oacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float *mrow = mask ? mask + (long)qc * L : nullptr;
    const bool vec_mask = (L & 3) == 0;

    // staging by one wave: lane -> (row = lane >> 1, half = lane & 1): 16 floats of K and of V
    const int srow = lane >> 1, shalf = lane & 1;
    float4 kn[4], vn[4];
    const int step = 32 * kWaves;
    int j0 = wave * 32;
    if (j0 < L) fetch_rows(kb, vb, st.k_l, st.v_l, j0 + srow, shalf, L, kn, vn);
    for (; j0 < L; j0 += step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&Ks[srow * KS + shalf * 16 + 4 * i]) = kn[i];
            *reinterpret_cast<float4 *>(&Vs[srow * VS + shalf * 16 + 4 * i]) = vn[i];
        }
        if (j0 + step < L)                                     // next tile in flight
            fetch_rows(kb, vb, st.k_l, st.v_l, j0 + step + srow, shalf, L, kn, vn);

        // ---- S^T = K Q^T : [32 keys] x [32 queries] -------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&Ks[l31 * KS + lhi * 16 + 4 * i]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, qreg[4 * i], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, qreg[4 * i + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, qreg[4 * i + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, qreg[4 * i + 3], sacc, 0, 0, 0);
        }

        // ---- mask, online softmax (lane = query; register e = key (e&3) + 8 (e>>2) + 4 lhi) ----
        float s[16];
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
            s[4 * g] = key < L ? sacc[4 * g] + mk.x : -INFINITY;
            s[4 * g + 1] = key + 1 < L ? sacc[4 * g + 1] + mk.y : -INFINITY;
            s[4 * g + 2] = key + 2 < L ? sacc[4 * g + 2] + mk.z : -INFINITY;
            s[4 * g + 3] = key + 3 < L ? sacc[4 * g + 3] + mk.w : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_safe);
        float p[16], rs = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { p[e] = __expf(s[e] - m_safe); rs += p[e]; }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] *= alpha;

        // ---- O^T += V^T P^T : step e pairs keys (e&3)+8(e>>2) [lanes 0-31] and +4 [lanes 32-63] -
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * VS + l31], p[e], oacc, 0, 0, 0);
        }
    }

    // ---- combine the four key ranges: partials through LDS (the staging space is free now) -------
    __syncthreads();
    float *part = smem + wave * kWaveLds;                      // [18][64]: 16 O^T registers, m, l
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e * 64 + lane] = oacc[e];
    part[16 * 64 + lane] = m_run;
    part[17 * 64 + lane] = l_run;
    __syncthreads();
    if (wave == 0 && qi < L) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) m = fmaxf(m, smem[w * kWaveLds + 16 * 64 + lane]);
        const float m_safe = m == -INFINITY ? 0.f : m;
        float l = 0.f, o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float *pw = smem + w * kWaveLds;
            const float f = __expf(pw[16 * 64 + lane] - m_safe);
            l += pw[17 * 64 + lane] * f;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] += pw[e * 64 + lane] * f;
        }
        const float inv = 1.f / l;
        float *op = out + (long)qi * st.o_l + (long)n * st.o_n + h * 32 + 4 * lhi;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(op + 8 * g) =
                make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        if (lse && lhi == 0) lse[((long)n * H + h) * L + qi] = m + logf(l);
    }
}

}  // namespace

extern "C" int datr_mha_forward_d32_f32(const float *q, const float *k, const float *v,
                                        const float *mask, int64_t L, int64_t N, int64_t H,
                                        const int64_t *strides, float scale, float *out, float *lse,
                                        void *stream) {
    if (L <= 0 || N <= 0 || H <= 0) return DATR_EINVAL;
    if (!q || !k || !v || !out || !strides) return DATR_EINVAL;
    if (L > 0x7fffff || N * H > 65535) return DATR_EUNSUPPORTED;
    for (int i = 0; i < 8; ++i)
        if (strides[i] % 4 != 0) return DATR_EUNSUPPORTED;            // float4 loads / stores
    Strides st{strides[0], strides[1], strides[2], strides[3], strides[4], strides[5], strides[6], strides[7]};
    dim3 grid((unsigned)((L + 31) / 32), (unsigned)(N * H));
    hipLaunchKernelGGL(mha_fwd_d32, grid, dim3(kThreads), 0, (hipStream_t)stream, q, k, v, mask, out, lse,
                       (int)L, (int)H, st, scale);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
