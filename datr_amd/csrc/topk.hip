// topk.hip -- deterministic row-wise top-k (k <= 1024) for the two selections of the hot path:
// the two-stage query selection `torch.topk(enc_outputs_class.max(-1)[0], 900, dim=1)[1]`
// (/root/reference/models/dino/deformable_transformer.py:342) and PostProcess's
// `torch.topk(prob.view(B, -1), num_select, dim=1)` (/root/reference/models/dino/dino.py:960).
//
// Why an own kernel: the north star asks for bit-exact index selection, and `torch.topk` fixes
// neither the order nor (at the k-th value) the membership of tied scores -- its CPU and GPU
// implementations disagree with each other on ties.  This kernel implements ONE total order:
// descending score, ties by ascending index; NaN sorts above +inf (as torch does).  Whenever the
// scores are distinct -- the generic case -- that is exactly the reference's result.
//
// One 1024-thread workgroup per row (the path has 2-4 rows of 1 700 - 22 223 scores: latency, not
// throughput, is what matters):
//   1. four 8-bit radix passes over the order-preserving 32-bit keys find the k-th largest key T
//      and how many keys are greater (histograms in LDS, wave-aggregated);
//   2. one ordered pass collects every key > T plus the lowest-index (k - greater) keys == T
//      (ballot + prefix ranks, so the choice among ties is by index, not by arrival);
//   3. the k (key, ~index) pairs are bitonic-sorted in LDS and written out.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

// float -> unsigned key with the same order (larger float = larger key); NaN above everything
__device__ __forceinline__ unsigned order_key(float x) {
    unsigned u = __float_as_uint(x);
    if (x != x) return 0xFFFFFFFFu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kThreads) void topk_rows_kernel(const float *__restrict__ scores, int n,
                                                             int k, long long *__restrict__ out_idx,
                                                             float *__restrict__ out_val) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel[1024];
    __shared__ unsigned wave_cnt[kWaves];
    __shared__ unsigned s_prefix, s_remaining, s_ngt, s_taken_eq;

    const float *row = scores + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- 1. radix select: the k-th largest key --------------------------------------------------
    if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)k; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n; i += kThreads) {
            const unsigned key = order_key(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned need = s_remaining, d = 255;
            for (;; --d) {                       // highest digit first
                const unsigned c = hist[d];
                if (c >= need || d == 0) break;
                need -= c;
            }
            s_prefix = prefix | (d << shift);
            s_remaining = need;                  // rank of the k-th key among those sharing the prefix
        }
        __syncthreads();
    }
    const unsigned T = s_prefix;                 // the k-th largest key
    const unsigned need_eq = s_remaining;        // how many keys == T belong to the top k

    // ---- 2. collect: all keys > T, and the lowest-index `need_eq` keys == T ----------------------
    if (tid == 0) { s_ngt = 0; s_taken_eq = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += kThreads) {
        const int i = base + tid;
        unsigned key = 0;
        bool gt = false, eq = false;
        if (i < n) {
            key = order_key(row[i]);
            gt = key > T;
            eq = key == T;
        }
        // ranks among the == T keys in index order: ballot inside the wave, prefix over the waves
        const unsigned long long eq_mask = __ballot(eq);
        const unsigned eq_before = __popcll(eq_mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(eq_mask);
        __syncthreads();
        unsigned offset = s_taken_eq;
        for (int w = 0; w < wave; ++w) offset += wave_cnt[w];
        if (gt) {
            const unsigned slot = atomicAdd(&s_ngt, 1u);          // order irrelevant: sorted below
            sel[slot] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)i);
        }
        const bool take = eq && (offset + eq_before) < need_eq;
        __syncthreads();                                          // everyone has read s_taken_eq
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < kWaves; ++w) tot += wave_cnt[w];
            s_taken_eq += tot;
        }
        if (take) {
            // equal keys go behind the greater ones: slots k - need_eq + rank
            sel[(unsigned)k - need_eq + offset + eq_before] =
                ((unsigned long long)key << 32) | (unsigned)(~(unsigned)i);
        }
        __syncthreads();
    }
    // pad to a power of two with the smallest composite key
    for (int i = k + tid; i < 1024; i += kThreads) sel[i] = 0ull;
    __syncthreads();

    // ---- 3. bitonic sort, descending (larger key first; equal keys: larger ~index = lower index) --
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = tid ^ stride;
            if (partner > tid) {
                const unsigned long long a = sel[tid], b = sel[partner];
                const bool desc = (tid & size) == 0;
                if (desc ? a < b : a > b) { sel[tid] = b; sel[partner] = a; }
            }
            __syncthreads();
        }
    }
    if (tid < k) {
        const unsigned idx = ~(unsigned)(sel[tid] & 0xFFFFFFFFull);
        out_idx[(size_t)blockIdx.x * k + tid] = (long long)idx;
        if (out_val) out_val[(size_t)blockIdx.x * k + tid] = row[idx];
    }
}

}  // namespace

extern "C" int datr_topk_rows_f32(const float *scores, int64_t rows, int64_t n, int64_t k,
                                  int64_t *out_idx, float *out_val, void *stream) {
    if (rows < 0 || n < 1 || k < 1 || k > n) return DATR_EINVAL;
    if (k > 1024 || n >= (1LL << 31) || rows >= (1LL << 31)) return DATR_EUNSUPPORTED;
    if (rows == 0) return DATR_OK;
    if (!scores || !out_idx) return DATR_EINVAL;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream,
                       scores, (int)n, (int)k, reinterpret_cast<long long *>(out_idx), out_val);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
