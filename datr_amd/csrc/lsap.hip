// lsap.hip -- rectangular linear-sum assignment (Hungarian matching) on the device.
//
// The reference solves one [num_queries x T_i] assignment per image and prediction set on the
// HOST: `C.cpu()` followed by scipy.optimize.linear_sum_assignment
// (/root/reference/models/dino/matcher.py:91-95) -- seven device->host synchronisations per
// training step, the only mid-step syncs the step has.  This kernel solves all of a step's
// problems (7 prediction sets x B images) in one launch and leaves the indices on the device, so
// the host never waits and can run ahead of the GPU.
//
// Algorithm: the one SciPy uses (Crouse's shortest-augmenting-path variant of
// Jonker-Volgenant), restated step for step so that the assignment is the same even under ties:
// the tall cost matrix is transposed (rows = the T ground-truth boxes, columns = the queries);
// for every row, grow a shortest augmenting path: scan the REMAINING columns in SciPy's order
// (`remaining[it] = nc-1-it` initially, swap-with-last removal), relax
// r = minVal + cost[i][j] - u[i] - v[j] (double, evaluated left to right), pick the smallest
// shortestPathCost -- among equal ones the LAST unassigned column in scan order, else the first
// -- until an unassigned column (the sink) is reached; update the duals u, v; flip the path.
// Costs arrive as fp32 and are widened to double exactly as numpy does for SciPy.
//
// Mapping: ONE WAVE per problem (wave-synchronous, no barriers).  Lane l owns columns
// l, l+64, ...: their duals v, shortestPathCosts, scan positions and path live in registers;
// the per-iteration arg-min is a 6-step xor-shuffle reduction on (value, tie key, column).
// Row state (u, col4row), row4col and the path (for the final flip) live in LDS, and so does the
// problem's transposed cost block when it fits (T x nc fp32 <= 96 KB), else rows are re-read
// from global memory (coalesced: the caller passes the cost TRANSPOSED, [.., T, nc]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "datr_hip.h"

namespace {

constexpr int kK = 16;                      // columns per lane  => nc <= 1024
constexpr int kMaxCols = 64 * kK;
constexpr int kStageBytes = 96 * 1024;

struct Best { double val; int key; int col; };

__device__ __forceinline__ bool better(const Best &a, const Best &b) {
    return a.val < b.val || (a.val == b.val && a.key < b.key);
}

__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
    for (int mask = 32; mask >= 1; mask >>= 1) {
        Best o;
        o.val = __shfl_xor(x.val, mask, 64);
        o.key = __shfl_xor(x.key, mask, 64);
        o.col = __shfl_xor(x.col, mask, 64);
        if (better(o, x)) x = o;
    }
    return x;
}

// costT: [P][Tsum][nc] fp32, problem p = g * B + b uses rows [offsets[b], offsets[b+1]).
// q_out / t_out: [G][Tsum] int64; problem (g, b) fills entries offsets[b] .. offsets[b+1]-1 with
// the matched query indices in ascending order and the box index matched to each.
__global__ __launch_bounds__(64) void lsap_wave_kernel(
    const float *__restrict__ costT, const int *__restrict__ offsets, int B, int Tsum, int nc,
    int nr_max, int stage, int64_t *__restrict__ q_out, int64_t *__restrict__ t_out,
    int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *u = reinterpret_cast<double *>(smem);                       // [nr_max]
    double *vis_val = u + nr_max;                                        // [nr_max]
    int *col4row = reinterpret_cast<int *>(vis_val + nr_max);            // [nr_max]
    int *vis_row = col4row + nr_max;                                     // [nr_max]
    int *row4col = vis_row + nr_max;                                     // [kMaxCols]
    int *path = row4col + kMaxCols;                                      // [kMaxCols]
    float *cst = reinterpret_cast<float *>(path + kMaxCols);             // [nr][nc] when staged

    const int p = blockIdx.x, b = p % B, g = p / B;
    const int off = offsets[b], nr = offsets[b + 1] - off;
    const int lane = threadIdx.x;
    if (nr <= 0) return;
    const float *cbase = costT + ((size_t)p * Tsum + off) * nc;

    bool bad = false;
    if (stage) {
        for (int e = lane; e < nr * nc; e += 64) {
            const float c = cbase[e];
            bad |= !(c == c) || c == -INFINITY;
            cst[e] = c;
        }
    }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < kMaxCols; j += 64) { row4col[j] = -1; path[j] = -1; }

    double v[kK], spc[kK];
    int pos[kK], pth[kK];
    unsigned amask = 0;                    // bit k: column lane + 64 k is assigned (row4col != -1)
#pragma unroll
    for (int k = 0; k < kK; ++k) { v[k] = 0.0; pth[k] = -1; }

    int failed = 0;
    for (int cur = 0; cur < nr && !failed; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            const int j = lane + 64 * k;
            pos[k] = j < nc ? nc - 1 - j : -1;          // remaining[it] = nc - it - 1
            spc[k] = INFINITY;
        }
        int nvis = 0, i = cur, sink = -1;
        while (sink == -1) {
            const double ui = u[i];
            Best best{INFINITY, 0x7fffffff, -1};
            const float *crow = stage ? cst + (size_t)i * nc : cbase + (size_t)i * nc;
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                const int j = lane + 64 * k;
                if (pos[k] >= 0) {
                    const float cf = crow[j];
                    if (!stage) bad |= !(cf == cf) || cf == -INFINITY;
                    const double r = ((minVal + (double)cf) - ui) - v[k];
                    if (r < spc[k]) { spc[k] = r; pth[k] = i; }
                    Best c{spc[k], ((amask >> k) & 1u) ? pos[k] : -pos[k] - 1, j};
                    if (better(c, best)) best = c;
                }
            }
            best = wave_best(best);
            const double lowest = best.val;
            minVal = lowest;
            if (!(lowest < INFINITY)) { failed = 1; break; }       // infeasible (or NaN) matrix
            const int js = best.col;
            const int it = best.key >= 0 ? best.key : -best.key - 1;
            const int r4c = row4col[js];
            if (r4c == -1) {
                sink = js;
            } else {
                i = r4c;
                if (lane == 0) { vis_row[nvis] = i; vis_val[nvis] = lowest; }
                ++nvis;
            }
            // SC[js] = true; remaining[it] = remaining[--num_remaining]
            --num_remaining;
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                if (pos[k] == num_remaining) pos[k] = it;
                if (lane + 64 * k == js) pos[k] = -2;
            }
        }
        if (failed) break;
        // dual variables
        if (lane == 0) u[cur] += minVal;
        for (int t = lane; t < nvis; t += 64) u[vis_row[t]] += minVal - vis_val[t];
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            if (pos[k] == -2) v[k] -= minVal - spc[k];
            path[lane + 64 * k] = pth[k];
        }
        // augment the previous solution along the path
        int j = sink;
        for (;;) {
            const int ii = path[j];
            row4col[j] = ii;
            if ((j & 63) == lane) amask |= 1u << (j >> 6);
            const int prev = col4row[ii];
            if (lane == 0) col4row[ii] = j;
            j = prev;
            if (ii == cur) break;
        }
    }
    bad |= failed != 0;
    const unsigned long long any_bad = __builtin_amdgcn_ballot_w64(bad);
    if (lane == 0) status[p] = any_bad ? 1 : 0;
    if (any_bad) return;

    // col4row[i] = query matched to box i; emit pairs in ascending query order (what SciPy
    // returns for a transposed problem)
    for (int i = lane; i < nr; i += 64) {
        const int q = col4row[i];
        int rank = 0;
        for (int t = 0; t < nr; ++t) rank += col4row[t] < q;
        q_out[(size_t)g * Tsum + off + rank] = q;
        t_out[(size_t)g * Tsum + off + rank] = i;
    }
}

}  // namespace

extern "C" int datr_lsap_f32(const float *cost_t, const int32_t *offsets, int64_t G, int64_t B,
                             int64_t Tsum, int64_t nc, int64_t max_rows, int64_t *q_idx,
                             int64_t *t_idx, int32_t *status, void *stream) {
    if (G < 0 || B < 0 || Tsum < 0 || nc <= 0 || max_rows < 0) return DATR_EINVAL;
    if (G * B == 0 || Tsum == 0 || max_rows == 0) return DATR_OK;
    // T == nc is a square problem, which SciPy solves WITHOUT transposing (different tie
    // resolution): not handled here
    if (nc > kMaxCols || max_rows >= nc || G * B > 0x7fffffff) return DATR_EUNSUPPORTED;
    if (!cost_t || !offsets || !q_idx || !t_idx || !status) return DATR_EINVAL;
    const size_t fixed = (size_t)max_rows * (8 + 8 + 4 + 4) + (size_t)kMaxCols * 8;
    const size_t cost_bytes = (size_t)max_rows * nc * 4;
    const int stage = cost_bytes <= (size_t)kStageBytes;
    const size_t lds = fixed + (stage ? cost_bytes : 0);
    if (lds > 150 * 1024) return DATR_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_wave_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return DATR_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(lsap_wave_kernel, dim3((unsigned)(G * B)), dim3(64), lds, (hipStream_t)stream,
                       cost_t, offsets, (int)B, (int)Tsum, (int)nc, (int)max_rows, stage, q_idx, t_idx,
                       status);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
