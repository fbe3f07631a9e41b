// msda_fwd_tiled.hip -- query-tiled MSDA forward for "self-attention over the pyramid" calls
// (Lq == S, D == 32): the encoder calls.
//
// Why.  The row kernel (msda.hip) gathers 64 corner rows x 128 B per output row straight from
// L1/L2: 2.9 GB per 1333x800 encoder call through a vector-memory path that tops out at
// ~24-28 TB/s chip-wide (profiles/r01_probes.md), i.e. >= 100 us no matter how the rows hit in
// cache.  LDS reads (ds_read_b128, 256 B/clk/CU) are ~5x faster, and neighbouring queries
// sample neighbouring pixels, so:
//
//   * a workgroup owns a 16x8 tile of queries of one level and one head (same tiling as the
//     tiled backward); every 8-lane group owns TWO of those queries for the whole kernel and
//     keeps their float4 accumulators in registers across the level loop;
//   * per target level the workgroup stages the value rows of a window (tile footprint mapped
//     to that level + a margin, <= 480 rows = 60 KB) into LDS with coalesced 128-B row loads,
//     ONE barrier, then every group gathers its 2 x P x 4 corners from LDS;
//   * a corner outside the window (large learned offsets) is fetched from global memory with
//     the same zero-filling buffer load the row kernel uses, so results never depend on the
//     window size; corners outside the image contribute zero;
//   * per-pair geometry (4 LDS/global addresses + 4 weights) lives in a 256-B LDS slot private
//     to the group: written and read by the same wave, no barrier needed.
// The result equals the row kernel's up to fp32 summation order (levels are accumulated in the
// same l, p order; corner products are summed in the same order).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "datr_hip.h"
#include "msda_tiled.h"

namespace {

constexpr int kThreads = 512;
constexpr int kLPR = 8;
constexpr int kGroups = kThreads / kLPR;          // 64 groups x 2 queries = 128 = 16 x 8 tile
constexpr int kWinRows = 480;
constexpr int kMargin = 5;                        // pixels of halo around the tile footprint
constexpr int kGlobalFlag = (int)0x80000000;       // slot address: bit 31 set = global pixel index

struct PairSlot { int a[4]; float w[4]; };          // 32 B

__device__ __forceinline__ float4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128");
    return __builtin_bit_cast(float4, r);
}

__global__ __launch_bounds__(kThreads) void msda_fwd_tiled_d32(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const DatrTiledMeta meta, int S, int M, int P, float *__restrict__ out)
{
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *win = reinterpret_cast<float *>(smem);                                     // 60 KB
    PairSlot *slots = reinterpret_cast<PairSlot *>(smem + kWinRows * D * 4);          // 16 KB

    const int L = meta.L, K = L * P, Lq = meta.Lq;
    const int bid = blockIdx.x;
    const int m = bid % M;
    const int tile = (bid / M) % meta.total_tiles;
    const int n = bid / (M * meta.total_tiles);
    int lq = 0;
    while (lq + 1 < meta.QL && tile >= meta.qlv[lq + 1].tile_base) ++lq;
    const int tl = tile - meta.qlv[lq].tile_base;
    const int tx = tl % meta.qlv[lq].tiles_x, ty = tl / meta.qlv[lq].tiles_x;
    const int qx0 = tx * meta.tile_w, qy0 = ty * meta.tile_h;
    const int qW = meta.qlv[lq].W, qH = meta.qlv[lq].H;
    const int tw = min(meta.tile_w, qW - qx0), th = min(meta.tile_h, qH - qy0);
    const int nq = tw * th;
    const int q_base = meta.qlv[lq].start + qy0 * qW + qx0;

    const int tid = threadIdx.x, g = tid / kLPR, j = tid % kLPR;
    const unsigned row_bytes = (unsigned)(M * D) * 4u;
    const size_t item = ((size_t)n * S * M + m) * D;
    const int records = (S * M - m) * D * 4;
    __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(value + item), 0, records, 0x00020000);
    const unsigned chan = (unsigned)j * 16u;

    // this group's two queries (the second may not exist in a ragged tile)
    const int qiA = g, qiB = g + kGroups;
    const bool liveA = qiA < nq, liveB = qiB < nq;
    const int qA = q_base + ((liveA ? qiA : 0) / tw) * qW + ((liveA ? qiA : 0) % tw);
    const int qB = q_base + ((liveB ? qiB : 0) / tw) * qW + ((liveB ? qiB : 0) % tw);
    // lane j prepares pair (k, p): k = which of the two queries, p = point
    const int kk = j / P, pp = j - kk * P;
    const bool lane_has_pair = j < 2 * P;
    const bool pair_live = lane_has_pair && (kk == 0 ? liveA : liveB);
    const int q_mine = kk == 0 ? qA : qB;
    PairSlot *my_slots = slots + g * 8;

    float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;

    // ---- all levels' sampling locations / weights for this lane's pair, loaded up front so that
    //      their (cold) latency is paid once, not once per level -------------------------------
    constexpr int kMaxL = DATR_TILED_MAX_LEVELS;
    float2 xy_l[kMaxL];
    float a_l[kMaxL];
#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        xy_l[l] = make_float2(-4.f, -4.f);
        a_l[l] = 0.f;
        if (l < L && pair_live) {
            const size_t k = (((size_t)n * Lq + q_mine) * M + m) * K + l * P + pp;
            xy_l[l] = reinterpret_cast<const float2 *>(loc)[k];
            a_l[l] = attn[k];
        }
    }

#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        if (l >= L) break;
        const int H = meta.lv[l].H, W = meta.lv[l].W, start = meta.lv[l].start;
        // ---- window: the tile's footprint in level l plus the largest margin (<= kMargin) that
        //      fits.  pixel-centre mapping: x_l = (x + 0.5) / qW * W - 0.5
        int fx0 = (int)floorf(((float)qx0 + 0.5f) / (float)qW * (float)W - 0.5f);
        int fx1 = (int)floorf(((float)(qx0 + tw) - 0.5f) / (float)qW * (float)W - 0.5f) + 1;
        int fy0 = (int)floorf(((float)qy0 + 0.5f) / (float)qH * (float)H - 0.5f);
        int fy1 = (int)floorf(((float)(qy0 + th) - 0.5f) / (float)qH * (float)H - 0.5f) + 1;
        int R = kMargin;
        int wx0, wx1, wy0, wy1;
        for (;;) {
            wx0 = max(fx0 - R, 0); wx1 = min(fx1 + R, W - 1);
            wy0 = max(fy0 - R, 0); wy1 = min(fy1 + R, H - 1);
            if ((wx1 - wx0 + 1) * (wy1 - wy0 + 1) <= kWinRows || R == 0) break;
            --R;
        }
        int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
        if (ww * wh > kWinRows) wh = kWinRows / ww;            // R == 0 and still too large
        const int wrows = ww * wh;

        // ---- stage the window's value rows (coalesced: 8 lanes x 16 B per row) -----------------
        for (int r = g; r < wrows; r += kGroups) {
            const int cy = r / ww, cx = r - cy * ww;
            const unsigned pix = (unsigned)(start + (wy0 + cy) * W + wx0 + cx);
            const float4 v = load_row4(vsrc, pix * row_bytes + chan);
            *reinterpret_cast<float4 *>(win + r * D + j * 4) = v;
        }
        // ---- geometry of this group's pairs (private LDS slots, same wave reads them) -----------
        if (lane_has_pair) {
            PairSlot ps;
            ps.a[0] = ps.a[1] = ps.a[2] = ps.a[3] = 0;
            ps.w[0] = ps.w[1] = ps.w[2] = ps.w[3] = 0.f;
            {
                const float2 xy = xy_l[l];
                const float a = a_l[l];
                const float Hf = (float)H, Wf = (float)W;
                const float h_im = xy.y * Hf - 0.5f, w_im = xy.x * Wf - 0.5f;
                if (pair_live && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int y0 = (int)hf, x0 = (int)wf;
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const float cw[4] = {a * (hh * hw), a * (hh * lw), a * (lh * hw), a * (lh * lw)};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int y = y0 + (c >> 1), x = x0 + (c & 1);
                        if (y >= 0 && y <= H - 1 && x >= 0 && x <= W - 1) {
                            ps.w[c] = cw[c];
                            const int cy = y - wy0, cx = x - wx0;
                            if ((unsigned)cy < (unsigned)wh && (unsigned)cx < (unsigned)ww)
                                ps.a[c] = (cy * ww + cx) * (D * 4);              // LDS byte address
                            else
                                ps.a[c] = kGlobalFlag | (start + y * W + x);       // global pixel
                        }
                    }
                }
            }
            my_slots[j] = ps;
        }
        __syncthreads();

        // ---- gather: 2 queries x P points x 4 corners per group, from LDS -----------------------
        const char *winb = reinterpret_cast<const char *>(win) + j * 16;
#pragma unroll 2
        for (int s = 0; s < 2 * P; ++s) {
            const int4 ad = reinterpret_cast<const int4 *>(my_slots + s)[0];
            const float4 w = reinterpret_cast<const float4 *>(my_slots + s)[1];
            float4 v0, v1, v2, v3;
            if ((ad.x | ad.y | ad.z | ad.w) >= 0) {               // all four in the window
                v0 = *reinterpret_cast<const float4 *>(winb + ad.x);
                v1 = *reinterpret_cast<const float4 *>(winb + ad.y);
                v2 = *reinterpret_cast<const float4 *>(winb + ad.z);
                v3 = *reinterpret_cast<const float4 *>(winb + ad.w);
            } else {                                               // some corner is out of window
#define DATR_FETCH(A) ((A) >= 0 ? *reinterpret_cast<const float4 *>(winb + (A))                    \
                                : load_row4(vsrc, (unsigned)((A) & 0x7fffffff) * row_bytes + chan))
                v0 = DATR_FETCH(ad.x);
                v1 = DATR_FETCH(ad.y);
                v2 = DATR_FETCH(ad.z);
                v3 = DATR_FETCH(ad.w);
#undef DATR_FETCH
            }
            float4 t;
            t.x = w.x * v0.x + w.y * v1.x + w.z * v2.x + w.w * v3.x;
            t.y = w.x * v0.y + w.y * v1.y + w.z * v2.y + w.w * v3.y;
            t.z = w.x * v0.z + w.y * v1.z + w.z * v2.z + w.w * v3.z;
            t.w = w.x * v0.w + w.y * v1.w + w.z * v2.w + w.w * v3.w;
            if (s < P) { accA.x += t.x; accA.y += t.y; accA.z += t.z; accA.w += t.w; }
            else       { accB.x += t.x; accB.y += t.y; accB.z += t.z; accB.w += t.w; }
        }
        __syncthreads();            // the next level re-uses the window
    }
    if (liveA)
        reinterpret_cast<float4 *>(out + (((size_t)n * Lq + qA) * M + m) * D)[j] = accA;
    if (liveB)
        reinterpret_cast<float4 *>(out + (((size_t)n * Lq + qB) * M + m) * D)[j] = accB;
}

}  // namespace

extern "C" int datr_internal_msda_fwd_tiled_d32(
    const float *value, const float *loc, const float *attn, const DatrTiledMeta *meta, int64_t N,
    int64_t S, int64_t M, int64_t P, float *out, void *stream)
{
    const int64_t blocks = N * M * meta->total_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return DATR_EUNSUPPORTED;
    const size_t lds = (size_t)kWinRows * 32 * 4 + (size_t)kGroups * 8 * sizeof(PairSlot);
    hipLaunchKernelGGL(msda_fwd_tiled_d32, dim3((unsigned)blocks), dim3(kThreads), lds,
                       (hipStream_t)stream, value, loc, attn, *meta, (int)S, (int)M, (int)P, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
