// msda_fwd_win.hip -- query-tiled MSDA forward for the encoder calls (Lq == S, D == 32, P == 4,
// L <= 4): per 16 x 8 query tile, head and level, the window of value rows the tile can reach is
// staged in LDS and gathered from there.  EXPERIMENTAL: parity-green, but measured SLOWER than
// the row kernel of msda.hip (131 us vs 104 us per 1333x800 encoder call), so it is off by
// default (datr_amd/msda.py TILED_FORWARD); it stays in the tree as the measured record of what
// the LDS route costs on gfx950 (profiles/r01_probes.md, "MSDA forward through LDS").
//
// What it does (MI355X_MICROARCH.md, "LDS"):
//   * ds_read_b128 is served in four fixed 16-lane groups -- {0-3,12-15,20-27},
//     {4-11,16-19,28-31} and the same +32 -- over 64 banks; a group is conflict-free iff its
//     four 64-byte pieces fall on distinct 16-bank quarters.  With 8 lanes per 128-B value row
//     ("row group", rg) a hardware group sees the low half of rg0's row, the high halves of
//     rg1's and rg2's, and the low half of rg3's: it is conflict-free iff the rows of
//     (rg0, rg3) and of (rg1, rg2) differ in row-index parity.  So row groups work in PAIRS on
//     one sample: the left group reads corner (y, x0), the right group corner (y, x0+1) -- their
//     window rows are r and r+1, always of opposite parity -- first for y = y0, then y0+1.
//   * windows are filled by LDS-DMA (`global_load_lds_dwordx4`: 8 rows = 1 KiB per wave
//     instruction, no staging registers, no ds_write pass);
//   * per-sample geometry (2 LDS addresses + 2 weights per column) is computed once per level by
//     the owning wave, branch-free, one sample per lane, parked in wave-private LDS slots and
//     read back as one broadcast ds_read_b128 per (sample, row group); the slots of the next
//     query are prefetched while the current one is gathered;
//   * all levels' sampling locations are loaded up front (one cold-miss latency per tile);
//   * a pair keeps the accumulators of its queries (left-column and right-column partial sums)
//     in registers across the level loop; the two partials meet once at the end (ds_swizzle);
//   * a corner in the image but outside the window (large learned offsets; coarse-level tiles
//     whose footprint in a fine level exceeds the window) is fetched from global memory with a
//     zero-filling buffer load, so results never depend on the window heuristic.
// Why it loses (ablations in profiles/r01_probes.md): the gather itself is 4x cheaper per byte,
// but geometry (one VALU pass per sample, ~100 instructions), window fills and the four
// fill -> barrier -> gather -> barrier phases per tile do not overlap within a 2-workgroup-per-CU
// budget (76 KB LDS each); a persistent 1024-thread double-buffered variant measured 150 us.
// Result = the row kernel's up to fp32 summation order.  Zero-weight (out-of-image) corners
// read window rows 0 / 1, so unlike the row kernel a non-finite value there would propagate.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "datr_hip.h"
#include "msda_tiled.h"

#ifdef DATR_PROBE
// per-phase cycle counters of thread 0 of every workgroup, kept in registers and added to one of
// 256 counter sets at the end of the kernel (a single hot address would serialise the blocks)
__device__ unsigned long long datr_fwd_phase_cycles[256][8];
#define DATR_TICK(i)                                                                   \
    do {                                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();                  \
        ticks_[i] += now_ - tick_;                                                     \
        tick_ = now_;                                                                  \
    } while (0)
#define DATR_TICK_INIT unsigned long long ticks_[8] = {0, 0, 0, 0, 0, 0, 0, 0};         \
                       unsigned long long tick_ = __builtin_readcyclecounter()
#define DATR_TICK_FLUSH                                                                \
    if (threadIdx.x == 0)                                                              \
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&datr_fwd_phase_cycles[blockIdx.x & 255][i_], ticks_[i_])
extern "C" void datr_probe_fwd_phase_cycles(unsigned long long *out, int reset) {
    static unsigned long long all[256][8];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(datr_fwd_phase_cycles), sizeof(all));
    for (int i = 0; i < 8; ++i) {
        out[i] = 0;
        for (int b = 0; b < 256; ++b) out[i] += all[b][i];
    }
    if (reset) {
        static unsigned long long z[256][8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(datr_fwd_phase_cycles), z, sizeof(z));
    }
}
#else
#define DATR_TICK(i) do {} while (0)
#define DATR_TICK_INIT do {} while (0)
#define DATR_TICK_FLUSH do {} while (0)
#endif

namespace {

constexpr int kWinRows = 480;                      // 60 KB of value rows
constexpr int kMargin = 5;                         // halo around the tile's footprint, pixels
constexpr int kGlobalFlag = (int)0x80000000;       // slot address with bit 31: global pixel index
constexpr int kTileQ = DATR_TILE_W * DATR_TILE_H;  // 128 queries per workgroup

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ float4 load_row4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    static_assert(sizeof(r) == 16, "b128");
    return __builtin_bit_cast(float4, r);
}

struct Window { int x0, y0, w, h; };

// The tile's footprint in level (H, W) plus the largest halo (<= kMargin) that fits kWinRows.
__device__ __forceinline__ Window level_window(int qx0, int qy0, int tw, int th, int qW, int qH,
                                               int W, int H) {
    const int fx0 = (int)floorf(((float)qx0 + 0.5f) / (float)qW * (float)W - 0.5f);
    const int fx1 = (int)floorf(((float)(qx0 + tw) - 0.5f) / (float)qW * (float)W - 0.5f) + 1;
    const int fy0 = (int)floorf(((float)qy0 + 0.5f) / (float)qH * (float)H - 0.5f);
    const int fy1 = (int)floorf(((float)(qy0 + th) - 0.5f) / (float)qH * (float)H - 0.5f) + 1;
    int R = kMargin, wx0, wx1, wy0, wy1;
    for (;;) {
        wx0 = max(fx0 - R, 0); wx1 = min(fx1 + R, W - 1);
        wy0 = max(fy0 - R, 0); wy1 = min(fy1 + R, H - 1);
        if ((wx1 - wx0 + 1) * (wy1 - wy0 + 1) <= kWinRows || R == 0) break;
        --R;
    }
    Window wd{wx0, wy0, wx1 - wx0 + 1, wy1 - wy0 + 1};
    if (wd.w * wd.h > kWinRows) wd.h = kWinRows / wd.w;     // R == 0 and still too large
    if (wd.h < 1) { wd.h = 1; wd.w = min(wd.w, kWinRows); }
    return wd;
}

template <int kThreads>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_fwd_win_d32(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const DatrTiledMeta meta, int S, int M, int abl, float *__restrict__ out)
{
    // abl (development): 1 = no window fill, 2 = no gather, 4 = no geometry
    constexpr int D = 32;
    constexpr int kWaves = kThreads / 64;
    constexpr int kQW = kTileQ / kWaves;           // queries per wave            (16)
    constexpr int kQP = kQW / 4;                   // queries per row-group pair  ( 4)
    static_assert(kQW * 4 == 64, "one (query, point) sample per lane and level");
    constexpr int kP = 4;                          // points per level (host checks P == 4)
    constexpr int kL = 4;                          // levels           (host checks L <= 4)
    constexpr int kSlotBytes = 32;                 // {aT,aB,wT,wB} left column, then right column
    constexpr int kWaveSlotBytes = kQW * 4 * kSlotBytes;

    DATR_TICK_INIT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char *win = smem;                                              // kWinRows x 128 B
    char *slots = smem + kWinRows * D * 4;                         // kTileQ x 4 x 32 B = 16 KB

    const int L = meta.L, K = L * kP, Lq = meta.Lq;
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 7, rg = lane >> 3;
    const int r3 = rg & 3;
    const int pair = ((rg & 4) >> 1) | ((r3 == 1 || r3 == 2) ? 1 : 0);     // 0..3
    const int col = r3 >> 1;                                                // 0 left, 1 right

    const unsigned row_bytes = (unsigned)(M * D) * 4u;
    const size_t item = ((size_t)n * S * M + m) * D;
    const float *vbase = value + item;
    const int records = (S * M - m) * D * 4;
    __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(vbase), 0, records, 0x00020000);
    const unsigned chan = (unsigned)j * 16u;

    // ---- geometry duty of this lane: point gp of wave-local query gq_w, every level -------------
    const int gq_w = lane >> 2, gp = lane & 3;
    const int gqi = wave * kQW + gq_w;                       // tile-local query
    const bool g_live = gqi < nq;
    const int gq = q_base + ((g_live ? gqi : 0) / tw) * qW + ((g_live ? gqi : 0) % tw);
    char *my_slots = slots + wave * kWaveSlotBytes;
    char *g_slot = my_slots + lane * kSlotBytes;            // (gq_w * 4 + gp) * 32
    // all levels' locations / weights up front: their cold-miss latency is paid once
    float lx[kL], ly[kL], la[kL];
    {
        const size_t k0 = (((size_t)n * Lq + gq) * M + m) * K + gp;
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            const size_t k = k0 + (l < L ? l : 0) * kP;
            const float2 xy = reinterpret_cast<const float2 *>(loc)[k];
            lx[l] = xy.x; ly[l] = xy.y; la[l] = g_live ? attn[k] : 0.f;
        }
    }
    // every level's window, computed by lane l and broadcast to scalars
    int wx0[kL], wy0[kL], ww[kL], wh[kL];
    {
        const int l = min(lane, L - 1);
        const Window wd = level_window(qx0, qy0, tw, th, qW, qH, meta.lv[l].W, meta.lv[l].H);
#pragma unroll
        for (int i = 0; i < kL; ++i) {
            wx0[i] = __builtin_amdgcn_readlane(wd.x0, i);
            wy0[i] = __builtin_amdgcn_readlane(wd.y0, i);
            ww[i] = __builtin_amdgcn_readlane(wd.w, i);
            wh[i] = __builtin_amdgcn_readlane(wd.h, i);
        }
    }

    float acc[kQP][4];
#pragma unroll
    for (int i = 0; i < kQP; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;

    // the slot this row group reads for (pair query i, point p): wave-local query = pair + 4 i
    const char *rd_slot = my_slots + pair * 4 * kSlotBytes + col * 16;
    const char *winj = win + chan;

    // window fill by LDS-DMA: 8 rows (1 KiB) per wave instruction
    auto fill_window = [&](int l) {
        if (abl & 1) return;
        const int W = meta.lv[l].W, start = meta.lv[l].start;
        const int wrows = ww[l] * wh[l];
        const unsigned inv = ((1u << 20) + (unsigned)ww[l] - 1u) / (unsigned)ww[l];
        for (int r0 = wave * 8; r0 < wrows; r0 += kWaves * 8) {
            const int r = min(r0 + rg, wrows - 1);
            const int cy = (int)(((unsigned)r * inv) >> 20), cx = r - cy * ww[l];
            const size_t pix = (size_t)(start + (wy0[l] + cy) * W + wx0[l] + cx);
            const float *src = vbase + pix * (size_t)(M * D) + j * 4;
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(win + r0 * 128), 16, 0, 0);
        }
    };
    // geometry of this lane's sample -> its wave-private slot (branch-free); returns whether any
    // corner of the WAVE's samples lies in the image but outside the window
    auto geometry = [&](int l, float x, float y, float a) -> bool {
        if (abl & 4) return false;
        const int H = meta.lv[l].H, W = meta.lv[l].W, start = meta.lv[l].start;
        const float Hf = (float)H, Wf = (float)W;
        const float h_im = y * Hf - 0.5f, w_im = x * Wf - 0.5f;
        const bool valid = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const int y0 = valid ? (int)hf : 0, x0 = valid ? (int)wf : 0;
        const bool okT = valid && y0 >= 0, okB = valid && y0 + 1 <= H - 1;
        const bool okL = x0 >= 0, okR = x0 + 1 <= W - 1;
        const int cy = y0 - wy0[l], cx = x0 - wx0[l];
        const bool inT = (unsigned)cy < (unsigned)wh[l], inB = (unsigned)(cy + 1) < (unsigned)wh[l];
        const bool inL = (unsigned)cx < (unsigned)ww[l], inR = (unsigned)(cx + 1) < (unsigned)ww[l];
        const int a_tl = (cy * ww[l] + cx) * 128, a_bl = a_tl + ww[l] * 128;
        const int g_tl = start + y0 * W + x0, g_bl = g_tl + W;
        int4 sl, sr;
        sl.x = okT && okL ? (inT && inL ? a_tl : (kGlobalFlag | g_tl)) : 0;
        sl.y = okB && okL ? (inB && inL ? a_bl : (kGlobalFlag | g_bl)) : 0;
        sr.x = okT && okR ? (inT && inR ? a_tl + 128 : (kGlobalFlag | (g_tl + 1))) : 128;
        sr.y = okB && okR ? (inB && inR ? a_bl + 128 : (kGlobalFlag | (g_bl + 1))) : 128;
        sl.z = __float_as_int(okT && okL ? a * (hh * hw) : 0.f);
        sl.w = __float_as_int(okB && okL ? a * (lh * hw) : 0.f);
        sr.z = __float_as_int(okT && okR ? a * (hh * lw) : 0.f);
        sr.w = __float_as_int(okB && okR ? a * (lh * lw) : 0.f);
        reinterpret_cast<int4 *>(g_slot)[0] = sl;
        reinterpret_cast<int4 *>(g_slot)[1] = sr;
        return __builtin_amdgcn_ballot_w64((sl.x | sl.y | sr.x | sr.y) < 0) != 0;
    };

    DATR_TICK(0);
    fill_window(0);
    bool wave_escaped = geometry(0, lx[0], ly[0], la[0]);
    DATR_TICK(1);

#pragma unroll
    for (int l = 0; l < kL; ++l) {
        if (l >= L) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA has landed
        DATR_TICK(2);
        __syncthreads();                                      // everybody's has
        DATR_TICK(3);

        // ---- gather: conflict-free paired ds_read_b128 --------------------------------------------
        if (abl & 2) {
        } else if (!wave_escaped) {
            int4 sl[kP], nx[kP];
#pragma unroll
            for (int p = 0; p < kP; ++p)
                nx[p] = *reinterpret_cast<const int4 *>(rd_slot + p * kSlotBytes);
#pragma unroll
            for (int i = 0; i < kQP; ++i) {
#pragma unroll
                for (int p = 0; p < kP; ++p) sl[p] = nx[p];
                float4 vT[kP], vB[kP];
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    vT[p] = *reinterpret_cast<const float4 *>(winj + sl[p].x);
                    vB[p] = *reinterpret_cast<const float4 *>(winj + sl[p].y);
                }
                if (i + 1 < kQP) {
#pragma unroll
                    for (int p = 0; p < kP; ++p)
                        nx[p] = *reinterpret_cast<const int4 *>(rd_slot + ((i + 1) * 16 + p) * kSlotBytes);
                }
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    const float wT = __int_as_float(sl[p].z), wB = __int_as_float(sl[p].w);
                    acc[i][0] += wT * vT[p].x + wB * vB[p].x;
                    acc[i][1] += wT * vT[p].y + wB * vB[p].y;
                    acc[i][2] += wT * vT[p].z + wB * vB[p].z;
                    acc[i][3] += wT * vT[p].w + wB * vB[p].w;
                }
            }
        } else {
            // some corner is in the image but outside the window: read both sources branch-free
            // (the LDS address is clamped, the buffer offset is out of range for in-window corners
            // and returns zeros without touching memory) and select
#pragma unroll
            for (int i = 0; i < kQP; ++i) {
#pragma unroll 2
                for (int p = 0; p < kP; ++p) {
                    const int4 sl = *reinterpret_cast<const int4 *>(rd_slot + (i * 16 + p) * kSlotBytes);
                    const float wT = __int_as_float(sl.z), wB = __int_as_float(sl.w);
                    const float4 lT = *reinterpret_cast<const float4 *>(winj + max(sl.x, 0));
                    const float4 lB = *reinterpret_cast<const float4 *>(winj + max(sl.y, 0));
                    const float4 gT = load_row4(vsrc, sl.x < 0 ? (unsigned)(sl.x & 0x7fffffff) * row_bytes + chan
                                                               : 0xffffff00u);
                    const float4 gB = load_row4(vsrc, sl.y < 0 ? (unsigned)(sl.y & 0x7fffffff) * row_bytes + chan
                                                               : 0xffffff00u);
                    const float4 vT = sl.x < 0 ? gT : lT, vB = sl.y < 0 ? gB : lB;
                    acc[i][0] += wT * vT.x + wB * vB.x;
                    acc[i][1] += wT * vT.y + wB * vB.y;
                    acc[i][2] += wT * vT.z + wB * vB.z;
                    acc[i][3] += wT * vT.w + wB * vB.w;
                }
            }
        }
        DATR_TICK(4);
        if (l + 1 < L) {
            // next level: geometry (own slots; this wave's gather reads are done), then, once every
            // wave has left the window, refill it
            wave_escaped = geometry(l + 1, lx[(l + 1) % kL], ly[(l + 1) % kL], la[(l + 1) % kL]);
            DATR_TICK(5);
            __syncthreads();
            DATR_TICK(6);
            fill_window(l + 1);
        }
    }

    // ---- left + right partial sums meet; the left row group stores the 128-B output row --------
#pragma unroll
    for (int i = 0; i < kQP; ++i) {
        float4 t;
        // ds_swizzle, bit-mask mode: lane ^ 24 (and 0x1f, or 0, xor 0x18) = the partner row group
        t.x = acc[i][0] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[i][0]), 0x601f));
        t.y = acc[i][1] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[i][1]), 0x601f));
        t.z = acc[i][2] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[i][2]), 0x601f));
        t.w = acc[i][3] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[i][3]), 0x601f));
        const int qi = wave * kQW + pair + 4 * i;
        if (col == 0 && qi < nq) {
            const int q = q_base + (qi / tw) * qW + (qi % tw);
            reinterpret_cast<float4 *>(out + (((size_t)n * Lq + q) * M + m) * D)[j] = t;
        }
    }
    DATR_TICK(7);
    DATR_TICK_FLUSH;
}

}  // namespace

extern "C" int datr_internal_msda_fwd_win_d32(
    const float *value, const float *loc, const float *attn, const DatrTiledMeta *meta, int64_t N,
    int64_t S, int64_t M, int64_t P, float *out, void *stream)
{
    const int64_t blocks = N * M * meta->total_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff || P != 4 || meta->L > 4) return DATR_EUNSUPPORTED;
    constexpr int kThreads = 512;
    int abl = 0;
#ifdef DATR_PROBE
    static const int abl_env = getenv("DATR_FWD_ABLATE") ? atoi(getenv("DATR_FWD_ABLATE")) : 0;
    abl = abl_env;
#endif
    const size_t lds = (size_t)kWinRows * 128 + (size_t)kTileQ * 4 * 32;
    hipLaunchKernelGGL(msda_fwd_win_d32<kThreads>, dim3((unsigned)blocks), dim3(kThreads), lds,
                       (hipStream_t)stream, value, loc, attn, *meta, (int)S, (int)M, abl, out);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
