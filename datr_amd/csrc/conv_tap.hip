// conv_tap.hip -- the STRIDED convolutions of the backbone / neck as exact-fp32 MFMA implicit GEMMs
// on NHWC tensors: the 3x3 / stride 2 `conv2` of layer2.0 / layer3.0 / layer4.0, the 1x1 / stride 2
// downsample convolutions beside them (/root/reference/models/dino/backbone.py:109-128 builds
// torchvision's resnet50: stride on the 3x3 of the first bottleneck of layer2-4) and the extra
// 3x3 / stride 2 pyramid level `input_proj[3]` (/root/reference/models/dino/dino.py:120-124):
// forward, data gradient and weight gradient.  These were the library's (MIOpen) launches of the step.
//
// One formulation covers forward and data gradient -- a TAP LIST:
//     out[n, oy OS + OOy, ox OS + OOx, co] = epi( sum_t sum_ci in[n, oy IS + dy_t, ox IS + dx_t, ci] Wt[w_t][ci][co] )
//   forward, stride 2:  IS = 2, OS = 1, taps (r - 1, s - 1);
//   data gradient of a stride-2 convolution: the input pixels fall into four parity classes; class
//     (py, px) receives through the taps with r = 1 - py (mod 2), s likewise -- 1, 2, 2 and 4 taps:
//     IS = 1, OS = 2, (OOy, OOx) = (py, px), dy_t = (py + 1 - r) / 2; four launches, 9 taps in all
//     (no multiplications by the zeros a dilated formulation inserts).
// Kernel `tap_conv<IS>`: pixels are MFMA rows, output channels MFMA columns (v_mfma_f32_32x32x2_f32,
// exact fp32).  Workgroup = 256 threads = 2 x 2 waves, tile = 8 x 16 positions x 128 output channels;
// K runs over 16-channel chunks and, inside a chunk, over the taps.  The input PATCH of a chunk (all
// pixels any tap of the tile touches, zero outside the image) is staged in LDS once and re-used by
// all taps; the 16 x 128 weights of the next (tap, chunk) go from L2 straight into MFMA registers while the
// current step is multiplied (round 4; a ring of LDS slabs with a barrier per step before: forward 285-297 ->
// 238-245 us on the three backbone layers).  A lane reads FOUR consecutive channels of its pixel with one ds_read_b128; with a
// row stride of 20 floats the reads of a stride-1 tile are bank-conflict free; for IS = 2 the patch
// columns are stored de-interleaved (even columns, then odd columns), so that consecutive positions
// of one tap are consecutive LDS pixels again.  Small maps (25 x 42, 13 x 21) split K over
// blockIdx.z into partial sums that `tap_fold` adds in a fixed order before the epilogue.
//
// Kernel `tap_wgrad`: dW[t][ci][co] = sum_positions in[n, oy 2 + dy_t, ox 2 + dx_t, ci] dY[n, oy, ox, co].
// Workgroup = (32 input channels, 128 output channels, a slice of the 4 x 16-position tiles); wave w
// owns output channels 32 w .. 32 w + 31 and ALL taps: 9 accumulator tiles of 32 x 32.  A k-step
// multiplies two positions 8 apart in a tile row: with pixel strides of 34 / 132 floats the two
// 128-B rows a ds_read_b32 fetches sit 32 banks apart.  Slices are summed in a fixed order by
// `tap_wgrad_fold` (deterministic, no atomics), which also writes torch's [co][ci][r][s] layout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 16;                 // positions per workgroup
constexpr int CK = 16;                         // input channels per chunk
constexpr int PSTR = 20;                       // patch pixel stride in floats (16 + 4 pad)
constexpr int kMaxTaps = 9;

struct TapClass {                               // one output class of a launch (forward: one; data gradient: four parities)
    int Hidx, Widx;                             // positions computed
    int OOy, OOx;                               // position (oy, ox) lands at (oy OS + OOy, ox OS + OOx)
    int tiles_x, tiles_y, tile0;                // its tiles are blockIdx.x in [tile0, tile0 + N tiles_x tiles_y)
    int mindy, mindx, PH, PW, PW2;              // patch geometry (PW2 = columns per parity plane)
    int ntaps;
    int toff[kMaxTaps];                         // LDS float offset of a tap relative to a lane's base
    int widx[kMaxTaps];                         // weight matrix of the tap
};
struct TapArgs {
    const float *x, *wt, *scale, *shift;
    float *y;                                   // output tensor, or the partial buffer when ksplit > 1
    int Hin, Win, Cin, Cout;
    int OS, Hout, Wout;
    int ksplit, ncls;
    float slope;
    TapClass cls[4];
};

constexpr int kPatchLoads = 9;                 // float4 per thread that cover the largest patch (17 x 33 x 4 / 256)

template <int IS, int BNT>                      // BNT = output channels per workgroup (128)
__global__ __launch_bounds__(kThreads) void tap_conv(const TapArgs a)
{
    // Dynamic LDS (it starts at address 0: the kernel has no static LDS), addressed through plain integers.
    //   patch  [PH * planes * PW2][PSTR] floats at 0
    constexpr int planes = IS;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) f4 lds_f4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int ci_ = a.ncls - 1;
    while (ci_ > 0 && (int)blockIdx.x < a.cls[ci_].tile0) --ci_;
    const TapClass &c = a.cls[ci_];
    int b = (int)blockIdx.x - c.tile0;
    const int tx = b % c.tiles_x; b /= c.tiles_x;
    const int ty = b % c.tiles_y; b /= c.tiles_y;
    const int n = b;
    constexpr int NJ = BNT / 64;                // 32-column accumulator tiles per wave
    const int co0 = blockIdx.y * BNT;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Cin = a.Cin, Cout = a.Cout;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // The weights of a (tap, chunk) step go from global memory (L2) STRAIGHT into the MFMA registers (round 4; a ring
    // of LDS slabs filled by LDS-DMA before): wave (wm, wn) needs the [16 ch][64 couts] block of its output-channel
    // half -- one float per lane and k-step, 128 contiguous bytes per 32 lanes -- and only the two waves of a column
    // half would have shared a slab, at the price of a barrier per STEP.  ONE register set, refilled k-step by k-step
    // right after its last use with the next step's values (a full step = 32 MFMAs ahead); the barrier now guards
    // the patch only (twice per chunk instead of once per tap).
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.wt), 0, kMaxTaps * Cin * Cout * 4, 0x00020000);    // (the host checks 9 Cin Cout 4 < 2^31)
    const unsigned w_voff = (unsigned)((lhi * 4 * Cout + co0 + wn * (BNT / 2) + l31) * 4);
    float bw[2][4][NJ];
    auto load_w = [&](int w, int ci0, int grp, int q) {      // channel ci0 + 8 grp + 4 lhi + q of weight matrix w
        const unsigned soff = (unsigned)(((size_t)w * Cin + ci0 + grp * 8 + q) * Cout * 4);
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
            bw[grp][q][jn] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, w_voff + 128u * jn, soff, 0));
    };
    // The input patch of a chunk: every thread fetches kPatchLoads float4 through a buffer descriptor
    // (pixels outside the image or past the patch get an out-of-range offset and read zeros: the
    // SAME number of loads for every thread and chunk, so the waits below can count them), one chunk
    // ahead into registers, and writes them to LDS when the chunk is switched.
    const int PH = c.PH, PW = c.PW, PW2 = c.PW2;
    const int gy0 = oy0 * IS + c.mindy, gx0 = ox0 * IS + c.mindx;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x + (size_t)n * a.Hin * a.Win * Cin), 0, a.Hin * a.Win * Cin * 4, 0x00020000);
    int poff[kPatchLoads], pat[kPatchLoads];
#pragma unroll
    for (int u = 0; u < kPatchLoads; ++u) {
        const int f = tid + u * kThreads;
        const int pix = f >> 2, q = f & 3;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int yy = gy0 + pr, xx = gx0 + pc;
        const bool in = f < PH * PW * 4 && yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win;
        poff[u] = in ? ((yy * a.Win + xx) * Cin + q * 4) * 4 : (int)0x80000000;
        const int at = IS == 2 ? ((pr * 2 + (pc & 1)) * PW2 + (pc >> 1)) : (pr * PW2 + pc);
        pat[u] = f < PH * PW * 4 ? at * PSTR + q * 4 : -1;
    }
    f4 pre[kPatchLoads];
    auto fetch_patch = [&](int ci0) {
#pragma unroll
        for (int u = 0; u < kPatchLoads; ++u) {
            const auto r = __builtin_amdgcn_raw_buffer_load_b128(xr, poff[u], ci0 * 4, 0);
            pre[u] = __builtin_bit_cast(f4, r);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int u = 0; u < kPatchLoads; ++u)
            if (pat[u] >= 0) *reinterpret_cast<lds_f4 *>((unsigned)pat[u] * 4u) = pre[u];
    };

    // A-operand base of pixel block i of this wave: rows 2 i, 2 i + 1 of its 4 position rows
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = wm * 4 + i * 2 + (l31 >> 4), px = l31 & 15;
        abase[i] = (py * IS * planes * PW2 + px) * PSTR + lhi * 4;
    }

    const int nchunks = Cin / CK;
    const int ch0 = (int)((long)blockIdx.z * nchunks / a.ksplit), ch1 = (int)((long)(blockIdx.z + 1) * nchunks / a.ksplit);
    const int ntaps = c.ntaps;
    fetch_patch(ch0 * CK);
    if (ch0 < ch1) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int q = 0; q < 4; ++q) load_w(c.widx[0], ch0 * CK, grp, q);
    }
    store_patch();                              // (the compiler waits for the fetched registers here)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int ch = ch0; ch < ch1; ++ch) {
        if (ch + 1 < ch1) fetch_patch((ch + 1) * CK);          // lands while this chunk's taps are multiplied
#pragma unroll 1
        for (int t = 0; t < ntaps; ++t) {
            // the step after this one: next tap of the chunk, or the first tap of the next chunk
            const bool last_tap = t + 1 == ntaps;
            const bool more = !(last_tap && ch + 1 == ch1);
            const int wn_ = c.widx[last_tap ? 0 : t + 1], cn_ = (last_tap ? ch + 1 : ch) * CK;
            const int toff = c.toff[t];
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {          // two 8-channel groups of the chunk
                f4 av[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    av[i] = *reinterpret_cast<const lds_f4 *>((unsigned)(abase[i] + toff + grp * 8) * 4u);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = q == 0 ? av[0].x : q == 1 ? av[0].y : q == 2 ? av[0].z : av[0].w;
                    const float a1 = q == 0 ? av[1].x : q == 1 ? av[1].y : q == 2 ? av[1].z : av[1].w;
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn) {
                        acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bw[grp][q][jn], acc[0][jn], 0, 0, 0);
                        acc[1][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bw[grp][q][jn], acc[1][jn], 0, 0, 0);
                    }
                    if (more) load_w(wn_, cn_, grp, q);
                }
            }
        }
        if (ch + 1 < ch1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the chunk's readers are done with the patch
            store_patch();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the next chunk's patch is in place
        }
    }

    // ---- epilogue: 32 lanes = 32 consecutive output channels (128 B) -----------------------------
    const bool raw = a.ksplit > 1;
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (!raw) {
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int ch = co0 + wn * (BNT / 2) + jn * 32 + l31;
            if (a.scale) sc[jn] = a.scale[ch];
            if (a.shift) sh[jn] = a.shift[ch];
        }
    }
    const float slope = raw ? 1.f : a.slope;
    // Stores through a buffer descriptor with 32-bit offsets (the host checks that the output and the partial buffer
    // stay below 2 GiB): a position outside the map gets an out-of-range offset and its store is dropped -- no branch
    // per position -- and scale / shift are settled before the first store.  With bounds branches around the stores the
    // compiler waited for everything in flight, the previous store included, ahead of every position (s_waitcnt vmcnt(0)
    // 23 times per lane).
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0x7ffffffc, 0x00020000);
    const unsigned ch_off = (unsigned)(co0 + wn * (BNT / 2) + l31) * 4u;
    // output pixel of position (oy, ox) = pix0 + oy pix_sy + ox pix_sx: the partial buffer [z][n][Hidx][Widx] of a split
    // launch (single-class launches only) or the strided place in the output tensor
    const int nimg = (int)gridDim.x / (c.tiles_x * c.tiles_y);
    const int pix0 = raw ? ((int)blockIdx.z * nimg + n) * c.Hidx * c.Widx : (n * a.Hout + c.OOy) * a.Wout + c.OOx;
    const int pix_sy = raw ? c.Widx : a.OS * a.Wout, pix_sx = raw ? 1 : a.OS;
    const int Hi = c.Hidx, Wi = c.Widx;
    asm volatile("" : : "v"(sc[0]), "v"(sc[1]), "v"(sh[0]), "v"(sh[1]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = (e & 3) + 8 * (e >> 2) + 4 * lhi;          // position within the 32-block
            const int py = wm * 4 + i * 2 + (prow >> 4), px = prow & 15;
            const int oy = oy0 + py, ox = ox0 + px;
            const unsigned pix = (unsigned)(pix0 + oy * pix_sy + ox * pix_sx);
            const unsigned off = ((oy < Hi) & (ox < Wi)) ? pix * (unsigned)Cout * 4u + ch_off : 0x80000000u;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                float v = acc[i][jn][e] * sc[jn] + sh[jn];
                v = v > 0.f ? v : v * slope;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ysrc, off, jn * 128, 0);
            }
        }
    }
}

// out = epi(sum_z partial[z]) for the split-K launches; a thread per 4 channels
__global__ __launch_bounds__(256) void tap_fold(const float *__restrict__ partial, int ksplit, long per_split,
                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                float slope, int N, int Hidx, int Widx, int Cout, int OS, int OOy,
                                                int OOx, int Hout, int Wout, float *__restrict__ y)
{
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= per_split) return;
    float4 s = *reinterpret_cast<const float4 *>(partial + i4 * 4);
    for (int z = 1; z < ksplit; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (size_t)z * per_split + i4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int c = (int)((i4 * 4) % Cout);
    long p = (i4 * 4) / Cout;
    const int ox = (int)(p % Widx); p /= Widx;
    const int oy = (int)(p % Hidx); p /= Hidx;
    const int n = (int)p;
    float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (scale) v[k] *= scale[c + k];
        if (shift) v[k] += shift[c + k];
        v[k] = v[k] > 0.f ? v[k] : v[k] * slope;
    }
    *reinterpret_cast<float4 *>(y + (((size_t)n * Hout + oy * OS + OOy) * Wout + ox * OS + OOx) * Cout + c) =
        make_float4(v[0], v[1], v[2], v[3]);
}

// K split of a launch.  The tiles of a map are few (4 x 50 x 84 outputs = 168 tiles x 2 channel blocks)
// and a CU's matrix pipes are shared by its two resident workgroups: 336 workgroups leave 176 CUs with
// one workgroup (done in half the time, then idle) -- measured 357 us against 286 us with K split
// in two (672 workgroups) for layer3.0.conv2; 401 -> 279 us with K split in four for layer4.0.conv2
// (192 tiles); splitting further, or 64-channel workgroups instead, was slower (tools/probes/
// conv_tap_plans.sh).  The partial sums are added in a fixed order (tap_fold).
int pick_ksplit(long wgs, long nchunks) {
    if (const char *f = getenv("DATR_TAP_PLAN")) {          // development: "128,ksplit"
        int bn = 0, ks = 0;
        if (sscanf(f, "%d,%d", &bn, &ks) == 2 && bn == 128 && ks >= 1 && ks <= 16 && ks <= nchunks) return ks;
    }
    int ks = 1;
    while (wgs * ks < 512 && ks * 2 <= nchunks / 2 && ks < 16) ks *= 2;
    return ks;
}

// One launch over `ncls` output classes that share input, weights and epilogue.
struct TapClassSpec { int Hidx, Widx, OOy, OOx, ntaps; int taps[kMaxTaps][3]; };

int launch_taps(int IS, const float *x, const float *wt, const float *scale, const float *shift, float slope,
                int N, int Hin, int Win, int Cin, int Cout, int OS, int Hout, int Wout, int ncls,
                const TapClassSpec *specs, float *y, float *partial, long partial_floats, hipStream_t st)
{
    TapArgs a{};
    a.x = x; a.wt = wt; a.scale = scale; a.shift = shift; a.slope = slope;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout;
    a.OS = OS; a.Hout = Hout; a.Wout = Wout;
    int tiles = 0, patch_floats = 0, nc = 0;
    for (int k = 0; k < ncls; ++k) {
        const TapClassSpec &sp = specs[k];
        if (sp.Hidx <= 0 || sp.Widx <= 0 || sp.ntaps <= 0) continue;
        TapClass &c = a.cls[nc++];
        c.Hidx = sp.Hidx; c.Widx = sp.Widx; c.OOy = sp.OOy; c.OOx = sp.OOx;
        c.tiles_x = (sp.Widx + TW - 1) / TW; c.tiles_y = (sp.Hidx + TH - 1) / TH;
        c.tile0 = tiles;
        tiles += N * c.tiles_x * c.tiles_y;
        int mindy = 1 << 20, mindx = 1 << 20, maxdy = -(1 << 20), maxdx = -(1 << 20);
        for (int t = 0; t < sp.ntaps; ++t) {
            mindy = std::min(mindy, sp.taps[t][0]); maxdy = std::max(maxdy, sp.taps[t][0]);
            mindx = std::min(mindx, sp.taps[t][1]); maxdx = std::max(maxdx, sp.taps[t][1]);
        }
        c.mindy = mindy; c.mindx = mindx;
        c.PH = (TH - 1) * IS + (maxdy - mindy) + 1;
        c.PW = (TW - 1) * IS + (maxdx - mindx) + 1;
        c.PW2 = IS == 2 ? (c.PW + 1) / 2 : c.PW;
        c.ntaps = sp.ntaps;
        for (int t = 0; t < sp.ntaps; ++t) {
            const int ddy = sp.taps[t][0] - mindy, ddx = sp.taps[t][1] - mindx;
            c.toff[t] = IS == 2 ? ((ddy * 2 + (ddx & 1)) * c.PW2 + (ddx >> 1)) * PSTR : (ddy * c.PW2 + ddx) * PSTR;
            c.widx[t] = sp.taps[t][2];
        }
        if (c.PH * c.PW * 4 > kPatchLoads * kThreads) return DATR_EUNSUPPORTED;
        patch_floats = std::max(patch_floats, (c.PH * IS * c.PW2 * PSTR + 3) & ~3);
    }
    if (nc == 0 || N == 0) return DATR_OK;
    a.ncls = nc;
    // K is split for single-class launches only (the partial buffer is laid out for one class)
    int ksplit = nc == 1 ? pick_ksplit((long)tiles * (Cout / 128), Cin / CK) : 1;
    const long per_split = (long)N * a.cls[0].Hidx * a.cls[0].Widx * Cout;
    if (ksplit > 1 && (!partial || partial_floats < per_split * ksplit)) ksplit = 1;
    a.ksplit = ksplit;
    a.y = ksplit > 1 ? partial : y;
    // the kernel's stores use 32-bit byte offsets
    if ((ksplit > 1 ? per_split * ksplit : (long)N * Hout * Wout * Cout) * 4 >= (1L << 31)) return DATR_EUNSUPPORTED;
    const size_t lds = (size_t)patch_floats * sizeof(float);
    dim3 grid((unsigned)tiles, (unsigned)(Cout / 128), (unsigned)ksplit);
    auto go = [&](auto kernel) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
        if (!ok) return false;
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, a);
        return true;
    };
    const bool launched = IS == 2 ? go(tap_conv<2, 128>) : go(tap_conv<1, 128>);
    if (!launched) return DATR_EUNSUPPORTED;
    if (ksplit > 1) {
        const TapClass &c = a.cls[0];
        const long n4 = per_split / 4;
        hipLaunchKernelGGL(tap_fold, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, partial, ksplit, per_split,
                           scale, shift, slope, N, c.Hidx, c.Widx, Cout, OS, c.OOy, c.OOx, Hout, Wout, y);
    }
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
constexpr int WTH = 4, WTW = 16;               // positions per tile
constexpr int WCI = 32, WCO = 128;             // channels per workgroup
constexpr int XSTR = 34, DSTR = 132;           // LDS pixel strides (floats)

struct WgradArgs {
    const float *x, *dy;
    float *partial;
    int N, Hin, Win, Cin, Ho, Wo, Cout;
    int tiles_x, tiles_y, slices;
};

__global__ __launch_bounds__(kThreads, 2) void tap_wgrad(const WgradArgs a)
{
    constexpr int KS = 3, NT = KS * KS;
    constexpr int PH = (WTH - 1) * 2 + KS, PW = (WTW - 1) * 2 + KS;
    constexpr int pad = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;                                   // [PH * PW][XSTR]
    float *ds = smem + ((PH * PW * XSTR + 3) & ~3);     // [64][DSTR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ci0 = blockIdx.x * WCI, co0 = blockIdx.y * WCO, slice = blockIdx.z;
    const int Cin = a.Cin, Cout = a.Cout;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // A tile's input patch and dY block travel global -> registers (one tile AHEAD, while the current
    // tile is multiplied) -> LDS.  Buffer loads: pixels outside the image / map get an out-of-range
    // offset and read zeros, so every thread issues the same loads for every tile.
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int kXLoads = (PH * PW * 8 + kThreads - 1) / kThreads, kDLoads = WTH * WTW * 32 / kThreads;
    f4 rx[kXLoads], rd[kDLoads];
    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    auto fetch = [&](int tile) {
        int b = tile;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y; b /= a.tiles_y;
        const int n = b;
        const int oy0 = ty * WTH, ox0 = tx * WTW;
        const int gy0 = oy0 * 2 - pad, gx0 = ox0 * 2 - pad;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.x + (size_t)n * a.Hin * a.Win * Cin), 0, a.Hin * a.Win * Cin * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.dy + (size_t)n * a.Ho * a.Wo * Cout), 0, a.Ho * a.Wo * Cout * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < kXLoads; ++u) {
            const int f = tid + u * kThreads;
            const int pix = f >> 3, q = f & 7;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int yy = gy0 + pr, xx = gx0 + pc;
            const bool in = f < PH * PW * 8 && yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win;
            const int off = in ? ((yy * a.Win + xx) * Cin + ci0 + q * 4) * 4 : (int)0x80000000;
            rx[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < kDLoads; ++u) {
            const int f = tid + u * kThreads;
            const int pos = f >> 5, q = f & 31;
            const int oy = oy0 + (pos >> 4), ox = ox0 + (pos & 15);
            const bool in = oy < a.Ho && ox < a.Wo;
            const int off = in ? ((oy * a.Wo + ox) * Cout + co0 + q * 4) * 4 : (int)0x80000000;
            rd[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(dr, off, 0, 0));
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < kXLoads; ++u) {
            const int f = tid + u * kThreads;
            if (f < PH * PW * 8) {
                float2 *d = reinterpret_cast<float2 *>(&xs[(f >> 3) * XSTR + (f & 7) * 4]);      // 136-B pixels: 8-B aligned
                d[0] = make_float2(rx[u].x, rx[u].y);
                d[1] = make_float2(rx[u].z, rx[u].w);
            }
        }
#pragma unroll
        for (int u = 0; u < kDLoads; ++u) {
            const int f = tid + u * kThreads;
            *reinterpret_cast<f4 *>(&ds[(f >> 5) * DSTR + (f & 31) * 4]) = rd[u];
        }
    };
    if (slice < ntiles) fetch(slice);
    for (int tile = slice; tile < ntiles; tile += a.slices) {
        __syncthreads();                                // the previous tile's readers are done
        stash();
        __syncthreads();
        if (tile + a.slices < ntiles) fetch(tile + a.slices);
        // 32 k-steps: positions (py, pxl) and (py, pxl + 8)
#pragma unroll 1
        for (int s = 0; s < 32; ++s) {
            const int py = s >> 3, px = (s & 7) + 8 * lhi;
            const float bv = ds[(py * 16 + px) * DSTR + wave * 32 + l31];
            const float *xp = &xs[((py * 2) * PW + px * 2) * XSTR + l31];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int r = t / KS, c = t - r * KS;
                const float av = xp[(r * PW + c) * XSTR];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // partial[slice][t][ci][co]
    float *pb = a.partial + (size_t)slice * NT * Cin * Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            pb[((size_t)t * Cin + ci0 + row) * Cout + co0 + wave * 32 + l31] = acc[t][e];
        }
}

// dw[co][ci][r][s] = sum over slices; a thread per (ci, co) with co fastest in the partials
__global__ __launch_bounds__(256) void tap_wgrad_fold(const float *__restrict__ partial, int slices, int Cin, int Cout,
                                                      float *__restrict__ dw, int64_t s_co, int64_t s_ci,
                                                      int64_t s_r, int64_t s_s)
{
    constexpr int NT = 9, KS = 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)NT * Cin * Cout) return;
    const int co = (int)(idx % Cout);
    const int ci = (int)((idx / Cout) % Cin);
    const int t = (int)(idx / ((long)Cout * Cin));
    float s = 0.f;
    for (int z = 0; z < slices; ++z) s += partial[(size_t)z * NT * Cin * Cout + idx];
    dw[co * s_co + ci * s_ci + (t / KS) * s_r + (t % KS) * s_s] = s;
}

// Both filter layouts of a launch pair from torch's [co][ci][3][3] in ONE pass: wt[t][ci][co] for the
// forward, wt_t[t][co][ci] for the data gradient.  A 32 x 32 (co, ci) tile per workgroup goes through
// LDS so that both outputs are written in 128-B runs (ATen's strided permute copy took 60-100 us per
// layout for the 2048 x 256 filter).
__global__ __launch_bounds__(256) void tap_weights(const float *__restrict__ w, int Cout, int Cin, int64_t s_co,
                                                   int64_t s_ci, int64_t s_r, int64_t s_s, float *__restrict__ wt,
                                                   float *__restrict__ wt_t)
{
    __shared__ float tile[9][32][33];
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int a = threadIdx.x >> 5, b = threadIdx.x & 31;          // 8 x 32
    for (int i = a; i < 32; i += 8) {                               // i = co row, b = ci column
        const float *src = w + (co0 + i) * s_co + (ci0 + b) * s_ci;
#pragma unroll
        for (int t = 0; t < 9; ++t) tile[t][i][b] = src[(t / 3) * s_r + (t % 3) * s_s];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t)
        for (int i = a; i < 32; i += 8) {
            if (wt_t) wt_t[((size_t)t * Cout + co0 + i) * Cin + ci0 + b] = tile[t][i][b];
            if (wt) wt[((size_t)t * Cin + ci0 + i) * Cout + co0 + b] = tile[t][b][i];
        }
}

int wgrad_slices(long N, long Ho, long Wo, long Cin, long Cout) {
    const long tiles = N * ((Ho + WTH - 1) / WTH) * ((Wo + WTW - 1) / WTW);
    const long blocks = (Cin / WCI) * (Cout / WCO);
    long s = (512 + blocks - 1) / blocks;               // ~2 workgroups per CU
    s = std::min<long>(s, std::max<long>(1, tiles / 4));     // at least 4 tiles per slice
    return (int)std::max<long>(1, std::min<long>(s, 128));
}

}  // namespace

extern "C" {

int64_t datr_conv3x3s2_workspace_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return -1;
    const int64_t Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t tiles = N * ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH);
    int64_t kf = pick_ksplit(tiles * (Cout / 128), Cin / CK);
    if (getenv("DATR_TAP_PLAN")) kf = 16;
    const int64_t fwd = kf > 1 ? kf * N * Ho * Wo * Cout : 0;            // split-K partial sums of the forward
    const int64_t wgr = (int64_t)wgrad_slices(N, Ho, Wo, Cin, Cout) * 9 * Cin * Cout;
    return std::max(fwd, wgr);
}

int datr_conv3x3s2_weights_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci, int64_t s_r,
                               int64_t s_s, float *wt, float *wt_t, void *stream)
{
    if (!w || (!wt && !wt_t) || Cout < 1 || Cin < 1) return DATR_EINVAL;
    if (Cout % 32 || Cin % 32) return DATR_EUNSUPPORTED;
    hipLaunchKernelGGL(tap_weights, dim3((unsigned)(Cin / 32), (unsigned)(Cout / 32)), dim3(256), 0, (hipStream_t)stream, w,
                       (int)Cout, (int)Cin, s_co, s_ci, s_r, s_s, wt, wt_t);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

int datr_conv3x3s2_forward_nhwc_f32(const float *x, const float *wt, const float *scale, const float *shift,
                                    float slope, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                    float *y, float *workspace, int64_t workspace_floats, void *stream)
{
    if (!x || !wt || !y || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if (Cin % CK || Cout % 128) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x1fffffffLL) return DATR_EUNSUPPORTED;        // 32-bit byte offsets
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    TapClassSpec sp{};
    sp.Hidx = Ho; sp.Widx = Wo; sp.OOy = sp.OOx = 0;
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            sp.taps[sp.ntaps][0] = r - 1; sp.taps[sp.ntaps][1] = s - 1; sp.taps[sp.ntaps][2] = r * 3 + s;
            ++sp.ntaps;
        }
    return launch_taps(2, x, wt, scale, shift, slope, (int)N, (int)H, (int)W, (int)Cin, (int)Cout, 1, Ho, Wo, 1, &sp, y,
                       workspace, workspace_floats, (hipStream_t)stream);
}

int datr_conv3x3s2_dgrad_nhwc_f32(const float *dy, const float *wt_t, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                  int64_t Cout, float *dx, float *workspace, int64_t workspace_floats, void *stream)
{
    if (!dy || !wt_t || !dx || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if (Cout % CK || Cin % 128) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x1fffffffLL) return DATR_EUNSUPPORTED;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    // the four parity classes of the input pixels in ONE launch (their workgroups fill the machine
    // together: no K split, no fold).  Input pixel (2 a + py, 2 b + px) was seen by output
    // (a + dy_t, b + dx_t) through tap (r, s) with 2 (a + dy_t) + r - 1 = 2 a + py.
    TapClassSpec sp[4] = {};
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            TapClassSpec &c = sp[py * 2 + px];
            c.Hidx = (int)((H - py + 1) / 2); c.Widx = (int)((W - px + 1) / 2); c.OOy = py; c.OOx = px;
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) {
                    if (((py + 1 - r) & 1) || ((px + 1 - s) & 1)) continue;
                    c.taps[c.ntaps][0] = (py + 1 - r) / 2; c.taps[c.ntaps][1] = (px + 1 - s) / 2;
                    c.taps[c.ntaps][2] = r * 3 + s;
                    ++c.ntaps;
                }
        }
    (void)workspace; (void)workspace_floats;
    return launch_taps(1, dy, wt_t, nullptr, nullptr, 1.f, (int)N, Ho, Wo, (int)Cout, (int)Cin, 2, (int)H, (int)W, 4, sp, dx,
                       nullptr, 0, (hipStream_t)stream);
}

int datr_conv3x3s2_wgrad_nhwc_f32(const float *x, const float *dy, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                  int64_t Cout, float *dw, int64_t s_co, int64_t s_ci, int64_t s_r, int64_t s_s,
                                  float *workspace, int64_t workspace_floats, void *stream)
{
    if (!x || !dy || !dw || !workspace || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if (Cin % WCI || Cout % WCO) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x1fffffffLL) return DATR_EUNSUPPORTED;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    WgradArgs a{};
    a.x = x; a.dy = dy; a.partial = workspace;
    a.N = (int)N; a.Hin = (int)H; a.Win = (int)W; a.Cin = (int)Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = (int)Cout;
    a.tiles_x = (Wo + WTW - 1) / WTW; a.tiles_y = (Ho + WTH - 1) / WTH;
    a.slices = wgrad_slices(N, Ho, Wo, Cin, Cout);
    if (workspace_floats < (int64_t)a.slices * 9 * Cin * Cout) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(Cin / WCI), (unsigned)(Cout / WCO), (unsigned)a.slices);
    if (N > 0) {
        constexpr int PH = (WTH - 1) * 2 + 3, PW = (WTW - 1) * 2 + 3;
        const size_t lds = (size_t)(((PH * PW * XSTR + 3) & ~3) + WTH * WTW * DSTR) * sizeof(float);
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(tap_wgrad),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
        if (!ok) return DATR_EUNSUPPORTED;
        hipLaunchKernelGGL(tap_wgrad, grid, dim3(kThreads), lds, st, a);
    } else if (hipMemsetAsync(workspace, 0, (size_t)a.slices * 9 * Cin * Cout * sizeof(float), st) != hipSuccess) {
        return DATR_ELAUNCH;
    }
    const long total = 9L * Cin * Cout;
    hipLaunchKernelGGL(tap_wgrad_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, a.slices,
                       (int)Cin, (int)Cout, dw, s_co, s_ci, s_r, s_s);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

}  // extern "C"
