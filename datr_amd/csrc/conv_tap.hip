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
// all taps; the 16 x 128 weight slab of the next (tap, chunk) arrives by LDS-DMA while the current one
// is multiplied.  A lane reads FOUR consecutive channels of its pixel with one ds_read_b128; with a
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

#include "datr_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 16;                 // positions per workgroup
constexpr int CK = 16;                         // input channels per chunk
constexpr int PSTR = 20;                       // patch pixel stride in floats (16 + 4 pad)
constexpr int BN = 128;                        // output channels per workgroup
constexpr int kMaxTaps = 9;

struct TapArgs {
    const float *x, *wt, *scale, *shift;
    float *y;                                   // output tensor, or the partial buffer when ksplit > 1
    int Hin, Win, Cin, Cout;
    int Hidx, Widx;                             // positions computed
    int OS, OOy, OOx, Hout, Wout;               // where position (oy, ox) lands in the output tensor
    int tiles_x, tiles_y, ksplit;
    int mindy, mindx, PH, PW, PW2;              // patch geometry (PW2 = columns per parity plane)
    int ntaps;
    int toff[kMaxTaps];                         // LDS float offset of a tap relative to a lane's base
    int widx[kMaxTaps];                         // weight matrix of the tap
    float slope;
};

template <int IS>
__global__ __launch_bounds__(kThreads) void tap_conv(const TapArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                                            // PH * planes * PW2 * PSTR
    constexpr int planes = IS;
    float (*wsl)[CK][BN] = reinterpret_cast<float (*)[CK][BN]>(smem + ((a.PH * planes * a.PW2 * PSTR + 3) & ~3));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int co0 = blockIdx.y * BN;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const float *Xn = a.x + (size_t)n * a.Hin * a.Win * a.Cin;
    const int Cin = a.Cin, Cout = a.Cout;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // weight slab [CK][BN] of (tap, chunk): 16 rows of 512 contiguous bytes -> LDS by LDS-DMA
    auto dma_w = [&](int w, int ci0, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (wave * 2 + u) * 2 + (lane >> 5);            // 0..15
            const float *src = a.wt + ((size_t)w * Cin + ci0 + row) * Cout + co0 + (lane & 31) * 4;
            __builtin_amdgcn_global_load_lds(
                (__attribute__((address_space(1))) const void *)src,
                (__attribute__((address_space(3))) void *)&wsl[buf][(wave * 2 + u) * 2][0], 16, 0, 0);
        }
    };
    const int PH = a.PH, PW = a.PW, PW2 = a.PW2;
    const int gy0 = oy0 * IS + a.mindy, gx0 = ox0 * IS + a.mindx;
    auto load_patch = [&](int ci0) {
        for (int f = tid; f < PH * PW * 4; f += kThreads) {
            const int pix = f >> 2, q = f & 3;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int yy = gy0 + pr, xx = gx0 + pc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win)
                v = *reinterpret_cast<const float4 *>(Xn + ((size_t)yy * a.Win + xx) * Cin + ci0 + q * 4);
            const int at = IS == 2 ? ((pr * 2 + (pc & 1)) * PW2 + (pc >> 1)) : (pr * PW2 + pc);
            *reinterpret_cast<float4 *>(&patch[at * PSTR + q * 4]) = v;
        }
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
    const int ntaps = a.ntaps;
    int step = 0;
    if (ch0 < ch1) dma_w(a.widx[0], ch0 * CK, 0);
    for (int ch = ch0; ch < ch1; ++ch) {
        const int ci0 = ch * CK;
        __syncthreads();                        // previous chunk's readers are done with the patch
        load_patch(ci0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < ntaps; ++t, ++step) {
            const int buf = step & 1;
            const bool last_tap = t + 1 == ntaps;
            const bool more = !last_tap || ch + 1 < ch1;
            if (more) dma_w(a.widx[last_tap ? 0 : t + 1], last_tap ? ci0 + CK : ci0, buf ^ 1);
            const int toff = a.toff[t];
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {          // two 8-channel groups of the chunk
                float4 av[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    av[i] = *reinterpret_cast<const float4 *>(&patch[abase[i] + toff + grp * 8]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = grp * 8 + lhi * 4 + s;
                    const float b0 = wsl[buf][k][wn * 64 + l31];
                    const float b1 = wsl[buf][k][wn * 64 + 32 + l31];
                    const float a0 = s == 0 ? av[0].x : s == 1 ? av[0].y : s == 2 ? av[0].z : av[0].w;
                    const float a1 = s == 0 ? av[1].x : s == 1 ? av[1].y : s == 2 ? av[1].z : av[1].w;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            if (more && !last_tap) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next slab landed (this wave's part)
                __syncthreads();
            }
        }
    }

    // ---- epilogue: 32 lanes = 32 consecutive output channels (128 B) -----------------------------
    const bool raw = a.ksplit > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = (e & 3) + 8 * (e >> 2) + 4 * lhi;          // position within the 32-block
            const int py = wm * 4 + i * 2 + (prow >> 4), px = prow & 15;
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy < a.Hidx && ox < a.Widx) {
                float *yb;
                if (raw)
                    yb = a.y + ((((size_t)blockIdx.z * gridDim.x / (a.tiles_x * a.tiles_y) + n) * a.Hidx + oy) * a.Widx + ox) * Cout;
                else
                    yb = a.y + (((size_t)n * a.Hout + oy * a.OS + a.OOy) * a.Wout + ox * a.OS + a.OOx) * Cout;
                yb += co0 + wn * 64 + l31;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    float v = acc[i][jn][e];
                    if (!raw) {
                        const int c = co0 + wn * 64 + jn * 32 + l31;
                        if (a.scale) v *= a.scale[c];
                        if (a.shift) v += a.shift[c];
                        v = v > 0.f ? v : v * a.slope;
                    }
                    yb[jn * 32] = v;
                }
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

// split K when the tiles do not fill the machine (deterministic: partial sums + fold)
int pick_ksplit(long N, long Hidx, long Widx, long Cin, long Cout) {
    const long wgs = N * ((Widx + TW - 1) / TW) * ((Hidx + TH - 1) / TH) * (Cout / BN);
    const long nchunks = Cin / CK;
    int ksplit = 1;
    while (wgs * ksplit < 384 && ksplit * 2 <= nchunks / 4 && ksplit < 16) ksplit *= 2;
    return ksplit;
}

int launch_taps(int IS, const float *x, const float *wt, const float *scale, const float *shift, float slope,
                int N, int Hin, int Win, int Cin, int Cout, int Hidx, int Widx, int OS, int OOy, int OOx, int Hout,
                int Wout, int ntaps, const int (*taps)[3], float *y, float *partial, long partial_floats,
                hipStream_t st)
{
    if (Hidx <= 0 || Widx <= 0) return DATR_OK;
    TapArgs a{};
    a.x = x; a.wt = wt; a.scale = scale; a.shift = shift; a.slope = slope;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.Hidx = Hidx; a.Widx = Widx;
    a.OS = OS; a.OOy = OOy; a.OOx = OOx; a.Hout = Hout; a.Wout = Wout;
    a.tiles_x = (Widx + TW - 1) / TW; a.tiles_y = (Hidx + TH - 1) / TH;
    int mindy = 1 << 20, mindx = 1 << 20, maxdy = -(1 << 20), maxdx = -(1 << 20);
    for (int t = 0; t < ntaps; ++t) {
        mindy = std::min(mindy, taps[t][0]); maxdy = std::max(maxdy, taps[t][0]);
        mindx = std::min(mindx, taps[t][1]); maxdx = std::max(maxdx, taps[t][1]);
    }
    a.mindy = mindy; a.mindx = mindx;
    a.PH = (TH - 1) * IS + (maxdy - mindy) + 1;
    a.PW = (TW - 1) * IS + (maxdx - mindx) + 1;
    a.PW2 = IS == 2 ? (a.PW + 1) / 2 : a.PW;
    a.ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) {
        const int ddy = taps[t][0] - mindy, ddx = taps[t][1] - mindx;
        a.toff[t] = IS == 2 ? ((ddy * 2 + (ddx & 1)) * a.PW2 + (ddx >> 1)) * PSTR : (ddy * a.PW2 + ddx) * PSTR;
        a.widx[t] = taps[t][2];
    }
    int ksplit = pick_ksplit(N, Hidx, Widx, Cin, Cout);
    const long per_split = (long)N * Hidx * Widx * Cout;
    if (ksplit > 1 && (!partial || partial_floats < per_split * ksplit)) ksplit = 1;
    a.ksplit = ksplit;
    a.y = ksplit > 1 ? partial : y;
    const size_t lds = (size_t)(((a.PH * IS * a.PW2 * PSTR + 3) & ~3) + 2 * CK * BN) * sizeof(float);
    dim3 grid((unsigned)(N * a.tiles_x * a.tiles_y), (unsigned)(Cout / BN), (unsigned)ksplit);
    if (IS == 2) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(tap_conv<2>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
        if (!ok) return DATR_EUNSUPPORTED;
        hipLaunchKernelGGL(tap_conv<2>, grid, dim3(kThreads), lds, st, a);
    } else {
        hipLaunchKernelGGL(tap_conv<1>, grid, dim3(kThreads), lds, st, a);
    }
    if (ksplit > 1) {
        const long n4 = per_split / 4;
        hipLaunchKernelGGL(tap_fold, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, partial, ksplit, per_split,
                           scale, shift, slope, N, Hidx, Widx, Cout, OS, OOy, OOx, Hout, Wout, y);
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
    int ksize;                                  // 3 (pad 1) or 1 (pad 0); stride 2
};

template <int KS>
__global__ __launch_bounds__(kThreads, 2) void tap_wgrad(const WgradArgs a)
{
    constexpr int NT = KS * KS;
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

    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    for (int tile = slice; tile < ntiles; tile += a.slices) {
        int b = tile;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y; b /= a.tiles_y;
        const int n = b;
        const int oy0 = ty * WTH, ox0 = tx * WTW;
        const int gy0 = oy0 * 2 - pad, gx0 = ox0 * 2 - pad;
        const float *Xn = a.x + (size_t)n * a.Hin * a.Win * Cin + ci0;
        const float *Dn = a.dy + (size_t)n * a.Ho * a.Wo * Cout + co0;
        __syncthreads();                                // the previous tile's readers are done
        for (int f = tid; f < PH * PW * 8; f += kThreads) {
            const int pix = f >> 3, q = f & 7;
            const int pr = pix / PW, pc = pix - pr * PW;
            const int yy = gy0 + pr, xx = gx0 + pc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win)
                v = *reinterpret_cast<const float4 *>(Xn + ((size_t)yy * a.Win + xx) * Cin + q * 4);
            float2 *d = reinterpret_cast<float2 *>(&xs[pix * XSTR + q * 4]);      // 136-B pixels: 8-B aligned
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
        for (int f = tid; f < WTH * WTW * 32; f += kThreads) {
            const int pos = f >> 5, q = f & 31;
            const int oy = oy0 + (pos >> 4), ox = ox0 + (pos & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < a.Ho && ox < a.Wo)
                v = *reinterpret_cast<const float4 *>(Dn + ((size_t)oy * a.Wo + ox) * Cout + q * 4);
            *reinterpret_cast<float4 *>(&ds[pos * DSTR + q * 4]) = v;
        }
        __syncthreads();
        // 32 k-steps: positions (py, pxl) and (py, pxl + 8)
#pragma unroll 2
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
__global__ __launch_bounds__(256) void tap_wgrad_fold(const float *__restrict__ partial, int slices, int NT, int Cin,
                                                      int Cout, float *__restrict__ dw, int64_t s_co, int64_t s_ci,
                                                      int64_t s_r, int64_t s_s, int KS)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)NT * Cin * Cout) return;
    const int co = (int)(idx % Cout);
    const int ci = (int)((idx / Cout) % Cin);
    const int t = (int)(idx / ((long)Cout * Cin));
    float s = 0.f;
    for (int z = 0; z < slices; ++z) s += partial[(size_t)z * NT * Cin * Cout + idx];
    dw[co * s_co + ci * s_ci + (t / KS) * s_r + (t % KS) * s_s] = s;
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

int64_t datr_conv_s2_workspace_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t ksize) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return -1;
    const int64_t Ho = (H + 1) / 2, Wo = (W + 1) / 2;       // both kernel sizes: floor((H + 2 pad - k) / 2) + 1
    const int64_t kf = pick_ksplit(N, Ho, Wo, Cin, Cout), kd = pick_ksplit(N, Ho, Wo, Cout, Cin);
    const int64_t fwd = kf > 1 ? kf * N * Ho * Wo * Cout : 0;            // split-K partial sums
    const int64_t dgr = kd > 1 ? kd * N * Ho * Wo * Cin : 0;             // per parity class (<= Ho x Wo positions)
    const int64_t wgr = (int64_t)wgrad_slices(N, Ho, Wo, Cin, Cout) * ksize * ksize * Cin * Cout;
    return std::max(fwd, std::max(dgr, wgr));
}

int datr_conv_s2_forward_nhwc_f32(const float *x, const float *wt, const float *scale, const float *shift,
                                  float slope, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                  int64_t ksize, float *y, float *workspace, int64_t workspace_floats, void *stream)
{
    if (!x || !wt || !y || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if ((ksize != 1 && ksize != 3) || Cin % CK || Cout % BN) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    if (N == 0) return DATR_OK;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    int taps[9][3], nt = 0;
    if (ksize == 3) {
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) { taps[nt][0] = r - 1; taps[nt][1] = s - 1; taps[nt][2] = nt; ++nt; }
    } else {
        taps[0][0] = taps[0][1] = taps[0][2] = 0; nt = 1;
    }
    return launch_taps(2, x, wt, scale, shift, slope, (int)N, (int)H, (int)W, (int)Cin, (int)Cout, Ho, Wo, 1, 0, 0, Ho, Wo,
                       nt, taps, y, workspace, workspace_floats, (hipStream_t)stream);
}

int datr_conv_s2_dgrad_nhwc_f32(const float *dy, const float *wt_t, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                int64_t Cout, int64_t ksize, float *dx, float *workspace, int64_t workspace_floats,
                                void *stream)
{
    if (!dy || !wt_t || !dx || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if ((ksize != 1 && ksize != 3) || Cout % CK || Cin % BN) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    if (N == 0) return DATR_OK;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 1) {
        // only the even pixels saw the filter: the others get zeros
        if (hipMemsetAsync(dx, 0, (size_t)N * H * W * Cin * sizeof(float), st) != hipSuccess) return DATR_ELAUNCH;
        int taps[1][3] = {{0, 0, 0}};
        return launch_taps(1, dy, wt_t, nullptr, nullptr, 1.f, (int)N, Ho, Wo, (int)Cout, (int)Cin, Ho, Wo, 2, 0, 0,
                           (int)H, (int)W, 1, taps, dx, workspace, workspace_floats, st);
    }
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            // input pixel (2 a + py, 2 b + px) was seen by output (a + dy_t, b + dx_t) through tap (r, s):
            // 2 (a + dy) + r - 1 = 2 a + py  =>  r = py + 1 - 2 dy
            int taps[9][3], nt = 0;
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) {
                    if (((py + 1 - r) & 1) || ((px + 1 - s) & 1)) continue;
                    taps[nt][0] = (py + 1 - r) / 2; taps[nt][1] = (px + 1 - s) / 2; taps[nt][2] = r * 3 + s; ++nt;
                }
            const int Hidx = (int)((H - py + 1) / 2), Widx = (int)((W - px + 1) / 2);
            const int rc = launch_taps(1, dy, wt_t, nullptr, nullptr, 1.f, (int)N, Ho, Wo, (int)Cout, (int)Cin, Hidx, Widx,
                                       2, py, px, (int)H, (int)W, nt, taps, dx, workspace, workspace_floats, st);
            if (rc != DATR_OK) return rc;
        }
    return DATR_OK;
}

int datr_conv_s2_wgrad_nhwc_f32(const float *x, const float *dy, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                int64_t Cout, int64_t ksize, float *dw, int64_t s_co, int64_t s_ci, int64_t s_r,
                                int64_t s_s, float *workspace, int64_t workspace_floats, void *stream)
{
    if (!x || !dy || !dw || !workspace || N < 0 || H < 1 || W < 1) return DATR_EINVAL;
    if ((ksize != 1 && ksize != 3) || Cin % WCI || Cout % WCO) return DATR_EUNSUPPORTED;
    if (N * H * W * std::max(Cin, Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int Ho = (int)((H + 1) / 2), Wo = (int)((W + 1) / 2);
    WgradArgs a{};
    a.x = x; a.dy = dy; a.partial = workspace;
    a.N = (int)N; a.Hin = (int)H; a.Win = (int)W; a.Cin = (int)Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = (int)Cout;
    a.tiles_x = (Wo + WTW - 1) / WTW; a.tiles_y = (Ho + WTH - 1) / WTH;
    a.slices = wgrad_slices(N, Ho, Wo, Cin, Cout);
    a.ksize = (int)ksize;
    const int NT = (int)(ksize * ksize);
    if (workspace_floats < (int64_t)a.slices * NT * Cin * Cout) return DATR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(Cin / WCI), (unsigned)(Cout / WCO), (unsigned)a.slices);
    if (N > 0) {
        if (ksize == 3) {
            constexpr int PH = (WTH - 1) * 2 + 3, PW = (WTW - 1) * 2 + 3;
            const size_t lds = (size_t)(((PH * PW * XSTR + 3) & ~3) + WTH * WTW * DSTR) * sizeof(float);
            static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(tap_wgrad<3>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
            if (!ok) return DATR_EUNSUPPORTED;
            hipLaunchKernelGGL(tap_wgrad<3>, grid, dim3(kThreads), lds, st, a);
        } else {
            constexpr int PH = (WTH - 1) * 2 + 1, PW = (WTW - 1) * 2 + 1;
            const size_t lds = (size_t)(((PH * PW * XSTR + 3) & ~3) + WTH * WTW * DSTR) * sizeof(float);
            static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(tap_wgrad<1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
            if (!ok) return DATR_EUNSUPPORTED;
            hipLaunchKernelGGL(tap_wgrad<1>, grid, dim3(kThreads), lds, st, a);
        }
    } else {
        if (hipMemsetAsync(workspace, 0, (size_t)a.slices * NT * Cin * Cout * sizeof(float), st) != hipSuccess)
            return DATR_ELAUNCH;
    }
    const long total = (long)NT * Cin * Cout;
    hipLaunchKernelGGL(tap_wgrad_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, a.slices, NT,
                       (int)Cin, (int)Cout, dw, s_co, s_ci, s_r, s_s, (int)ksize);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

}  // extern "C"
