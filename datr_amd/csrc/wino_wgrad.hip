// wino_wgrad.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolutions in the Winograd
// F(2x2, 3x3) domain on the MFMA units, exact fp32, NHWC: `FCDiscriminator_img`'s three 3x3 layers
// (/root/reference/models/dino/DA_utils.py:61-79, all pyramid levels of all 2B images,
// dino.py:351-359) and conv2 of the trainable ResNet bottlenecks (backbone.py:109-128).
//
//   forward  Y = A^T [ (G g G^T) o (B^T d B) ] A   per 2x2 output tile, summed over input channels
//   =>  dL/dg = G^T [ sum_tiles (A dY A^T) o (B^T d B) ] G
// i.e. for each of the 16 transform positions a [Cout x tiles] x [tiles x Cin] product (2.25x fewer
// multiplies than the direct weight gradient), then a 4x4 -> 3x3 fold per (cout, cin).
//
// Work split: a workgroup owns a 64 cout x 64 cin block of all 16 positions (8 waves x 2 positions x
// 2x2 accumulator blocks of v_mfma_f32_32x32x2_f32 = 128 accumulator registers) for one SEGMENT of
// tiles (split-K; the tiles of all pyramid levels, in chunks of 8, form one index space that is cut
// into equal segments, ~one workgroup per CU in flight) and writes its partial sums; `wino_wgrad_fold` adds the segments in a fixed order
// (deterministic, unlike an atomic accumulation) and applies G^T . G.
// Tiles go by in chunks of 8, double-buffered in LDS with one barrier per chunk.  Every thread has
// the same role: wave = tile of the chunk, lane = (4 channels, patch row r): it fetches row r of the
// tile's 4x4 input patch (4 buffer_load_b128) and one pixel of its 2x2 dY tile (raw buffer loads,
// out-of-image pixels read as zero), for TWO chunks ahead (two register sets; every call issues the
// same number of loads, so the compiler waits with vmcnt(5) for the older chunk only).  The four rows
// of a patch sit in the four lanes of a quad: the row transforms B^T d and A dY are one DPP quad_perm
// plus a per-lane +-1 combination, the thread then owns transform row xi = r, finishes the column
// transform in registers and writes its 4 positions x 4 channels of both operands into the layout
// [pos][row][8 tiles], in which an MFMA lane reads the four k-steps of a chunk with ONE ds_read_b128
// (k-step j pairs the j-th tile of the lower half with the j-th of the upper).  LDS rows are channels
// in the order row = 16 (c % 4) + c / 4, the two 4-tile halves of a row are swapped when bit 3 XOR bit 2
// of the row is set and the tiles of a half are rotated by xi: the b32 writes of a wave (32-lane groups,
// 32 banks) and the b128 reads (16-lane groups, 64 banks) are bank-conflict free -- by the counters
// (SQ_LDS_BANK_CONFLICT 0; with bit 3 alone the writes were two-way conflicts, 40 % of the LDS cycles,
// at no measurable cost in time).  The partial sums keep that row order; the fold kernel undoes it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "datr_hip.h"

#ifndef WGRAD_NT_PARTIALS
#define WGRAD_NT_PARTIALS 0   // non-temporal partial-sum stores: measured SLOWER (layer2 147 -> 164 us): the fold reads them next
#endif
#ifndef WGRAD_ABLATE
#define WGRAD_ABLATE 0     // development only (wrong results): 1 no fetch (16 / 32: none by the patch / dY waves), 2 no transform, 4 no multiply, 8 no loop barrier
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;                   // 8 waves, two per SIMD: 128 accumulator + <= 128 other registers each
constexpr int PW = 2;                           // positions per wave
constexpr int BC = 64;                          // channels per block (both cout and cin)
constexpr int TK = 8;                           // tiles per chunk
constexpr int kOpF = 16 * BC * TK;              // floats per operand buffer: [pos][row][8]
constexpr int kLdsBytes = 4 * kOpF * 4;         // U', V, double-buffered: 128 KiB
constexpr unsigned kOOB = 0x80000000u;

// Tiles of a level: t = (n * TH + ty) * TW + tx, padded to whole chunks; the levels' chunks are laid end
// to end (`first` = a level's first chunk) and a segment is a range of that chunk space, so a chunk
// never straddles two levels but a segment may.
struct Level { const float *x; const float *dy; int H, W, TH, TW, tiles, first; };
struct Args {
    Level lv[DATR_WINO_MAX_LEVELS];
    int nlevels, chunks, nseg, N, Cin, Cout;
};

// quad_perm DPP: lane r of every quad reads lane S_r of the same quad
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                                 S0 | (S1 << 2) | (S2 << 4) | (S3 << 6), 0xF, 0xF, true));
}
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ float4 quad_perm(float4 v) {
    return make_float4(quad_perm<S0, S1, S2, S3>(v.x), quad_perm<S0, S1, S2, S3>(v.y), quad_perm<S0, S1, S2, S3>(v.z),
                       quad_perm<S0, S1, S2, S3>(v.w));
}

__global__ __launch_bounds__(kThreads) void wino_wgrad_nhwc(Args args, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Us = smem;                           // [2][16][64][8]  A dY A^T
    float *Vs = smem + 2 * kOpF;                // [2][16][64][8]  B^T d B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int Cin = args.Cin, Cout = args.Cout;
    const int ci0 = blockIdx.x * BC, co0 = blockIdx.y * BC;
    // this workgroup's chunks [cbeg, cend): equal shares of the chunk space
    const int cbeg = (int)((long long)args.chunks * blockIdx.z / args.nseg);
    const int cend = (int)((long long)args.chunks * (blockIdx.z + 1) / args.nseg);

    // ---- fetch / transform role (every thread): tile ft = the wave, channel quad fq, patch row fr.
    // The four rows of a patch sit in the four lanes of a quad, so the row transform B^T d (and A dY)
    // is one DPP exchange inside the quad; the thread then owns transform row xi = fr.
    const int ft = wave, fq = lane >> 2, fr = lane & 3;
    const unsigned xchan = (unsigned)((ci0 + fq * 4) * 4), ychan = (unsigned)((co0 + fq * 4) * 4);
    int lvl = -1, H = 0, W = 0, TH = 0, TW = 0, ltiles = 0, lend = 0;     // current level (uniform)
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(args.lv[0].x), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t ry = rx;
    int fc = cbeg;                                 // chunk the next fetch() reads
    int tt = 0, tx = 0, ty = 0, tn = 0;            // this WAVE's tile of that chunk

    struct Raw { float4 x[4]; float4 y; };        // patch row fr (4 columns) and dY pixel (fr >> 1, fr & 1)
    auto fetch = [&](Raw &raw) {
        // Called unconditionally, also past the segment's (and the chunk space's) end -- the loads then
        // carry out-of-range offsets and return zeros: with a fixed number of loads per call the compiler
        // can wait for the OLDER chunk's five loads only (vmcnt(5)); a conditional fetch made it wait for
        // everything in flight, i.e. for the loads it had just issued.
        const bool past = fc >= args.chunks;
        if (!past && fc >= lend) {                 // (first call, or) the chunk opens the next level
            do { ++lvl; lend = (lvl + 1 < args.nlevels) ? args.lv[lvl + 1].first : args.chunks; } while (fc >= lend);
            const Level L = args.lv[lvl];
            H = L.H; W = L.W; TH = L.TH; TW = L.TW; ltiles = L.tiles;
            rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, args.N * H * W * Cin * 4, 0x00020000);
            ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.dy), 0, args.N * H * W * Cout * 4, 0x00020000);
            tt = (fc - L.first) * TK + ft;
            tx = tt % TW; ty = (tt / TW) % TH; tn = tt / (TW * TH);
        }
        const bool live = !past && tt < ltiles;
        {
            const int y = 2 * ty - 1 + fr, x0 = 2 * tx - 1;
            const bool rv = live && y >= 0 && y < H;
            const unsigned roff = (unsigned)((tn * H + y) * W + x0) * (unsigned)(Cin * 4) + xchan;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned off = (rv && x0 + b >= 0 && x0 + b < W) ? roff + (unsigned)(b * Cin * 4) : kOOB;
                raw.x[b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            }
        }
        {
            const int y = 2 * ty + (fr >> 1), x = 2 * tx + (fr & 1);
            const unsigned off = (live && y < H && x < W)
                ? (unsigned)((tn * H + y) * W + x) * (unsigned)(Cout * 4) + ychan : kOOB;
            raw.y = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
        }
        // the wave's tile of the next chunk
        ++fc; tt += TK; tx += TK;
        while (tx >= TW) { tx -= TW; if (++ty == TH) { ty = 0; ++tn; } }
    };

    // LDS float index of (pos, row, tile): halves swapped on bit 3 XOR bit 2 of the row (ds_write_b32 banks are
    // (address / 4) mod 32 over 32-lane groups: with bit 3 alone rows q and q + 4 of a group shared a bank, 40 % of the
    // kernel's LDS cycles were conflicts; the b128 reads stay conflict free in their 16-lane groups), tiles rotated inside their
    // half by the transform row xi (= fr for the writer): the four lanes of a quad then hit four banks,
    // and both operands of a position are rotated alike, so the MFMA's k pairing is unaffected
    const int fslot = (((ft >> 2) ^ ((fq >> 3) & 1) ^ ((fq >> 2) & 1)) << 2) + ((ft + fr) & 3);
    // Packed arithmetic (v_pk_add_f32 / v_pk_fma_f32: two floats per instruction) on the float4s.
    struct P4 { f32x2 lo, hi; };
    auto pk = [](float4 v) { return P4{f32x2{v.x, v.y}, f32x2{v.z, v.w}}; };
    auto sub = [](P4 a, P4 b) { return P4{a.lo - b.lo, a.hi - b.hi}; };
    auto add = [](P4 a, P4 b) { return P4{a.lo + b.lo, a.hi + b.hi}; };
    auto fma2 = [](float q, P4 b, P4 a) {                       // a + q b
        const f32x2 qq = {q, q};
        return P4{__builtin_elementwise_fma(qq, b.lo, a.lo), __builtin_elementwise_fma(qq, b.hi, a.hi)};
    };
    auto put = [&](float *buf, int nu, P4 v) {                  // pos = 4 fr + nu; channels 4 fq + e -> rows 16 e + fq
        float *p = buf + ((fr * 4 + nu) * BC + fq) * TK + fslot;
        p[0 * 16 * TK] = v.lo.x; p[1 * 16 * TK] = v.lo.y; p[2 * 16 * TK] = v.hi.x; p[3 * 16 * TK] = v.hi.y;
    };
    // Row transforms as ONE multiply-add per value, own + q * (a quad neighbour's value):
    //   B^T rows: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3.  Lane 3 (own row d3) forms d3 - d1, the
    //   NEGATIVE of row 3 -- and so does the dY side below, so the products of position row 3 are unchanged.
    //   A rows on the dY tile: xi 0: y0, 1: y0 + y1, 2: y0 - y1, 3: -y1 (stored as +y1, see above).
    const float xq = fr == 1 ? 1.f : -1.f;                      // neighbour: the row of quad lane 2, 2, 1, 1
    const float yq = fr == 1 ? 1.f : fr == 2 ? -1.f : 0.f;      // first: y0 (lane 3: y1), second: y1

    auto transform = [&](int buf, const Raw &raw) {
        {
            P4 R[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) R[b] = fma2(xq, pk(quad_perm<2, 2, 1, 1>(raw.x[b])), pk(raw.x[b]));
            float *dst = Vs + buf * kOpF;
            put(dst, 0, sub(R[0], R[2]));
            put(dst, 1, add(R[1], R[2]));
            put(dst, 2, sub(R[2], R[1]));
            put(dst, 3, sub(R[1], R[3]));
        }
        {
            // the quad holds dY[0][0], [0][1], [1][0], [1][1] in lanes 0..3
            const P4 S0 = fma2(yq, pk(quad_perm<2, 2, 2, 2>(raw.y)), pk(quad_perm<0, 0, 0, 2>(raw.y)));
            const P4 S1 = fma2(yq, pk(quad_perm<3, 3, 3, 3>(raw.y)), pk(quad_perm<1, 1, 1, 3>(raw.y)));
            float *dst = Us + buf * kOpF;
            put(dst, 0, S0);
            put(dst, 1, add(S0, S1));
            put(dst, 2, sub(S0, S1));
            put(dst, 3, P4{-S1.lo, -S1.hi});
        }
    };

    // ---- multiply role: wave w owns positions 2 w, 2 w + 1 (128 accumulator registers) ----------------
    // (not zero-initialised: the first chunk's first k-step multiplies onto the inline constant 0)
    f32x16 acc[PW][2][2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int rslot = (lhi ^ ((l31 >> 3) & 1) ^ ((l31 >> 2) & 1)) << 2;
    auto multiply = [&](int buf, auto first) {
        float4 a[PW][2], b[PW][2];
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = buf * kOpF + ((PW * wave + p) * BC + i * 32 + l31) * TK + rslot;
                a[p][i] = *reinterpret_cast<const float4 *>(Us + off);
                b[p][i] = *reinterpret_cast<const float4 *>(Vs + off);
            }
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            reinterpret_cast<const float *>(&a[p][i])[k], reinterpret_cast<const float *>(&b[p][j])[k],
                            (decltype(first)::value && k == 0) ? zero : acc[p][i][j], 0, 0, 0);
    };

    // Two chunks of raw data in flight: iteration c issues the loads of chunk c + 2 into the registers
    // chunk c freed, multiplies chunk c and then transforms chunk c + 1 (fetched one iteration earlier:
    // the loads have two multiply phases to land).  One barrier per chunk.
    const int nchunks = cend - cbeg;               // >= 1: the host never makes more segments than chunks
    Raw r0, r1;
    fetch(r0);
    transform(0, r0);
    fetch(r1);
    __syncthreads();
    auto iteration = [&](int c, Raw &mine, Raw &other, auto first) {   // `mine`: chunk c's registers (free), `other`: chunk c + 1
        if (!(WGRAD_ABLATE & 1)) fetch(mine);
        if (!(WGRAD_ABLATE & 4) || decltype(first)::value) multiply(c & 1, first);
        if (!(WGRAD_ABLATE & 2) && c + 1 < nchunks) transform((c + 1) & 1, other);
        if (!(WGRAD_ABLATE & 8)) __syncthreads();
    };
    iteration(0, r0, r1, std::true_type{});
    for (int c = 1; c < nchunks; c += 2) {
        iteration(c, r1, r0, std::false_type{});
        if (c + 1 < nchunks) iteration(c + 1, r0, r1, std::false_type{});
    }

    // ---- partial[segment][pos][cout row][cin row], rows in LDS order within each 64-block -----------
    float *out = partial + (size_t)blockIdx.z * 16 * Cout * Cin;
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
#if WGRAD_NT_PARTIALS
                    __builtin_nontemporal_store(acc[p][i][j][r], &out[((size_t)(PW * wave + p) * Cout + co0 + row) * Cin + ci0 + j * 32 + l31]);
#else
                    out[((size_t)(PW * wave + p) * Cout + co0 + row) * Cin + ci0 + j * 32 + l31] = acc[p][i][j][r];
#endif
                }
}

// dW[co][ci][r][s] = sum_{xi,nu} G[xi][r] G[nu][s] sum_segments partial[.][xi * 4 + nu][co][ci]
__global__ __launch_bounds__(256) void wino_wgrad_fold(const float *__restrict__ partial, int nseg, int Cout, int Cin,
                                                       float *__restrict__ dw, int64_t s_co, int64_t s_ci,
                                                       int64_t s_r, int64_t s_s)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    // a thread per (cout, cin) in TRUE channel order: the 64 lanes of a wave read one 64-channel block of
    // the row-ordered partials (a permutation of one contiguous 256 B line) and write contiguous weights
    const int ci = idx % Cin, co = idx / Cin;
    auto row = [](int c) { return (c & ~63) + 16 * (c & 3) + ((c & 63) >> 2); };
    const size_t at = (size_t)row(co) * Cin + row(ci);
    float M[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) M[p] = 0.f;
    for (int s = 0; s < nseg; ++s)
#pragma unroll
        for (int p = 0; p < 16; ++p) M[p] += partial[((size_t)(s * 16 + p)) * Cout * Cin + at];
    // T[r][nu] = sum_xi G[xi][r] M[xi][nu];  G columns: (1, .5, .5, 0), (0, .5, -.5, 0), (0, .5, .5, 1)
    float T[3][4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        const float h = 0.5f * (M[4 + nu] + M[8 + nu]), d = 0.5f * (M[4 + nu] - M[8 + nu]);
        T[0][nu] = M[nu] + h; T[1][nu] = d; T[2][nu] = h + M[12 + nu];
    }
    float *o = dw + co * s_co + ci * s_ci;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float h = 0.5f * (T[r][1] + T[r][2]), d = 0.5f * (T[r][1] - T[r][2]);
        o[r * s_r + 0 * s_s] = T[r][0] + h;
        o[r * s_r + 1 * s_s] = d;
        o[r * s_r + 2 * s_s] = h + T[r][3];
    }
}

// chunk space of the levels, and how many segments to cut it into: ~one workgroup per CU
void plan(const datr_wino_wgrad_level *levels, int nlevels, int64_t N, int64_t blocks, Args &a) {
    int chunks = 0;
    for (int l = 0; l < nlevels; ++l) {
        const int TH = (int)((levels[l].H + 1) / 2), TW = (int)((levels[l].W + 1) / 2);
        const int tiles = (int)(N * TH * TW);
        a.lv[l] = Level{levels[l].x, levels[l].dy, (int)levels[l].H, (int)levels[l].W, TH, TW, tiles, chunks};
        chunks += (tiles + TK - 1) / TK;
    }
    a.nlevels = nlevels;
    a.chunks = chunks;
    const int64_t want = std::max<int64_t>(1, 256 / blocks);
    a.nseg = (int)std::min<int64_t>(std::min<int64_t>(want, DATR_WINO_WGRAD_MAX_SEGMENTS), std::max(1, chunks / 4));
}

bool bad_dims(const datr_wino_wgrad_level *levels, int64_t nlevels, int64_t N, int64_t Cin, int64_t Cout) {
    if (!levels || nlevels < 1 || nlevels > DATR_WINO_MAX_LEVELS || N < 1 || Cin < 1 || Cout < 1) return true;
    for (int l = 0; l < nlevels; ++l)
        if (levels[l].H < 1 || levels[l].W < 1) return true;
    return false;
}

}  // namespace

extern "C" int64_t datr_wino_wgrad_partial_floats(const datr_wino_wgrad_level *levels, int64_t nlevels, int64_t N,
                                                  int64_t Cin, int64_t Cout) {
    if (bad_dims(levels, nlevels, N, Cin, Cout) || Cin % BC || Cout % BC) return -1;
    for (int l = 0; l < nlevels; ++l)
        if (N * levels[l].H * levels[l].W * std::max(Cin, Cout) * 4 >= (int64_t)1 << 31) return -1;
    Args a;
    plan(levels, (int)nlevels, N, (Cin / BC) * (Cout / BC), a);
    return (int64_t)a.nseg * 16 * Cin * Cout;
}

extern "C" int datr_conv3x3_wino_wgrad_nhwc_f32(const datr_wino_wgrad_level *levels, int64_t nlevels, int64_t N,
                                                int64_t Cin, int64_t Cout, float *partial, float *dw, int64_t s_co,
                                                int64_t s_ci, int64_t s_r, int64_t s_s, void *stream) {
    if (bad_dims(levels, nlevels, N, Cin, Cout) || !partial || !dw) return DATR_EINVAL;
    if (Cin % BC || Cout % BC) return DATR_EUNSUPPORTED;
    Args a;
    for (int l = 0; l < nlevels; ++l) {
        if (!levels[l].x || !levels[l].dy) return DATR_EINVAL;
        if (N * levels[l].H * levels[l].W * std::max(Cin, Cout) * 4 >= (int64_t)1 << 31) return DATR_EUNSUPPORTED;
    }
    plan(levels, (int)nlevels, N, (Cin / BC) * (Cout / BC), a);
    a.N = (int)N; a.Cin = (int)Cin; a.Cout = (int)Cout;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino_wgrad_nhwc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes) != hipSuccess)
            return DATR_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(wino_wgrad_nhwc, dim3((unsigned)(Cin / BC), (unsigned)(Cout / BC), (unsigned)a.nseg), dim3(kThreads),
                       kLdsBytes, st, a, partial);
    const int64_t pairs = Cin * Cout;
    hipLaunchKernelGGL(wino_wgrad_fold, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, partial, a.nseg, (int)Cout,
                       (int)Cin, dw, s_co, s_ci, s_r, s_s);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
