// wino.hip -- Winograd F(2x2, 3x3) convolution on the MFMA units, exact fp32, NHWC: the 3x3 /
// stride 1 / pad 1 convolutions of DATR's image-level domain discriminator (`FCDiscriminator_img`,
// /root/reference/models/dino/DA_utils.py:61-79, applied to the four pyramid levels of all 2B
// images behind the gradient-reversal layer, /root/reference/models/dino/dino.py:351-359) and the
// conv2 of the ResNet bottlenecks (/root/reference/models/dino/backbone.py:62-72, 109-128) -- the
// forward with scale / shift (bias or frozen BN) + (Leaky)ReLU fused, and the data gradient (the same
// kernel on dY with the transposed / flipped filter) with the activation gate of the previous layer
// and the GRL sign fused.
//
//   Y = A^T [ sum_ci (G g G^T) o (B^T d B) ] A        (Lavin & Gray's minimal filtering, 2.25x fewer
//   multiplies than the direct form; MIOpen's fp32 kernel for these layers is the same algorithm on
//   the VALU, this one runs the 16 per-position channel contractions as v_mfma_f32_32x32x2_f32).
//
// Work split (round 4; the round-2/3 kernel used 512-thread workgroups of 8 x 8 tiles, one per CU, with
// the filter slabs staged in 64 KB of LDS -- profiles/HISTORY.md): a workgroup = 256 threads = 4 waves
// owns 4 x 8 tiles (8 x 16 output pixels) x 64 output channels, TWO workgroups per CU.  Wave (wp, wb)
// holds transform rows xi = 2 wp, 2 wp + 1 (8 of the 16 positions = 8 accumulator blocks = 128
// registers) of all 32 tiles for channels 32 wb .. + 31, so a SIMD runs two waves that belong to
// DIFFERENT workgroups: one workgroup's prologue (dependent patch loads, first transform), barrier
// waits and inverse transform + stores pass behind the other's MFMAs; blocks are small (finer tail
// rounds on the 50 x 84 / 25 x 42 maps) and the 4 x 8 tile block wastes less on map edges (25 x 42
// tiles: 78 % of the computed tiles are real against 68 % for 8 x 8).  Input channels go by in chunks
// of FOUR, software-pipelined with one barrier per chunk: while chunk c is multiplied, the raw 10 x 18
// patch of chunk c + 1 (staged in LDS, zero outside the image) is turned into V = B^T d B -- a thread
// owns (tile, one transform row, two columns) --, the raw patch of chunk c + 2 is written to LDS and
// that of chunk c + 3 is in flight in a register.  The FILTER operand goes from global memory (L2)
// straight into the MFMA registers: wave (wp, wb) is the only reader of its positions x channels, an
// LDS stage would share nothing.  The transformed filter is stored so that a lane finds what it
// multiplies in TWO consecutive chunks in one 16-byte load (U[pos][Cin/8][Cout][8], wino_weights);
// there is one register set, refilled position by position as the odd chunk of a pair finishes with
// them.  A k-step multiplies channels (m, 2 + m) of the
// chunk: lane half `lhi` reads channels 2 lhi, 2 lhi + 1 of its tile with one ds_read_b64; V is laid out
// [pos][channel half][tile][2], so each 32-lane pass of the read covers 256 contiguous bytes (round 5; with
// [pos][tile][4] the lanes of a pass sat 16 B apart: two-way bank conflicts, measured).  The inverse transform A^T M A is lane-local up to one
// exchange of 32 floats per lane between the two position halves (through LDS, once per workgroup),
// and a lane's 32-lane row stores 128 contiguous bytes of NHWC output (non-temporal: -3 %).  All
// levels of the pyramid share the filter, so they are ONE launch (level table in the arguments).
//
// Measured (profiles/r04_wino_s32.txt; 4 images 1333 x 800, us per launch, old -> new): 64 ch @
// 200 x 334: 152 -> 129; 128 ch @ 100 x 167: 152 -> 129; 256 ch @ 50 x 84: 179 -> 139; 512 ch @
// 25 x 42: 170 -> 164; 256 -> 256 @ 100 x 167 (discriminator): 441 -> 392 = 201 TF/s direct-
// equivalent; training step -1.0 ms (before the 16-byte filter loads, which took another 4-5 % off
// every layer).  With pieces compiled out (same file): the MFMA stream with
// prologue / epilogue alone is 72 % of the time, filter loads 10 %, transform 5 %, barrier 5 %,
// patch copies 4 %, operand reads < 1 %.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "datr_hip.h"

// where in a chunk's four MFMA slots the next chunk's transform reads its patch rows, writes V, and where the
// chunk's barrier sits (measured on the four backbone layers, us: (0, 2, 3) 139 / 132 / 144 / 170;
// (0, 1, 1) 140 / 143 / 173 / 167; (1, 3, 3) 141 / 142 / 150 / 174; (1, 2, 2) 135 / 141 / 163 / 162)
#ifndef WINO_TR_LOAD_SLOT
#define WINO_TR_LOAD_SLOT 0
#define WINO_TR_STORE_SLOT 2
#define WINO_BARRIER_SLOT 3
#endif
#ifndef WINO_ABLATE
#define WINO_ABLATE 0      // development only (wrong results): 1 no filter loads, 2 no transform, 4 no operand reads, 8 no barrier, 16 no patch copies
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoLevel { const float *x; float *y; const float *gate; int H, W, tbx, tby, first; };
struct WinoArgs { WinoLevel lv[DATR_WINO_MAX_LEVELS]; int nlevels, Cin, Cout; float slope, gate_slope, out_scale; };

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

constexpr int BN = 64;                         // output channels per workgroup
constexpr int CK8 = 8;                         // channel granularity of the filter layout U (wino_weights)
constexpr int kThreads = 256;
constexpr int TBY = 4, TBX = 8;                // tiles per workgroup
constexpr int PH = 2 * TBY + 2, PW = 2 * TBX + 2;   // 10 x 18 patch
constexpr int PPIX = PH * PW;                  // 180
constexpr int CK = 4;
// raw patch in LDS: [row][column parity][10 pixel slots][4 channels] -- a row is 320 B.  A transform thread reads
// pixels 2 t_tx + {0..3} of rows 2 t_ty + {0..3}: with even and odd columns in separate planes the eight tiles of a
// tile row read 128 contiguous bytes, and two patch rows down (the next tile row, the other half of a 16-lane LDS
// phase) is 640 B = 128 B further in the bank map: conflict free.  ([pixel][4] with 18 pixels per row put every second
// tile of the two tile rows of a phase on the same banks: 57 % of the kernel's LDS cycles were bank conflicts.)
constexpr int kPatchRowF = 80;
constexpr int kPatchF = PH * kPatchRowF;       // floats
__device__ __forceinline__ int patch_off(int py, int px) { return py * kPatchRowF + (px & 1) * 40 + (px >> 1) * 4; }
constexpr int kVPos = 32 * 4;                  // floats between positions of V: [pos][channel half][tile][2] -- the 32 lanes of a
                                               // ds_read_b64 pass read 256 contiguous bytes ([pos][tile][4] put them 16 B apart: two-way conflicts)
constexpr int kVF = 16 * kVPos;
constexpr int kXchF = 4 * 32 * 64;             // the inverse transform's exchange buffer (re-uses everything)
constexpr int kLdsBytes = (2 * kPatchF + 3 * kVF > kXchF ? 2 * kPatchF + 3 * kVF : kXchF) * 4;   // 32 KB

__global__ __launch_bounds__(kThreads, 2) void wino_conv_nhwc(WinoArgs args, const float *__restrict__ U,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                       // [2][180][4]
    float *Vs = patch + 2 * kPatchF;           // [3][16][32][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wp = wave >> 1, wb = wave & 1;

    // (blockIdx.x = spatial block, blockIdx.y = output-channel tile.  Giving each XCD its own channel tiles --
    // so that a tile's filter slabs stay in one L2 -- was measured and is no faster: 149 -> 153 us on 256 ch.)
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < DATR_WINO_MAX_LEVELS; ++i)
        if (i < args.nlevels && (int)blockIdx.x >= args.lv[i].first) lvl = i;
    const WinoLevel L = args.lv[lvl];
    int r = blockIdx.x - L.first;
    const int tbx = r % L.tbx; r /= L.tbx;
    const int tby = r % L.tby;
    const int n = r / L.tby;
    const int H = L.H, W = L.W, Cin = args.Cin, Cout = args.Cout;
    const int y0 = tby * 2 * TBY, x0 = tbx * 2 * TBX;
    const int co0 = blockIdx.y * BN;
    const float *Xn = L.x + (size_t)n * H * W * Cin;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    const int nchunks = Cin / CK;
    // The filter operand goes from global memory (L2) STRAIGHT into the MFMA registers: wave (wp, wb) is the
    // only reader of positions 8 wp .. + 7 x channels 32 wb .. + 31, so an LDS stage would share nothing.
    // U[pos][Cin/8][Cout][8]: the 8 floats of (pos, 8-channel block, cout) are ordered [lhi][chunk half][m] --
    // lane half lhi finds the channels it multiplies in BOTH 4-channel chunks of the block (2 lhi + m of each) in
    // one 16-byte load: 8 loads per TWO chunks, issued a pair ahead.
    const __amdgpu_buffer_rsrc_t usrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(U), 0, 16 * Cin * Cout * 4, 0x00020000);
    const unsigned b_voff = (unsigned)((co0 + wb * 32 + l31) * 32 + lhi * 16);
    const unsigned b_pos_stride = (unsigned)(Cin / 8) * (unsigned)Cout * 32u;      // bytes between positions
    // ONE register set: position p of the next pair is requested right after the odd chunk's last MFMA on
    // position p (the load lands at least three MFMA slots before its first use)
    float4 bq[8];
    auto load_b = [&](int P, int p) {
        const unsigned soff = (unsigned)(wp * 8 + p) * b_pos_stride + (unsigned)P * (unsigned)Cout * 32u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(usrc, b_voff, soff, 0));
    };
    // raw patch of chunk c: 180 pixels x one float4, zero outside the image (out-of-range buffer offset)
    float4 pf;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Xn), 0, H * W * Cin * 4, 0x00020000);
    unsigned poff;
    {
        const int py = tid / PW, px = tid - py * PW;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool in = tid < PPIX && yy >= 0 && yy < H && xx >= 0 && xx < W;
        poff = in ? (unsigned)(((yy * W + xx) * Cin) * 4) : 0x80000000u;
    }
    auto fetch_patch = [&](int c) {
        const auto r_ = __builtin_amdgcn_raw_buffer_load_b128(xsrc, poff, c * CK * 4, 0);
        pf = __builtin_bit_cast(float4, r_);
    };
    const int pdst = patch_off(tid / PW, tid % PW);        // this thread's pixel of the raw patch
    auto store_patch = [&](float *dst) {
        if (tid < PPIX) *reinterpret_cast<float4 *>(&dst[pdst]) = pf;
    };

    // transform unit of this thread: (tile, ONE transform row xi = 2 rp + q, the column pair 2 cp, 2 cp + 1);
    // rp, cp are wave-uniform, q = lane half.  Row xi of (+-) B^T d is  alpha d[B] + d[R]  over the patch
    // rows (A, B, C) = (0, 2, 1) / (2, 1, 3) of the row pair: q = 0: A - B, q = 1: C + s_r B -- no branches.
    // (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]; the LAST row / column are computed negated, the filter
    // transform carries the same sign: wino_weights.)
    const int t_tile = lane & 31, t_q = lane >> 5;
    const int t_rp = wave & 1, t_cp = wave >> 1;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const int p_base = 2 * t_ty * kPatchRowF + t_tx * 4;
    const int rowA = t_rp ? 2 : 0, rowB = t_rp ? 1 : 2, rowC = t_rp ? 3 : 1;
    const int r_off[2] = {rowB * kPatchRowF, (t_q ? rowC : rowA) * kPatchRowF};
    // patch column 2 t_tx + cx sits in plane cx & 1 at slot t_tx + (cx >> 1)
    auto col = [](int cx) { return (cx & 1) * 40 + (cx >> 1) * 4; };
    const int c_off[3] = {col(t_cp ? 2 : 0), col(t_cp ? 1 : 2), col(t_cp ? 3 : 1)};
    const float alpha = t_q ? (t_rp ? -1.f : 1.f) : -1.f, s_c = t_cp ? -1.f : 1.f;
    const int v_off = ((2 * t_rp + t_q) * 4 + 2 * t_cp) * kVPos + t_tile * 2;

    float4 d[2][3];
    auto f4fma = [](float s_, float4 b, float4 c) {
        return make_float4(fmaf(s_, b.x, c.x), fmaf(s_, b.y, c.y), fmaf(s_, b.z, c.z), fmaf(s_, b.w, c.w));
    };
    auto tr_load = [&](const float *pb) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                d[rr][cc] = *reinterpret_cast<const float4 *>(pb + p_base + r_off[rr] + c_off[cc]);
    };
    auto tr_store = [&](float *vbuf) {
        float4 tq[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) tq[cc] = f4fma(alpha, d[0][cc], d[1][cc]);
        float *vb = vbuf + v_off;
        const float4 v0 = f4sub(tq[0], tq[1]), v1 = f4fma(s_c, tq[1], tq[2]);
        *reinterpret_cast<float2 *>(vb) = make_float2(v0.x, v0.y);                  // channels 0, 1: lane half 0's k-steps
        *reinterpret_cast<float2 *>(vb + 64) = make_float2(v0.z, v0.w);             // channels 2, 3: lane half 1's
        *reinterpret_cast<float2 *>(vb + kVPos) = make_float2(v1.x, v1.y);
        *reinterpret_cast<float2 *>(vb + kVPos + 64) = make_float2(v1.z, v1.w);
    };

    // ---- prologue ------------------------------------------------------------------------------------
    {
        // the first three raw patches in flight together (one memory latency, not three)
        const auto r0 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, poff, 0, 0);
        const auto r1 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, poff, CK * 4, 0);
        if (nchunks > 2) fetch_patch(2);
#pragma unroll
        for (int p = 0; p < 8; ++p) bq[p] = load_b(0, p);
        if (tid < PPIX) {
            *reinterpret_cast<float4 *>(&patch[pdst]) = __builtin_bit_cast(float4, r0);
            *reinterpret_cast<float4 *>(&patch[kPatchF + pdst]) = __builtin_bit_cast(float4, r1);
        }
    }
    __syncthreads();
    tr_load(patch);
    tr_store(Vs);
    __syncthreads();

    // Pipeline: chunk c multiplies V[c % 3]; one of its MFMA slots reads the raw patch of chunk c + 1, a later
    // one writes its transform to V[(c + 1) % 3], and ONE barrier per chunk (LDS writes only: no wait for the
    // loads in flight) publishes it.  V has three buffers so that the barrier may sit anywhere in the chunk
    // (WINO_BARRIER_SLOT; with it before the last slot the first operands of chunk c + 1 are fetched inside
    // chunk c -- measured slower than the barrier at the end, see the table at the top).
    const int a_off = wp * 8 * kVPos + lhi * 64 + l31 * 2;
    float2 a0 = *reinterpret_cast<const float2 *>(Vs + a_off);
    float2 a1 = *reinterpret_cast<const float2 *>(Vs + a_off + kVPos);
    int vi = 0;                                // c % 3
    auto chunk = [&](int c, auto h_tag, auto more_tag) {
        constexpr int buf = decltype(h_tag)::value;        // = c & 1: raw-patch buffer; which half of the filter pair
        constexpr bool more = decltype(more_tag)::value;
        // invariant: V[c % 3] holds chunk c, bq the filter of chunks (c & ~1, c | 1) -- positions already multiplied
        // in the odd chunk hold the next pair's --; patch[buf ^ 1] holds the raw chunk c + 1; the register holds
        // the raw chunk c + 2
        if (!(WINO_ABLATE & 16)) {
        if (c + 2 < nchunks) store_patch(patch + buf * kPatchF);
        if (c + 3 < nchunks) fetch_patch(c + 3);
        }

        const int vn = vi == 2 ? 0 : vi + 1;
        const float *va = Vs + vi * kVF + a_off;
        const float *van = Vs + vn * kVF + a_off;
        const float *pnext = patch + (buf ^ 1) * kPatchF;
        float *vnext = Vs + vn * kVF;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int p0 = 2 * sl, p1 = 2 * sl + 1;
            const float *nxt = sl < 3 ? va + (2 * sl + 2) * kVPos : van;
            const float b0x = buf ? bq[p0].z : bq[p0].x, b0y = buf ? bq[p0].w : bq[p0].y;
            const float b1x = buf ? bq[p1].z : bq[p1].x, b1y = buf ? bq[p1].w : bq[p1].y;
            float2 an0 = a0, an1 = a1;
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0x, acc[p0], 0, 0, 0);
            if (!(WINO_ABLATE & 4) && (sl < 3 || (more && WINO_BARRIER_SLOT < 3))) an0 = *reinterpret_cast<const float2 *>(nxt);
            __builtin_amdgcn_sched_barrier(0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1x, acc[p1], 0, 0, 0);
            if (!(WINO_ABLATE & 4) && (sl < 3 || (more && WINO_BARRIER_SLOT < 3))) an1 = *reinterpret_cast<const float2 *>(nxt + kVPos);
            __builtin_amdgcn_sched_barrier(0);
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0y, acc[p0], 0, 0, 0);
            if (more && sl == WINO_TR_LOAD_SLOT && !(WINO_ABLATE & 2)) tr_load(pnext);
            if (more && sl == WINO_TR_STORE_SLOT && !(WINO_ABLATE & 2)) tr_store(vnext);
            __builtin_amdgcn_sched_barrier(0);
            acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1y, acc[p1], 0, 0, 0);
            if (buf == 1 && more && !(WINO_ABLATE & 1)) {  // the next pair's filter for the two positions just finished
                bq[p0] = load_b((c >> 1) + 1, p0);
                bq[p1] = load_b((c >> 1) + 1, p1);
            }
            // LDS writes only: __syncthreads() would also wait for the filter / patch loads just issued
            if (more && sl == WINO_BARRIER_SLOT && !(WINO_ABLATE & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (more && sl == 3 && WINO_BARRIER_SLOT == 3 && !(WINO_ABLATE & 4)) {
                an0 = *reinterpret_cast<const float2 *>(nxt);
                an1 = *reinterpret_cast<const float2 *>(nxt + kVPos);
            }
            a0 = an0; a1 = an1;
        }
        vi = vn;
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    for (int c = 0; c + 2 < nchunks; c += 2) {           // nchunks is even (Cin % 8 == 0)
        chunk(c, H0{}, std::true_type{});
        chunk(c + 1, H1{}, std::true_type{});
    }
    chunk(nchunks - 2, H0{}, std::true_type{});
    chunk(nchunks - 1, H1{}, std::false_type{});
    __syncthreads();                           // everyone is done with V: the exchange buffer re-uses it

    // ---- inverse transform: lane = cout, register e = tile.  This wave holds transform rows
    // xi = 2 wp, 2 wp + 1: T[xi][j] = (M A)[xi][j], and its share of Y = A^T T is
    //   wp 0: Y[0][j] += T[0][j] + T[1][j], Y[1][j] += T[1][j];   wp 1: Y[0][j] += T[2][j], Y[1][j] += -T[2][j] - T[3][j].
    // Wave wp finishes output row i = wp of every tile: the partial of the OTHER row goes to the
    // partner wave (wave ^ 2: same channels) through LDS (32 floats per lane).
    float mine[2][16], other[2][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float t[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float m0 = acc[q * 4][e], m1 = acc[q * 4 + 1][e], m2 = acc[q * 4 + 2][e], m3 = acc[q * 4 + 3][e];
            t[q][0] = m0 + m1 + m2;
            t[q][1] = m1 - m2 - m3;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (wp == 0) { mine[j][e] = t[0][j] + t[1][j]; other[j][e] = t[1][j]; }
            else         { mine[j][e] = -t[0][j] - t[1][j]; other[j][e] = t[0][j]; }
        }
    }
    float *xch = smem;                         // [4 waves][32][64]: every wave is past the last chunk's barrier
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[(wave * 32 + j * 16 + e) * 64 + lane] = other[j][e];
    __syncthreads();
    const float *got = xch + ((wave ^ 2) * 32) * 64 + lane;

    const int co = co0 + wb * 32 + l31;
    const float sc = scale ? scale[co] : 1.f;
    const float sh = shift ? shift[co] : 0.f;
    const float slope = args.slope, gslope = args.gate_slope, oscale = args.out_scale;
    // Output rows and gate values go through buffer descriptors with 32-bit offsets: a pixel outside the image gets
    // an out-of-range offset (its store is dropped, its gate reads zero) -- no branch per element, no 64-bit address
    // arithmetic.  All 32 gate values of the lane are requested TOGETHER and before the first store, and scale / shift
    // are settled before the loop: written as one loop over (load gate, blend, store) with bounds branches, the compiler
    // waited for EVERYTHING in flight -- the previous store included -- ahead of every element (vmcnt(0) 32 times:
    // 163-178 us for the data-gradient launches of layer2 / layer3 against 112-125 us forward).
    auto uniform_ptr = [](const float *p) {         // wave-uniform by construction (blockIdx only): say so, or the
        const unsigned long long v = (unsigned long long)p;        // descriptor lands in VGPRs and every access in a waterfall loop
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (float *)(((unsigned long long)hi << 32) | lo);
    };
    const bool gated = __builtin_amdgcn_readfirstlane(L.gate != nullptr ? 1 : 0) != 0;
    const int img_bytes = __builtin_amdgcn_readfirstlane(H * W * Cout * 4);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(L.y + (size_t)n * H * W * Cout), 0, img_bytes, 0x00020000);
    // ungated: a descriptor of zero bytes -- every load is out of range and returns 0 without touching memory
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(gated ? L.gate + (size_t)n * H * W * Cout : L.y), 0, gated ? img_bytes : 0, 0x00020000);
    unsigned ooff[2][16];
    float gv[2][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tile = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const int yy = y0 + 2 * (tile >> 3) + wp, ox = x0 + 2 * (tile & 7);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int xx = ox + j;
            ooff[j][e] = (yy < H && xx < W) ? (unsigned)(((yy * W + xx) * Cout + co) * 4) : 0x80000000u;
            gv[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gsrc, ooff[j][e], 0, 2));
        }
    }
    asm volatile("" : : "v"(sc), "v"(sh));        // scale / shift have landed before the first store is issued
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v = (mine[j][e] + got[(j * 16 + e) * 64]) * sc + sh;
            v = v > 0.f ? v : v * slope;
            v = (!gated || gv[j][e] > 0.f) ? v : v * gslope;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v * oscale), ysrc, ooff[j][e], 0, 2);
        }
}

// U[pos][Cin/8][Cout][8] = G g G^T of every (ci, co) filter, the 8 channels of a block in the order the MFMA
// lanes consume them ([lane half][4-channel chunk of the block][k-step]: channel j = 4 h + 2 lhi + m sits at
// 4 lhi + 2 h + m); flip = the data-gradient filter (taps mirrored; the caller swaps the channel strides)
// Uf (may be null) = the data-gradient filter of the SAME weight from the same launch: mirroring the taps permutes
// the transform rows / columns by (3, 1, 2, 0) -- G flip(g) G^T = P (G g G^T) P -- so every product computed here
// is also an element of the flipped transform, with the channel roles swapped.
__global__ void wino_weights(const float *__restrict__ w, long s_co, long s_ci, long s_r, long s_s, int flip,
                             int Cin, int Cout, float *__restrict__ U, float *__restrict__ Uf)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int co = idx % Cout, ci = idx / Cout;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int rr = flip ? 2 - r : r, qq = flip ? 2 - q : q;
            g[r][q] = w[co * s_co + ci * s_ci + rr * s_r + qq * s_s];
        }
    float gg[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        gg[0][q] = g[0][q];
        gg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
        gg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
        gg[3][q] = g[2][q];
    }
    const int nchunks = Cin / CK8;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const float u[4] = {gg[xi][0], 0.5f * (gg[xi][0] + gg[xi][1] + gg[xi][2]),
                            0.5f * (gg[xi][0] - gg[xi][1] + gg[xi][2]), gg[xi][2]};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int pos = xi * 4 + nu;
            // row 3 / column 3 of the input transform are computed negated (wino_conv_nhwc): same sign here
            const float sg = ((xi == 3) != (nu == 3)) ? -1.f : 1.f;
            const int j = ci % CK8;
            U[(((size_t)pos * nchunks + ci / CK8) * Cout + co) * 8 + ((j & 3) >> 1) * 4 + (j >> 2) * 2 + (j & 1)] = sg * u[nu];
            if (Uf) {                              // input channels of the data gradient = co, output channels = ci
                const int xf = xi == 0 ? 3 : xi == 3 ? 0 : xi, nf = nu == 0 ? 3 : nu == 3 ? 0 : nu;
                const float sf = ((xf == 3) != (nf == 3)) ? -1.f : 1.f;
                const int jf = co % CK8;
                Uf[(((size_t)(xf * 4 + nf) * (Cout / CK8) + co / CK8) * Cin + ci) * 8 + ((jf & 3) >> 1) * 4 + (jf >> 2) * 2 + (jf & 1)] = sf * u[nu];
            }
        }
    }
}

}  // namespace

extern "C" int datr_wino_weights_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci,
                                     int64_t s_r, int64_t s_s, int flip, float *u, void *stream) {
    if (!w || !u || Cin <= 0 || Cout <= 0) return DATR_EINVAL;
    if (Cin % CK8 != 0 || Cout % BN != 0 || Cin * Cout > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int total = (int)(Cin * Cout);
    hipLaunchKernelGGL(wino_weights, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (long)s_co,
                       (long)s_ci, (long)s_r, (long)s_s, flip, (int)Cin, (int)Cout, u, (float *)nullptr);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_wino_weights_pair_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci,
                                          int64_t s_r, int64_t s_s, float *u, float *u_flip, void *stream) {
    if (!w || !u || !u_flip || Cin <= 0 || Cout <= 0) return DATR_EINVAL;
    if (Cin % BN != 0 || Cout % BN != 0 || Cin * Cout > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    const int total = (int)(Cin * Cout);
    hipLaunchKernelGGL(wino_weights, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (long)s_co,
                       (long)s_ci, (long)s_r, (long)s_s, 0, (int)Cin, (int)Cout, u, u_flip);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_conv3x3_wino_nhwc_f32(const datr_wino_level *levels, int64_t nlevels, int64_t N, int64_t Cin,
                                          int64_t Cout, const float *u, const float *scale, const float *shift,
                                          float slope, float gate_slope, float out_scale, void *stream) {
    if (!levels || !u || nlevels <= 0 || N <= 0 || Cin <= 0 || Cout <= 0) return DATR_EINVAL;
    if (nlevels > DATR_WINO_MAX_LEVELS || Cin % CK8 != 0 || Cout % BN != 0) return DATR_EUNSUPPORTED;
    WinoArgs a;
    a.nlevels = (int)nlevels; a.Cin = (int)Cin; a.Cout = (int)Cout;
    a.slope = slope; a.gate_slope = gate_slope; a.out_scale = out_scale;
    long blocks = 0;
    for (int i = 0; i < DATR_WINO_MAX_LEVELS; ++i) {
        const datr_wino_level &s = levels[i < nlevels ? i : 0];
        if (i < nlevels) {
            if (!s.x || !s.y || s.H <= 0 || s.W <= 0) return DATR_EINVAL;
            if (N * s.H * s.W * (Cin > Cout ? Cin : Cout) > 0x7fffffffLL) return DATR_EUNSUPPORTED;
            if (s.H * s.W * (Cin > Cout ? Cin : Cout) * 4 > 0x7fffffffLL) return DATR_EUNSUPPORTED;   // per-image buffer descriptors
        }
        WinoLevel &d = a.lv[i];
        d.x = s.x; d.y = s.y; d.gate = s.gate; d.H = (int)s.H; d.W = (int)s.W;
        d.tbx = (int)((s.W + 2 * TBX - 1) / (2 * TBX)); d.tby = (int)((s.H + 2 * TBY - 1) / (2 * TBY));
        d.first = (int)blocks;
        if (i < nlevels) blocks += (long)N * d.tbx * d.tby;
    }
    if (blocks > 0x7fffffffLL || 16 * Cin * Cout * 4 > 0x7fffffffLL) return DATR_EUNSUPPORTED;
    dim3 grid((unsigned)blocks, (unsigned)(Cout / BN));
    hipLaunchKernelGGL(wino_conv_nhwc, grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, a, u, scale, shift);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
