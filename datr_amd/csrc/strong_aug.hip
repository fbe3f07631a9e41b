// strong_aug.hip -- the photometric ("strong") augmentation of the target-domain images on the
// device, bit-exact with Pillow: /root/reference/datasets/DAcoco.py:330-360 builds
//   RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8), RandomGrayscale(p=0.2),
//   RandomApply([GaussianBlur([0.1, 2.0])], p=0.5)
// on PIL images; torchvision runs those through Pillow (ImageEnhance.Brightness / Contrast / Color =
// Image.blend against a black / mean-grey / luma image, an RGB->HSV->RGB round trip for the hue,
// Image.convert("L"), ImageFilter.GaussianBlur = three passes of the "extended box blur" per axis).
// Pillow is un-vendored; its arithmetic is restated (and pinned against Pillow itself) in
// oracle/pillow_ops.py, which this file follows operation for operation:
//   * blend: float32 `d + a * (x - d)`, no fused multiply-add, truncation for a in [0, 1], clipped
//     truncation outside;
//   * luma: (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//   * contrast's grey level: int(sum(luma) / pixels + 0.5) in double -- the sum is an integer
//     reduction (pixel_chain<true>), so it does not depend on the order of the atomics;
//   * HSV: float32 ratios, double for the hue fold and for the way back, round-half-away;
//   * box blur: 8.24 fixed point `(ww * sum_{|d|<=r} x[q+d] + fw * (x[q-r-1] + x[q+r+1]) + 2^23) >> 24`
//     in 32-bit unsigned arithmetic with edge clamping, per pass.
// A chain of pixel operations (the jitter's four in their drawn order, then the grayscale) is ONE
// pass over the image: 12 bytes (4 pixels) per thread, read once, written once; a contrast step in
// the chain costs one extra read-only pass for its mean.  The blur is ONE kernel: a 2-D tile with a
// halo of 3 (r + 1) pixels staged in LDS, three horizontal then three vertical passes in LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "datr_hip.h"

namespace {

// Pillow's C is compiled without fused multiply-add: every product below must round before the sum
// that follows it.  hip's __fmul_rn & co are plain operators compiled under the header's contraction
// mode, so they may still fuse; these are the operators with contraction switched off at the source.
#pragma clang fp contract(off)
#define DATR_EXACT_OP(name, T, op)                    \
    __device__ __forceinline__ T name(T a, T b) {     \
        _Pragma("clang fp contract(off)") return a op b; \
    }
DATR_EXACT_OP(fmul, float, *)
DATR_EXACT_OP(fadd, float, +)
DATR_EXACT_OP(fsub, float, -)
DATR_EXACT_OP(fdiv, float, /)
DATR_EXACT_OP(dmul, double, *)
DATR_EXACT_OP(dadd, double, +)
DATR_EXACT_OP(dsub, double, -)
DATR_EXACT_OP(ddiv, double, /)
#undef DATR_EXACT_OP

struct Chain {
    int32_t n;
    int32_t code[DATR_PIXEL_OPS_MAX];
    float alpha[DATR_PIXEL_OPS_MAX];
    int32_t shift[DATR_PIXEL_OPS_MAX];
};

__device__ __forceinline__ int luma(int r, int g, int b) {
    return (int)(((uint32_t)r * 19595u + (uint32_t)g * 38470u + (uint32_t)b * 7471u + 0x8000u) >> 16);
}

__device__ __forceinline__ int blend1(int d, int x, float a, bool inside) {
    const float t = fadd((float)d, fmul(a, (float)(x - d)));
    if (inside) return (int)t;
    return t <= 0.f ? 0 : t >= 255.f ? 255 : (int)t;
}

__device__ __forceinline__ int round_away(double x) { return (int)(x >= 0 ? floor(x + 0.5) : ceil(x - 0.5)); }
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__device__ __forceinline__ void hue_shift(int &r, int &g, int &b, int shift) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    if (maxc != minc) {
        const float cr = (float)(maxc - minc);
        const float s = fdiv(cr, (float)maxc);
        const float rc = fdiv((float)(maxc - r), cr), gc = fdiv((float)(maxc - g), cr),
                    bc = fdiv((float)(maxc - b), cr);
        float h;
        if (r == maxc) h = fsub(bc, gc);
        else if (g == maxc) h = (float)dsub(dadd(2.0, (double)rc), (double)bc);
        else h = (float)dsub(dadd(4.0, (double)gc), (double)rc);
        const double t = dadd(ddiv((double)h, 6.0), 1.0);
        h = (float)(t - floor(t));                                       // fmod(t, 1.0), t > 0
        uh = clip255((int)dmul((double)h, 255.0));
        us = clip255((int)dmul((double)s, 255.0));
    }
    uh = (uh + shift) & 0xFF;
    const int v = maxc;
    if (us == 0) { r = g = b = v; return; }
    const double hf = ddiv(dmul((double)(float)uh, 6.0), 255.0);
    const int i = (int)floor(hf);
    const double f = (double)(float)dsub(hf, (double)(float)i);
    const double fs = (double)(float)ddiv((double)(float)us, 255.0);
    const double vd = (double)v;
    const int p = clip255(round_away(dmul(vd, dsub(1.0, fs))));
    const int q = clip255(round_away(dmul(vd, dsub(1.0, dmul(fs, f)))));
    const int t = clip255(round_away(dmul(vd, dsub(1.0, dmul(fs, dsub(1.0, f))))));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// the first `n` operations of the chain on one pixel; grey[k] = contrast k's grey level
__device__ __forceinline__ void apply_chain(int &r, int &g, int &b, const Chain &ch, int n, const int *grey) {
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        const float a = ch.alpha[k];
        const bool inside = a >= 0.f && a <= 1.f;
        switch (ch.code[k]) {
            case DATR_PIXEL_BRIGHTNESS:
                r = blend1(0, r, a, inside); g = blend1(0, g, a, inside); b = blend1(0, b, a, inside); break;
            case DATR_PIXEL_CONTRAST: {
                const int m = grey[k];
                r = blend1(m, r, a, inside); g = blend1(m, g, a, inside); b = blend1(m, b, a, inside); break;
            }
            case DATR_PIXEL_SATURATION: {
                const int l = luma(r, g, b);
                r = blend1(l, r, a, inside); g = blend1(l, g, a, inside); b = blend1(l, b, a, inside); break;
            }
            case DATR_PIXEL_HUE: hue_shift(r, g, b, ch.shift[k]); break;
            default: r = g = b = luma(r, g, b); break;                    // DATR_PIXEL_GRAYSCALE
        }
    }
}

// REDUCE: apply ops [0, n) and add the luma of the result into sums[n] (the mean contrast op n needs).
// else:   apply the whole chain and write dst.  4 pixels = 3 dwords per thread, grid-stride.
template <bool REDUCE>
__global__ __launch_bounds__(256) void pixel_chain(const uint8_t *src, uint8_t *dst, int64_t npix, Chain ch, int n,
                                                   unsigned long long *sums)      // src may alias dst
{
    int grey[DATR_PIXEL_OPS_MAX];
#pragma unroll
    for (int k = 0; k < DATR_PIXEL_OPS_MAX; ++k)
        grey[k] = (k < n && ch.code[k] == DATR_PIXEL_CONTRAST) ? (int)((double)sums[k] / (double)npix + 0.5) : 0;
    const int64_t quads = npix >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    unsigned long long local = 0;
    for (int64_t qd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += stride) {
        uint32_t w[3] = {s32[3 * qd], s32[3 * qd + 1], s32[3 * qd + 2]};
        int c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) c[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            apply_chain(c[3 * p], c[3 * p + 1], c[3 * p + 2], ch, n, grey);
            if (REDUCE) local += luma(c[3 * p], c[3 * p + 1], c[3 * p + 2]);
        }
        if (!REDUCE) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                d32[3 * qd + j] = (uint32_t)c[4 * j] | ((uint32_t)c[4 * j + 1] << 8) | ((uint32_t)c[4 * j + 2] << 16) |
                                  ((uint32_t)c[4 * j + 3] << 24);
        }
    }
    // the last npix % 4 pixels
    const int64_t tail = (quads << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < npix) {
        int r = src[3 * tail], g = src[3 * tail + 1], b = src[3 * tail + 2];
        apply_chain(r, g, b, ch, n, grey);
        if (REDUCE) local += luma(r, g, b);
        else { dst[3 * tail] = (uint8_t)r; dst[3 * tail + 1] = (uint8_t)g; dst[3 * tail + 2] = (uint8_t)b; }
    }
    if (REDUCE) {                                  // one atomic per workgroup: same-address atomics serialise in L2
        __shared__ unsigned long long part[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&sums[n], part[0] + part[1] + part[2] + part[3]);
    }
}

// ---- extended box blur, `passes` horizontal then `passes` vertical, one 2-D tile per workgroup ----
// The LDS tile is kSlotsH x kSlotsW slots (two planes); its inner (kSlotsH - 2 halo) x (kSlotsW - 2 halo)
// slots are the workgroup's output pixels, halo = passes * (radius + 1).
constexpr int kSlotsW = 128, kSlotsH = 64, kBlurThreads = 1024;

__device__ __forceinline__ int med3(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(kBlurThreads) void box_blur_tile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                     int H, int W, int radius, uint32_t ww, uint32_t fw, int passes)
{
    __shared__ uint8_t lds[2][kSlotsH * kSlotsW * 3];
    const int halo = passes * (radius + 1);
    const int ow = kSlotsW - 2 * halo, oh = kSlotsH - 2 * halo;
    const int x0 = blockIdx.x * ow - halo, y0 = blockIdx.y * oh - halo;
    const int tx = threadIdx.x & (kSlotsW - 1), ty = threadIdx.x >> 7;       // kBlurThreads / 128 rows in flight
    uint8_t *cur = lds[0], *nxt = lds[1];
    {
        const int sx = med3(x0 + tx, 0, W - 1);
        for (int i = ty; i < kSlotsH; i += kBlurThreads / kSlotsW) {
            const uint8_t *p = src + ((size_t)med3(y0 + i, 0, H - 1) * W + sx) * 3;
            uint8_t *o = cur + (i * kSlotsW + tx) * 3;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        }
    }
    __syncthreads();
    // a slot stands for the image position clamp(origin + slot): slots beyond the image edge repeat the
    // edge value of the CURRENT pass, which is what the next pass's clamped reads expect.  Slots whose
    // taps fall outside the tile hold garbage that never reaches the inner slots (halo = passes taps).
    for (int pass = 0; pass < 2 * passes; ++pass) {
        const bool horiz = pass < passes;
        const int len = horiz ? kSlotsW : kSlotsH, origin = horiz ? x0 : y0, extent = horiz ? W : H;
        const int step = horiz ? 3 : kSlotsW * 3;
        const int smin = med3(-origin, 0, len - 1), smax = med3(extent - 1 - origin, 0, len - 1);
        for (int i = ty; i < kSlotsH; i += kBlurThreads / kSlotsW) {
            const int pos = horiz ? tx : i;
            const int q = med3(pos, smin, smax);                       // slot of clamp(origin + pos)
            const uint8_t *base = cur + (horiz ? i * kSlotsW * 3 : tx * 3);
            uint32_t a0 = 0, a1 = 0, a2 = 0;
            for (int d = -radius; d <= radius; ++d) {
                const uint8_t *p = base + med3(q + d, smin, smax) * step;
                a0 += p[0]; a1 += p[1]; a2 += p[2];
            }
            const uint8_t *pl = base + med3(q - radius - 1, smin, smax) * step;
            const uint8_t *pr = base + med3(q + radius + 1, smin, smax) * step;
            uint8_t *o = nxt + (i * kSlotsW + tx) * 3;
            o[0] = (uint8_t)((a0 * ww + (uint32_t)(pl[0] + pr[0]) * fw + (1u << 23)) >> 24);
            o[1] = (uint8_t)((a1 * ww + (uint32_t)(pl[1] + pr[1]) * fw + (1u << 23)) >> 24);
            o[2] = (uint8_t)((a2 * ww + (uint32_t)(pl[2] + pr[2]) * fw + (1u << 23)) >> 24);
        }
        __syncthreads();
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
    const int x = x0 + tx;
    if (tx >= halo && tx < halo + ow && x < W) {
        for (int i = halo + ty; i < halo + oh; i += kBlurThreads / kSlotsW) {
            const int y = y0 + i;
            if (y >= H) break;
            const uint8_t *p = cur + (i * kSlotsW + tx) * 3;
            uint8_t *o = dst + ((size_t)y * W + x) * 3;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        }
    }
}

}  // namespace

extern "C" int datr_pixel_ops_u8(const uint8_t *src, uint8_t *dst, int64_t npix, const datr_pixel_op *ops,
                                 int64_t nops, uint64_t *sums, void *stream) {
    if (!src || !dst || npix <= 0 || nops < 0 || (nops > 0 && !ops)) return DATR_EINVAL;
    if (nops > DATR_PIXEL_OPS_MAX) return DATR_EUNSUPPORTED;
    if (((uintptr_t)src | (uintptr_t)dst) & 3) return DATR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Chain ch{};
    ch.n = (int)nops;
    bool has_contrast = false;
    for (int k = 0; k < nops; ++k) {
        if (ops[k].code < DATR_PIXEL_BRIGHTNESS || ops[k].code > DATR_PIXEL_GRAYSCALE) return DATR_EINVAL;
        ch.code[k] = ops[k].code; ch.alpha[k] = ops[k].alpha; ch.shift[k] = ops[k].shift & 0xFF;
        has_contrast |= ops[k].code == DATR_PIXEL_CONTRAST;
    }
    if (has_contrast && !sums) return DATR_EINVAL;
    const int64_t quads = npix >> 2;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>((quads + 255) / 256, 1), 256 * 8);
    if (has_contrast) {
        if (hipMemsetAsync(sums, 0, DATR_PIXEL_OPS_MAX * sizeof(uint64_t), st) != hipSuccess) return DATR_ELAUNCH;
        for (int k = 0; k < nops; ++k)
            if (ch.code[k] == DATR_PIXEL_CONTRAST)
                hipLaunchKernelGGL(pixel_chain<true>, dim3(std::min(blocks, 512)), dim3(256), 0, st, src, dst, npix, ch, k,
                                   reinterpret_cast<unsigned long long *>(sums));
    }
    hipLaunchKernelGGL(pixel_chain<false>, dim3(blocks), dim3(256), 0, st, src, dst, npix, ch, (int)nops,
                       reinterpret_cast<unsigned long long *>(sums));
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}

extern "C" int datr_box_blur_u8(const uint8_t *src, uint8_t *dst, int64_t H, int64_t W, int64_t radius,
                                uint32_t ww, uint32_t fw, int64_t passes, void *stream) {
    if (!src || !dst || src == dst || H <= 0 || W <= 0 || radius < 0 || passes <= 0) return DATR_EINVAL;
    if (H > (1 << 24) || W > (1 << 24)) return DATR_EUNSUPPORTED;
    const int64_t halo = passes * (radius + 1);
    if (2 * halo > kSlotsH - 8) return DATR_EUNSUPPORTED;
    const int64_t ow = kSlotsW - 2 * halo, oh = kSlotsH - 2 * halo;
    dim3 grid((unsigned)((W + ow - 1) / ow), (unsigned)((H + oh - 1) / oh));
    hipLaunchKernelGGL(box_blur_tile, grid, dim3(kBlurThreads), 0, static_cast<hipStream_t>(stream), src, dst, (int)H, (int)W,
                       (int)radius, ww, fw, (int)passes);
    return hipGetLastError() == hipSuccess ? DATR_OK : DATR_ELAUNCH;
}
