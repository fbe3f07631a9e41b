// VALU / LDS issue-rate probe for the MSDA gather inner loop (round 3).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/valu_lds_probe tools/probes/valu_lds_probe.hip
// Questions:
//   1. v_fma_f32 vs v_pk_fma_f32: lane-FMAs per clock per SIMD with 1..4 waves per SIMD.
//   2. ds_read_b128 gathers (random 128-B rows, the quad mapping of msda_fwd_pyr.hip) alone, and
//      interleaved with the 8 (or 4 packed) FMAs each pair of reads feeds: do LDS and VALU overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0 = v_fma_f32, 1 = v_pk_fma_f32
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float w0) {
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f4 v0 = {1.f + threadIdx.x, 2.f, 3.f, 4.f}, v1 = v0 * 1.5f, v2 = v0 * 2.5f, v3 = v0 * 3.5f;
    float w = w0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                a0.x = fmaf(w, v0.x, a0.x); a0.y = fmaf(w, v0.y, a0.y); a0.z = fmaf(w, v0.z, a0.z); a0.w = fmaf(w, v0.w, a0.w);
                a1.x = fmaf(w, v1.x, a1.x); a1.y = fmaf(w, v1.y, a1.y); a1.z = fmaf(w, v1.z, a1.z); a1.w = fmaf(w, v1.w, a1.w);
                a2.x = fmaf(w, v2.x, a2.x); a2.y = fmaf(w, v2.y, a2.y); a2.z = fmaf(w, v2.z, a2.z); a2.w = fmaf(w, v2.w, a2.w);
                a3.x = fmaf(w, v3.x, a3.x); a3.y = fmaf(w, v3.y, a3.y); a3.z = fmaf(w, v3.z, a3.z); a3.w = fmaf(w, v3.w, a3.w);
            } else {
                const f2 ww = {w, w};
                f2 t;
#define PK(acc, v, lo, hi) t = f2{acc.lo, acc.hi}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(ww), "v"(f2{v.lo, v.hi})); acc.lo = t.x; acc.hi = t.y;
                PK(a0, v0, x, y) PK(a0, v0, z, w) PK(a1, v1, x, y) PK(a1, v1, z, w)
                PK(a2, v2, x, y) PK(a2, v2, z, w) PK(a3, v3, x, y) PK(a3, v3, z, w)
#undef PK
            }
            asm volatile("" : "+v"(w));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w + a2.x + a2.y + a2.z + a2.w + a3.x + a3.y + a3.z + a3.w;
}

// MODE bit0: issue the LDS gathers; bit1: issue the FMAs (scalar); bit2: FMAs packed instead
template <int MODE>
__global__ __launch_bounds__(768) void k_gather(float *out, const int *rows, int iters, int nrows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < nrows * 32; i += 768) reinterpret_cast<float *>(lds)[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const int slot = lane >> 2, j = lane & 3;
    const int chan = 16 * j + 64 * (slot & 1), chan2 = chan ^ 64;
    f4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    const int *rp = rows + (blockIdx.x * 12 + (tid >> 6)) * 16 * 64 + slot;
    for (int it = 0; it < iters; ++it) {
        // one "sample": 4 corner rows (base, +1 row, +W rows, +W+1), 2 x ds_read_b128 each
        const int r = rp[(it & 63) * 16];
        const unsigned base = (unsigned)r * 128u;
        float w = 0.25f + 1e-6f * it;
        f4 ra[4], rb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned o = base + (k & 1) * 128u + (k >> 1) * 20u * 128u;
            if (MODE & 1) {
                ra[k] = *reinterpret_cast<const __attribute__((address_space(3))) f4 *>(o + chan);
                rb[k] = *reinterpret_cast<const __attribute__((address_space(3))) f4 *>(o + chan2);
            } else {
                ra[k] = f4{(float)o, 1.f, 2.f, 3.f};
                rb[k] = f4{(float)o, 2.f, 3.f, 4.f};
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE & 4) {
                const f2 ww = {w, w};
                f2 t;
#define PK(acc, v, lo, hi) t = f2{acc.lo, acc.hi}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(ww), "v"(f2{v.lo, v.hi})); acc.lo = t.x; acc.hi = t.y;
                PK(acc0, ra[k], x, y) PK(acc0, ra[k], z, w) PK(acc1, rb[k], x, y) PK(acc1, rb[k], z, w)
#undef PK
            } else if (MODE & 2) {
                acc0.x = fmaf(w, ra[k].x, acc0.x); acc0.y = fmaf(w, ra[k].y, acc0.y);
                acc0.z = fmaf(w, ra[k].z, acc0.z); acc0.w = fmaf(w, ra[k].w, acc0.w);
                acc1.x = fmaf(w, rb[k].x, acc1.x); acc1.y = fmaf(w, rb[k].y, acc1.y);
                acc1.z = fmaf(w, rb[k].z, acc1.z); acc1.w = fmaf(w, rb[k].w, acc1.w);
            } else {
                acc0.x += ra[k].x; acc1.x += rb[k].x;
            }
            w += 0.125f;
        }
    }
    out[blockIdx.x * 768 + tid] = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d; hipMalloc(&d, 64 << 20);
    const int iters = 4000;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        const int blocks = 256 * bpc;
        const float m0 = time_ms([&] { k_fma<0><<<blocks, 256>>>(d, iters, 0.5f); });
        const float m1 = time_ms([&] { k_fma<1><<<blocks, 256>>>(d, iters, 0.5f); });
        const double fl = 2.0 * 128 * iters * 256.0 * blocks;
        printf("fma probe, %d waves/SIMD: v_fma_f32 %.3f ms = %.1f TF/s; v_pk_fma_f32 %.3f ms = %.1f TF/s\n", bpc,
               m0, fl / m0 / 1e9, m1, fl / m1 / 1e9);
    }
    // gather probe: 768-thread workgroups, one per CU, window of `nrows` 128-B rows in LDS
    const int nrows = 1100, giters = 2000, blocks = 256;
    int *rows; hipMalloc(&rows, blocks * 12 * 16 * 64 * sizeof(int));
    {
        int *h = (int *)malloc(blocks * 12 * 16 * 64 * sizeof(int));
        for (int pattern = 0; pattern < 2; ++pattern) {
            for (int b = 0; b < blocks * 12; ++b)
                for (int it = 0; it < 64; ++it)
                    for (int s = 0; s < 16; ++s)
                        h[(b * 64 + it) * 16 + s] = pattern == 0 ? (rand() % (nrows - 24))
                                                                 : ((it * 7 + b * 13) % 900 + s);   // consecutive pixels
            hipMemcpy(rows, h, blocks * 12 * 16 * 64 * sizeof(int), hipMemcpyHostToDevice);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            const size_t lds = (size_t)nrows * 128;
            const float g1 = time_ms([&] { k_gather<1><<<blocks, 768, lds>>>(d, rows, giters, nrows); });
            const float g2 = time_ms([&] { k_gather<2><<<blocks, 768, lds>>>(d, rows, giters, nrows); });
            const float g3 = time_ms([&] { k_gather<3><<<blocks, 768, lds>>>(d, rows, giters, nrows); });
            const float g4 = time_ms([&] { k_gather<4><<<blocks, 768, lds>>>(d, rows, giters, nrows); });
            const float g5 = time_ms([&] { k_gather<5><<<blocks, 768, lds>>>(d, rows, giters, nrows); });
            // per launch: blocks * 12 waves * giters samples * 16 queries * 4 rows * 128 B
            const double bytes = (double)blocks * 12 * giters * 16 * 4 * 128;
            printf("gather probe (%s rows): LDS only %.3f ms (%.1f TB/s) | fma only %.3f | LDS+fma %.3f (%.1f TB/s) | pk only %.3f | LDS+pk %.3f (%.1f TB/s)\n",
                   pattern == 0 ? "random" : "consecutive", g1, bytes / g1 / 1e9, g2, g3, bytes / g3 / 1e9, g4, g5, bytes / g5 / 1e9);
        }
        free(h);
    }
    return 0;
}
