// LDS gather-rate probe for the MSDA forward (round 3): 128-B value rows in LDS, 4 lanes per
// query, 16 queries per wave; which instruction / lane->bank assignment sustains the most bytes?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/lds_gather_probe tools/probes/lds_gather_probe.hip
// Variants: B128 with the first-half choice h(slot) = slot&1 (round 2), (slot>>2)&1, 0;
//           B64 with sub-piece rotation pi(slot) = (slot>>1)&3 (conflict-free for row strides 1, 1/2, 1/4, 1/8).
// Row patterns per wave-instruction: rows[slot] = floor(x0 + slot * stride) for stride in {1, .5, .25, 2}, or random.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) f2 lds_f2;

template <int VARIANT, int THREADS>   // 0: b128 h=slot&1, 1: b128 h=(slot>>2)&1, 2: b128 h=0, 3: b64 rotation
__global__ __launch_bounds__(THREADS) void k_gather(float *out, const int *rows, int iters, int nrows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < nrows * 32; i += THREADS) reinterpret_cast<float *>(lds)[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const int slot = lane >> 2, j = lane & 3;
    const int h = VARIANT == 0 ? (slot & 1) : VARIANT == 1 ? ((slot >> 2) & 1) : 0;
    const int chan = 16 * j + 64 * h;
    const int pi = (slot >> 1) & 3;
    int c[4];
    for (int k = 0; k < 4; ++k) c[k] = ((pi + k) & 3) * 32 + 8 * j;
    f4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    // the 64 row indices of this quad are loaded once: no memory access in the timed loop but LDS
    const int *rp = rows + ((blockIdx.x * (THREADS / 64) + (tid >> 6)) & 1023) * 16 * 64 + slot;
    int mine[8];
    for (int i = 0; i < 8; ++i) mine[i] = rp[i * 16];
    float w = 0.25f;
    for (int it0 = 0; it0 < iters; it0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned base = (unsigned)(mine[u] + ((it0 >> 3) & 31) * 2) * 128u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                 // 4 corner rows: base, +1 row, +20 rows, +21 rows
            const unsigned o = base + (k & 1) * 128u + (k >> 1) * 20u * 128u;
            if (VARIANT < 3) {
                const f4 a = *reinterpret_cast<const lds_f4 *>(o + chan);
                const f4 b = *reinterpret_cast<const lds_f4 *>((o + chan) ^ 64u);
                acc0 += a * w; acc1 += b * w;
            } else {
                const f2 p0 = *reinterpret_cast<const lds_f2 *>(o + c[0]);
                const f2 p1 = *reinterpret_cast<const lds_f2 *>(o + c[1]);
                const f2 p2 = *reinterpret_cast<const lds_f2 *>(o + c[2]);
                const f2 p3 = *reinterpret_cast<const lds_f2 *>(o + c[3]);
                acc0.x += p0.x * w; acc0.y += p0.y * w; acc0.z += p1.x * w; acc0.w += p1.y * w;
                acc1.x += p2.x * w; acc1.y += p2.y * w; acc1.z += p3.x * w; acc1.w += p3.y * w;
            }
        }
        w += 1e-6f;
      }
    }
    out[blockIdx.x * THREADS + tid] = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int V, int THREADS>
void run(const char *name, float *d, const int *rows, int blocks, size_t lds, int nrows) {
    const int giters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<V, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const float ms = time_ms([&] { k_gather<V, THREADS><<<blocks, THREADS, lds>>>(d, rows, giters, nrows); });
    const double bytes = (double)blocks * (THREADS / 64) * giters * 16 * 4 * 128;
    printf("  %-28s %.3f ms  %.1f TB/s  (%.0f B/clk/CU at 2.4 GHz)\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3 * 1e3 / 1e0 / 1e3);
}

int main() {
    float *d; (void)hipMalloc(&d, 64 << 20);
    const int nrows = 590;
    int *rows; (void)hipMalloc(&rows, 1024 * 16 * 64 * sizeof(int));
    int *h = (int *)malloc(1024 * 16 * 64 * sizeof(int));
    const double strides[] = {1.0, 0.5, 0.25, 2.0, -1.0};
    for (double stride : strides) {
        for (int b = 0; b < 1024; ++b)
            for (int it = 0; it < 64; ++it) {
                const double x0 = (rand() % 4000) / 10.0;
                for (int s = 0; s < 16; ++s)
                    h[(b * 64 + it) * 16 + s] = stride < 0 ? rand() % (nrows - 130) : ((int)floor(x0 + s * stride)) % (nrows - 130);
            }
        (void)hipMemcpy(rows, h, 1024 * 16 * 64 * sizeof(int), hipMemcpyHostToDevice);
        printf("row stride per query %.2f%s, 2 workgroups of 384 threads per CU:\n", stride, stride < 0 ? " (random rows)" : "");
        const size_t lds = (size_t)nrows * 128;
        run<0, 384>("b128, h = slot & 1", d, rows, 512, lds, nrows);
        run<1, 384>("b128, h = (slot >> 2) & 1", d, rows, 512, lds, nrows);
        run<2, 384>("b128, h = 0", d, rows, 512, lds, nrows);
        run<3, 384>("b64, rotation (slot >> 1) & 3", d, rows, 512, lds, nrows);
    }
    free(h);
    return 0;
}
