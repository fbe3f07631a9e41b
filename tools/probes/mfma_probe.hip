// fp32 MFMA issue-rate probe: waves per SIMD x independent accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/mfma_probe tools/probes/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SMALL>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) {
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    }
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SMALL) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int e = 0; e < 16; ++e) s += acc[i][e]; for (int e = 0; e < 4; ++e) s += acc4[i][e]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// distinct A / B registers for every MFMA of the burst (as a real GEMM inner loop has)
__global__ __launch_bounds__(256) void k_regs(float *out, int iters, float a0, float b0, long long *clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 + i; }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + 1) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + 1) & 7], b[t], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + 1) & 7], b[(t + 1) & 7], acc[3], 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool SMALL>
void run(const char *name, int blocks_per_cu, float *d) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, SMALL><<<blocks, 256>>>(d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    k<NACC, SMALL><<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops_per = SMALL ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    const double n = (double)blocks * 4 * iters * NACC;               // wave-level MFMAs
    const double per_simd = n / 1024.0;                                 // 256 CUs x 4 SIMDs
    printf("%-34s %d wave(s)/SIMD: %7.1f TF/s, %6.1f ns per MFMA and SIMD (= %5.1f cycles @2.4 GHz)\n", name,
           blocks_per_cu, n * flops_per / ms / 1e9, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<4, false>("32x32x2 f32, 4 accumulators", 1, d);
    run<4, false>("32x32x2 f32, 4 accumulators", 2, d);
    run<1, false>("32x32x2 f32, 1 accumulator (dep.)", 1, d);
    run<4, true>("16x16x4 f32, 4 accumulators", 1, d);
    run<4, true>("16x16x4 f32, 4 accumulators", 2, d);
    {
        long long *clk; hipMalloc(&clk, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 5000;
        k_regs<<<256, 256>>>(d, 10, 1.f, 2.f, clk);
        hipEventRecord(e0);
        k_regs<<<256, 256>>>(d, iters, 1.f, 2.f, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double n = 32.0 * iters;
        printf("32x32x2 f32, distinct operand registers: %.1f ns per MFMA and SIMD; clock64 ticks per MFMA %.1f; "
               "wall_clock64 (100 MHz) ticks %lld => clock64 runs at %.0f MHz\n", ms * 1e6 / n, h[0] / n, h[1],
               h[0] / (h[1] / 100.0));
    }
    return 0;
}
