// Micro-probe 2 (not product code): how float-atomic throughput depends on the address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// G lanes share one random 128-B row; lane i of the group adds to dword (i*STRIDE + c*CSTEP) % 32, c = 0..PER-1
template <int G, int STRIDE, int CSTEP, int PER>
__global__ void atomic_pat(float *buf, unsigned rows, int iters) {
    const unsigned lane = threadIdx.x % G, grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % rows;
        float *p = buf + (size_t)r * 32;
#pragma unroll
        for (int c = 0; c < PER; ++c) atomicAdd(p + (lane * STRIDE + c * CSTEP) % 32, 1.0f);
    }
}
// same but rows are CONSECUTIVE across groups (streaming flush pattern)
template <int G, int PER>
__global__ void atomic_stream(float *buf, unsigned rows, int iters) {
    const unsigned lane = threadIdx.x % G, grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const unsigned ngrp = gridDim.x * blockDim.x / G;
    for (int it = 0; it < iters; ++it) {
        unsigned r = (grp + it * ngrp) % rows;
        float *p = buf + (size_t)r * 32;
#pragma unroll
        for (int c = 0; c < PER; ++c) atomicAdd(p + (lane + c * G) % 32, 1.0f);
    }
}
// LDS: lane pattern as above on a 32 KB tile
template <int G, int STRIDE, int CSTEP, int PER>
__global__ void lds_pat(float *out, int iters) {
    __shared__ float tile[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x % G, grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % 256;
        float *p = tile + r * 32;
#pragma unroll
        for (int c = 0; c < PER; ++c) atomicAdd(p + (lane * STRIDE + c * CSTEP) % 32, 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[5];
}
template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    float *buf; CK(hipMalloc(&buf, 64ull << 20)); CK(hipMemset(buf, 0, 64ull << 20));
    float *out; CK(hipMalloc(&out, 1 << 20));
    const int blocks = 2048, threads = 256, iters = 64;
    const double lanes = (double)blocks * threads;
    const unsigned rows = 177784;  // 22 MB
#define RUN(NAME, KERN, PER) { float t = timeit([&] { hipLaunchKernelGGL(KERN, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters); }); \
    printf("%-52s %.1f Gatom/s\n", NAME, lanes * iters * PER / t / 1e6); }
    RUN("global G=8 stride16B x4 (current bwd)", (atomic_pat<8, 4, 1, 4>), 4)
    RUN("global G=8 contiguous 32B, x4 at +32B", (atomic_pat<8, 1, 8, 4>), 4)
    RUN("global G=32 contiguous 128B row, x1", (atomic_pat<32, 1, 0, 1>), 1)
    RUN("global G=64 two lanes per dword (same row) x1", (atomic_pat<64, 1, 0, 1>), 1)
    RUN("global G=16 contiguous 64B, x2", (atomic_pat<16, 1, 16, 2>), 2)
    RUN("global stream G=32 consecutive rows x1", (atomic_stream<32, 1>), 1)
    RUN("global stream G=8 consecutive rows x4", (atomic_stream<8, 4>), 4)
#define RUNL(NAME, KERN, PER) { float t = timeit([&] { hipLaunchKernelGGL(KERN, dim3(blocks), dim3(threads), 0, 0, out, 256); }); \
    printf("%-52s %.1f Gatom/s\n", NAME, lanes * 256 * PER / t / 1e6); }
    RUNL("LDS G=8 stride16B x4", (lds_pat<8, 4, 1, 4>), 4)
    RUNL("LDS G=8 contiguous 32B x4", (lds_pat<8, 1, 8, 4>), 4)
    RUNL("LDS G=32 contiguous row x1", (lds_pat<32, 1, 0, 1>), 1)
    return 0;
}
