// Micro-probe 3 (not product code): LDS accumulation primitives on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: ds_add_f32, 1: ds_add_u32, 2: ds_add_u64 (2 dwords/lane), 3: non-atomic read+add+write f32 (racy, rate only)
// 4: ds_add_f32 conflict-free addresses (lane-linear within the wave), 5: ds_add_u32 conflict-free
template <int MODE>
__global__ void lds_acc(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned tile[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tile[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 31, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const unsigned wl = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % 256;
        if (MODE == 0) atomicAdd((float *)tile + r * 32 + lane, 1.0f);
        if (MODE == 1) atomicAdd(tile + r * 32 + lane, 1u);
        if (MODE == 2) atomicAdd((unsigned long long *)tile + r * 16 + (lane & 15), 1ull);
        if (MODE == 3) { float *p = (float *)tile + r * 32 + lane; *p = *p + 1.0f; }
        if (MODE == 4) atomicAdd((float *)tile + ((hash(it) % 128) * 64 + wl), 1.0f);
        if (MODE == 5) atomicAdd(tile + ((hash(it) % 128) * 64 + wl), 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)tile[5];
}
template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    float *out; CK(hipMalloc(&out, 1 << 20));
    const int blocks = 2048, threads = 256, iters = 1024;
    const double lanes = (double)blocks * threads;
#define RUNL(NAME, MODE, PER) { float t = timeit([&] { hipLaunchKernelGGL(lds_acc<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters); }); \
    printf("%-52s %.1f G lane-ops/s (%.2f per clk per CU @2.4GHz)\n", NAME, lanes * iters * PER / t / 1e6, lanes * iters * PER / t / 1e6 / 256 / 2.4); }
    RUNL("LDS ds_add_f32 random rows (32 lanes/row)", 0, 1)
    RUNL("LDS ds_add_u32 random rows", 1, 1)
    RUNL("LDS ds_add_u64 random rows (16 lanes/row)", 2, 1)
    RUNL("LDS read+add+write f32 (non-atomic)", 3, 1)
    RUNL("LDS ds_add_f32 lane-linear", 4, 1)
    RUNL("LDS ds_add_u32 lane-linear", 5, 1)
    return 0;
}
