// Micro-probe (not product code): float-atomic and gather rates on MI355X, used to size the
// MSDA backward / forward designs.  hipcc --offload-arch=gfx950 -O3 atomics_probe.hip -o atomics_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// rows: number of 128-B rows in the target region; each 8-lane group picks a random row, each lane
// adds 4 floats at 16-B stride (the MSDA backward pattern). SCOPE: 0 agent, 1 workgroup, 2 wavefront
template <int SCOPE>
__global__ void atomic_rows(float *buf, unsigned rows, int iters, int per_xcd_partition) {
    const unsigned lane = threadIdx.x & 7, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    unsigned xcc = 0;
    if (per_xcd_partition) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
    const unsigned part = per_xcd_partition ? rows / 8 : rows;
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % part + (per_xcd_partition ? xcc * part : 0);
        float *p = buf + (size_t)r * 32 + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (SCOPE == 0) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (SCOPE == 1) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (SCOPE == 2) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}

// gather: each 8-lane group loads random 128-B rows (float4 per lane), accumulates
__global__ void gather_rows(const float4 *buf, unsigned rows, int iters, float4 *out) {
    const unsigned lane = threadIdx.x & 7, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 8
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % rows;
        float4 v = buf[(size_t)r * 8 + lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// LDS float atomics: ds_add_f32 on random rows of a 32 KB tile
__global__ void lds_atomic_rows(float *out, int iters) {
    __shared__ float tile[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 7, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    for (int it = 0; it < iters; ++it) {
        unsigned r = hash(grp * 131u + it) % 256;
        float *p = tile + r * 32 + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(p + c, 1.0f);
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
    const size_t bytes = 64ull << 20;
    float *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    float4 *out; CK(hipMalloc(&out, 2048 * 256 * sizeof(float4)));
    const int blocks = 2048, threads = 256, iters = 64;
    const double groups = (double)blocks * threads / 8;
    struct R { const char *name; unsigned rows; } regs[] = {{"35KB(273 rows)", 273}, {"2.8MB", 22223}, {"22MB", 177784}, {"64MB", 524288}};
    for (auto &r : regs) {
        for (int part = 0; part < 2; ++part) {
            if (part && r.rows < 1024) continue;
            float t0 = timeit([&] { hipLaunchKernelGGL(atomic_rows<0>, dim3(blocks), dim3(threads), 0, 0, buf, r.rows, iters, part); });
            float t1 = timeit([&] { hipLaunchKernelGGL(atomic_rows<1>, dim3(blocks), dim3(threads), 0, 0, buf, r.rows, iters, part); });
            float t2 = timeit([&] { hipLaunchKernelGGL(atomic_rows<2>, dim3(blocks), dim3(threads), 0, 0, buf, r.rows, iters, part); });
            double n = groups * iters * 32;  // float atomics
            printf("atomics region %-16s xcd_part=%d : agent %.1f Gatom/s  workgroup %.1f  wavefront %.1f\n", r.name, part, n / t0 / 1e6, n / t1 / 1e6, n / t2 / 1e6);
        }
    }
    struct G { const char *name; unsigned rows; } gs[] = {{"16KB", 128}, {"2.8MB", 22223}, {"22MB", 177784}, {"64MB", 524288}};
    for (auto &g : gs) {
        float t = timeit([&] { hipLaunchKernelGGL(gather_rows, dim3(blocks), dim3(threads), 0, 0, (const float4 *)buf, g.rows, 256, out); });
        double b = groups * 256 * 128.0;
        printf("gather 128B rows from %-6s : %.2f TB/s\n", g.name, b / t / 1e9);
    }
    float tl = timeit([&] { hipLaunchKernelGGL(lds_atomic_rows, dim3(blocks), dim3(threads), 0, 0, (float *)out, 256); });
    printf("LDS ds_add_f32: %.1f Gatom/s\n", groups * 256 * 32 / tl / 1e6);
    return 0;
}
