// Standalone A/B harness for the MSDA forward kernels of libdatr_hip.so (no torch: starts in
// milliseconds on a fresh GPU box).  Builds BASELINE's encoder call (N=2, 1333x800 pyramid,
// M=8, D=32, L=P=4, Lq=S=22223), runs the row kernel (datr_msda_forward_f32) and the
// query-tiled kernel (datr_msda_forward_tiled_f32), compares them and times both.
//
//   hipcc -O2 -o /tmp/fwd_probe tools/probes/fwd_probe.cpp -Iinclude -Ldatr_amd/lib -ldatr_hip
//   LD_LIBRARY_PATH=datr_amd/lib /tmp/fwd_probe [model|wide|uniform] [iters]
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "datr_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const char *dist = argc > 1 ? argv[1] : "model";
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    const int N = argc > 3 ? atoi(argv[3]) : 2;
    const int M = 8, D = 32, L = 4, P = 4;
    const int64_t shapes[8] = {100, 167, 50, 84, 25, 42, 13, 21};
    int64_t lsi[4], S = 0;
    for (int l = 0; l < L; ++l) { lsi[l] = S; S += shapes[2 * l] * shapes[2 * l + 1]; }
    const int64_t Lq = S;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> G(0.f, 1.f);
    std::vector<float> value((size_t)N * S * M * D), loc((size_t)N * Lq * M * L * P * 2),
        attn((size_t)N * Lq * M * L * P);
    for (auto &v : value) v = U(rng) * 0.01f;
    for (size_t i = 0; i < attn.size(); i += L * P) {
        float s = 0.f;
        for (int k = 0; k < L * P; ++k) { attn[i + k] = std::exp(G(rng)); s += attn[i + k]; }
        for (int k = 0; k < L * P; ++k) attn[i + k] /= s;
    }
    const float wide = !strcmp(dist, "wide") ? 6.f : (!strcmp(dist, "wide2") ? 12.f : 0.f);
    for (int n = 0; n < N; ++n) {
        int64_t q = 0;
        for (int lq = 0; lq < L; ++lq)
            for (int y = 0; y < shapes[2 * lq]; ++y)
                for (int x = 0; x < shapes[2 * lq + 1]; ++x, ++q) {
                    const float rx = (x + 0.5f) / shapes[2 * lq + 1], ry = (y + 0.5f) / shapes[2 * lq];
                    for (int m = 0; m < M; ++m) {
                        const float th = 2.f * 3.14159265f * m / M;
                        float cx = std::cos(th), sy = std::sin(th);
                        const float mx = std::max(std::fabs(cx), std::fabs(sy));
                        cx /= mx; sy /= mx;
                        for (int l = 0; l < L; ++l)
                            for (int p = 0; p < P; ++p) {
                                float *o = &loc[((((size_t)(n * Lq + q) * M + m) * L + l) * P + p) * 2];
                                if (!strcmp(dist, "uniform")) { o[0] = U(rng); o[1] = U(rng); continue; }
                                // learned offsets are not integers: jitter so corners are generic
                                const float ox = cx * (p + 1) + (U(rng) - 0.5f) * (1.f + 2.f * wide);
                                const float oy = sy * (p + 1) + (U(rng) - 0.5f) * (1.f + 2.f * wide);
                                o[0] = rx + ox / shapes[2 * l + 1];
                                o[1] = ry + oy / shapes[2 * l];
                            }
                    }
                }
    }
    float *dv, *dl, *da, *o1, *o2;
    int64_t *dsh, *dls;
    CK(hipMalloc(&dv, value.size() * 4)); CK(hipMalloc(&dl, loc.size() * 4));
    CK(hipMalloc(&da, attn.size() * 4));
    const size_t on = (size_t)N * Lq * M * D;
    CK(hipMalloc(&o1, on * 4)); CK(hipMalloc(&o2, on * 4));
    CK(hipMalloc(&dsh, 64)); CK(hipMalloc(&dls, 32));
    CK(hipMemcpy(dv, value.data(), value.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, loc.data(), loc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, attn.data(), attn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsh, shapes, 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(dls, lsi, 32, hipMemcpyHostToDevice));
    CK(hipMemset(o1, 0xff, on * 4)); CK(hipMemset(o2, 0xff, on * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto rows = [&] { return datr_msda_forward_f32(dv, dsh, dls, dl, da, N, S, M, D, L, Lq, P, o1, st); };
    auto tiled = [&] { return datr_msda_forward_tiled_f32(dv, dsh, dls, shapes, lsi, dl, da, N, S, M, D, L, Lq, P, o2, st); };
    int rc = rows(); if (rc) { printf("rows rc=%d\n", rc); return 1; }
    rc = tiled(); if (rc) { printf("tiled rc=%d\n", rc); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<float> h1(on), h2(on);
    CK(hipMemcpy(h1.data(), o1, on * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), o2, on * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; size_t bad = 0;
    for (size_t i = 0; i < on; ++i) {
        const double d = std::fabs((double)h1[i] - h2[i]);
        if (!(d <= 1e-6)) ++bad;
        if (d > maxd || d != d) maxd = d;
        maxv = std::max(maxv, (double)std::fabs(h1[i]));
    }
    printf("dist=%s N=%d  max|rows|=%.4g  max|rows-tiled|=%.3g  elems>1e-6: %zu\n", dist, N, maxv, maxd, bad);
    const double bytes = 4.0 * N * (S * M * D + Lq * M * L * P * 3 + Lq * M * D);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 5; ++i) which ? tiled() : rows();
        std::vector<float> ts;
        for (int i = 0; i < iters; ++i) {
            CK(hipEventRecord(e0, st));
            which ? tiled() : rows();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-6s median %.1f us  min %.1f us  -> %.0f GB/s algorithmic\n", which ? "tiled" : "rows",
               ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] / 1e3);
    }
    typedef void (*probe_fn)(unsigned long long *, int);
    if (probe_fn pf = (probe_fn)dlsym(RTLD_DEFAULT, "datr_probe_fwd_phase_cycles")) {
        unsigned long long buf[8];
        pf(buf, 1);
        tiled();
        CK(hipStreamSynchronize(st));
        pf(buf, 1);
        double tot = 0;
        for (auto v : buf) tot += (double)v;
        const char *names[8] = {"setup", "loc+dma-issue", "geometry", "vmcnt-wait", "barrier1", "gather", "barrier2", "epilogue"};
        printf("phase cycles (thread 0 of each block): ");
        for (int i = 0; i < 8; ++i) printf("%s=%.1f%% ", names[i], 100.0 * buf[i] / tot);
        printf(" total=%.1f Mcycles\n", tot / 1e6);
    }
    return bad != 0;
}
