// msda_pyr.h -- the pyramid-region decomposition shared by the encoder kernels
// (msda_fwd_pyr.hip, msda_bwd_pyr.hip).  The image plane is cut into nRy x nRx regions; a
// workgroup owns all queries (pixels of all four levels) whose reference point lies in one
// region, and for every level the WINDOW of value rows those queries can reach: the region's
// footprint in that level plus a halo.  Host-side geometry only; the kernels take the struct by
// value.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef PYR_MAXR
#define PYR_MAXR 16
#endif
constexpr int kPyrMaxR = PYR_MAXR;               // regions per axis

struct PyrMeta {
    int H[4], W[4], start[4];
    int nRy, nRx;
    int WH[4], WW[4];                            // window dims per level (max over the regions)
    int lds_base[4];                             // byte offset of a level's window in LDS (fwd)
    int lds_bytes;
    short yb[4][kPyrMaxR + 1], xb[4][kPyrMaxR + 1];   // query rows / cols of level l in region i: [b[i], b[i+1])
    short wy0[4][kPyrMaxR], wx0[4][kPyrMaxR];    // window origin (may be negative: outside the image)
    // per-(head, level) trims of the symmetric window (backward, envelope plans): rows / columns cut
    // at the low side (the origin moves by as much) and in total; zero without an envelope
    short cut_y0[8][4], cut_x0[8][4], cut_y[8][4], cut_x[8][4];
};

// ceil(a / b) for b > 0 and any a
inline long pyr_ceil_div(long a, long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

// Fills `pm` for a 4-level pyramid (H, W per level in `sh`, first index per level in `ls`).
// Regions start at about target_h x target_w level-0 pixels and shrink until fits(pm, queries of
// the largest region) holds.  halo: how far (pixels of the sampled level) a sample may lie from
// its query's reference point and still be inside the window.
// `halo` may also be given per level; force_ry / force_rx > 0 try exactly that grid.
template <typename Fits>
inline bool build_pyr_meta(PyrMeta &pm, const int64_t *sh, const int64_t *ls, int64_t S,
                           const float (&halo)[4], double target_h, double target_w, Fits fits,
                           int force_ry = 0, int force_rx = 0) {
    long total = 0;
    for (int l = 0; l < 4; ++l) {
        const long H = sh[2 * l], W = sh[2 * l + 1];
        if (H < 1 || W < 1 || H > 4096 || W > 4096 || ls[l] != total) return false;
        pm.H[l] = (int)H; pm.W[l] = (int)W; pm.start[l] = (int)total;
        total += H * W;
    }
    if (total != S || S * 8 * 128 >= (1L << 31)) return false;
    // level l must be the coarser the larger l (windows are sized for a pyramid)
    for (int l = 1; l < 4; ++l)
        if (pm.H[l] > pm.H[l - 1] || pm.W[l] > pm.W[l - 1]) return false;
    for (int m = 0; m < 8; ++m)
        for (int l = 0; l < 4; ++l) pm.cut_y0[m][l] = pm.cut_x0[m][l] = pm.cut_y[m][l] = pm.cut_x[m][l] = 0;
    const int H0 = pm.H[0], W0 = pm.W[0];
    int nRy = std::min(kPyrMaxR, std::max(1, (int)std::lround(H0 / target_h)));
    int nRx = std::min(kPyrMaxR, std::max(1, (int)std::lround(W0 / target_w)));
    if (force_ry > 0 && force_rx > 0) {
        nRy = std::min(kPyrMaxR, force_ry); nRx = std::min(kPyrMaxR, force_rx);
    } else if (const char *e = std::getenv("DATR_MSDA_PYR_REGIONS")) {          // development: "RYxRX"
        int a = 0, b = 0;
        if (std::sscanf(e, "%dx%d", &a, &b) == 2 && a >= 1 && b >= 1 && a <= kPyrMaxR && b <= kPyrMaxR) {
            nRy = a; nRx = b;
        }
    }
    for (;;) {
        pm.nRy = nRy; pm.nRx = nRx;
        for (int axis = 0; axis < 2; ++axis) {
            const int nR = axis ? nRx : nRy;
            const int *dim = axis ? pm.W : pm.H;
            short (*qb)[kPyrMaxR + 1] = axis ? pm.xb : pm.yb;
            short (*w0)[kPyrMaxR] = axis ? pm.wx0 : pm.wy0;
            int *wdim = axis ? pm.WW : pm.WH;
            const long D0 = dim[0];
            for (int l = 0; l < 4; ++l) {
                for (int i = 0; i <= nR; ++i) {
                    const long b0 = (long)i * D0 / nR;                 // level-0 boundary
                    // first pixel of level l whose centre (y + 0.5) / D_l >= b0 / D0
                    long y = pyr_ceil_div(2 * b0 * dim[l] - D0, 2 * D0);
                    y = std::min<long>(std::max<long>(y, 0), dim[l]);
                    qb[l][i] = (short)(i == nR ? dim[l] : y);
                }
            }
            for (int l = 0; l < 4; ++l) {
                int widest = 2;
                for (int i = 0; i < nR; ++i) {
                    double lo = 1e30, hi = -1e30;
                    for (int lq = 0; lq < 4; ++lq) {
                        if (qb[lq][i + 1] <= qb[lq][i]) continue;
                        lo = std::min(lo, (qb[lq][i] + 0.5) / dim[lq] * dim[l] - 0.5);
                        hi = std::max(hi, (qb[lq][i + 1] - 0.5) / dim[lq] * dim[l] - 0.5);
                    }
                    if (lo > hi) { lo = hi = 0; }
                    const int a = (int)std::floor(lo - halo[l]), b = (int)std::floor(hi + halo[l]) + 1;
                    w0[l][i] = (short)a;
                    widest = std::max(widest, b - a + 1);
                }
                wdim[l] = widest;
            }
        }
        for (int l = 0; l < 4; ++l) pm.lds_base[l] = 0;
        pm.lds_bytes = 0;
        int most = 0;                              // queries of the largest region
        for (int i = 0; i < nRy; ++i)
            for (int k = 0; k < nRx; ++k) {
                int c = 0;
                for (int l = 0; l < 4; ++l)
                    c += (pm.yb[l][i + 1] - pm.yb[l][i]) * (pm.xb[l][k + 1] - pm.xb[l][k]);
                most = std::max(most, c);
            }
        if (fits(pm, most)) return true;
        if (force_ry > 0 && force_rx > 0) return false;
        // too large: more, smaller regions along the longer region side
        if ((double)H0 / nRy >= (double)W0 / nRx && nRy < kPyrMaxR) ++nRy;
        else if (nRx < kPyrMaxR) ++nRx;
        else if (nRy < kPyrMaxR) ++nRy;
        else return false;
    }
}

template <typename Fits>
inline bool build_pyr_meta(PyrMeta &pm, const int64_t *sh, const int64_t *ls, int64_t S, float halo,
                           double target_h, double target_w, Fits fits) {
    const float h4[4] = {halo, halo, halo, halo};
    return build_pyr_meta(pm, sh, ls, S, h4, target_h, target_w, fits);
}

inline float pyr_halo_from_env() {
    const char *e = std::getenv("DATR_MSDA_PYR_HALO");
    const float h = e ? (float)std::atof(e) : 4.5f;
    return h >= 0.5f && h <= 16.f ? h : 4.5f;
}
