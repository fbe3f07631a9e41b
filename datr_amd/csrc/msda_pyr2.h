// msda_pyr2.h -- host-side plan of the phased pyramid-region MSDA forward (msda_fwd_pyr2.hip).
//
// The image plane is cut into nRy x nRx regions; a 384-thread workgroup (two per CU, 80 KB of LDS
// each) owns all queries (pixels of all four levels) whose reference point lies in one region, for
// one head.  For every level it needs the WINDOW of value rows those queries can reach.  Unlike
// round 2's plan (msda_pyr.h: one symmetric halo) the reach is an ENVELOPE per (head, level):
// offsets in pixels of the sampled level within [oy_lo, oy_hi] x [ox_lo, ox_hi].  A DINO encoder's
// heads look in one direction each (the ring initialisation of
// /root/reference/models/dino/ops/modules/ms_deform_attn.py:59-68: head m's four points sit at
// 1..4 px along direction m), so a head's windows are footprint + ~5 px on one side instead of
// footprint + 11 px on both axes -- half the rows.  The caller measures the envelope
// (datr_amd/msda.py OffsetMonitor) or passes none (symmetric +-4.5 px); samples outside their
// window take the kernel's slow path, so results never depend on it.
//
// All four levels go through LDS, in PHASES that re-use one window buffer: phase p stages the
// windows of a set of levels (greedy: as many consecutive levels as fit), all of the region's
// queries gather their samples of those levels, accumulators stay in registers across phases.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int kP2MaxR = 16;          // regions per axis
constexpr int kP2Heads = 8;          // per-head tables
// Launch configurations.  The waves of a workgroup must spread EVENLY over the 4 SIMDs: with 384
// threads (2,2,1,1 waves per SIMD) the second workgroup only fits beside the first when the
// dispatcher happens to rotate its start SIMD -- measured: one workgroup per CU most of the time.
//   config 0: 512 threads, two workgroups per CU (80 KB of windows each), <= 3 tasks per wave
//   config 1: 256 threads, three per CU (53 KB each), <= 3 tasks per wave -- more independent
//             workgroups to cover each other's fills; level with config 0 at best (DATR_MSDA_PYR2_CONFIG=1).
struct Pyr2Config { int threads, wgs_per_cu, max_tasks; };
constexpr Pyr2Config kP2Configs[2] = {{512, 2, 3}, {256, 3, 3}};
constexpr int p2_waves(const Pyr2Config &c) { return c.threads / 64; }
constexpr int p2_lds_bytes(const Pyr2Config &c) { return (160 / c.wgs_per_cu) * 1024; }
constexpr int p2_window_rows(const Pyr2Config &c) { return (p2_lds_bytes(c) - 1024) / 128; }   // 128-B rows per phase

// Every field the kernel indexes with a run-time (wave-uniform) index is a 32-bit word or an
// aligned group of 16-bit words: hipcc turns those into s_load_dword(x2/x4) from the kernel
// argument segment; sub-dword fields it fetched with per-level global_load_ubyte / ushort whose
// vmcnt waits serialised the location prefetch (81 us of a 145 us skeleton, profiles/r03_msda_fwd.md).
struct P2HeadLevel { int WH, WW, row0, pad; };           // window dims, first 128-B row in its phase
struct Pyr2Meta {
    int H[4], W[4], start[4];
    int nRy, nRx, nph, tpw;
    int config, pad_[3];                                  // index into kP2Configs
    int ph_mask[4];                                      // levels staged in phase p (bit l)
    int yb[kP2MaxR + 1][4], xb[kP2MaxR + 1][4];          // [region index][level]: first query row / col
    P2HeadLevel hl[kP2Heads][4];
    short wy0[kP2Heads][kP2MaxR][4], wx0[kP2Heads][kP2MaxR][4];   // window origin [head][region index][level]
};
static_assert(sizeof(Pyr2Meta) <= 3584, "kernel argument budget");

struct Pyr2Envelope { float v[kP2Heads][4][4]; };        // [head][level]{oy_lo, oy_hi, ox_lo, ox_hi}

inline long p2_ceil_div(long a, long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

inline void p2_symmetric_envelope(Pyr2Envelope &e, float halo) {
    for (int m = 0; m < kP2Heads; ++m)
        for (int l = 0; l < 4; ++l) {
            e.v[m][l][0] = e.v[m][l][2] = -halo;
            e.v[m][l][1] = e.v[m][l][3] = halo;
        }
}

// Fills the geometry part of `pm` for the grid nRy x nRx; returns false when a level's window does
// not fit a phase, a region has too many queries, or a dimension overflows its table type.
// `cost` = estimated fill bytes of the whole launch per image (all heads), for the grid search.
inline bool p2_try_grid(Pyr2Meta &pm, const Pyr2Config &cfg, const Pyr2Envelope &env, int M, int nRy, int nRx,
                        double *cost) {
    pm.nRy = nRy; pm.nRx = nRx;
    for (int axis = 0; axis < 2; ++axis) {
        const int nR = axis ? nRx : nRy;
        const int *dim = axis ? pm.W : pm.H;
        int (*qb)[4] = axis ? pm.xb : pm.yb;
        const long D0 = dim[0];
        for (int l = 0; l < 4; ++l)
            for (int i = 0; i <= nR; ++i) {
                const long b0 = (long)i * D0 / nR;                 // level-0 boundary
                // first pixel of level l whose centre (y + 0.5) / D_l >= b0 / D0
                long y = p2_ceil_div(2 * b0 * dim[l] - D0, 2 * D0);
                y = std::min<long>(std::max<long>(y, 0), dim[l]);
                qb[i][l] = (int)(i == nR ? dim[l] : y);
            }
        for (int m = 0; m < M; ++m)
            for (int l = 0; l < 4; ++l) {
                short (*w0)[kP2MaxR][4] = axis ? pm.wx0 : pm.wy0;
                const float o_lo = env.v[m][l][axis ? 2 : 0], o_hi = env.v[m][l][axis ? 3 : 1];
                int widest = 2;
                for (int i = 0; i < nR; ++i) {
                    double lo = 1e30, hi = -1e30;
                    for (int lq = 0; lq < 4; ++lq) {
                        if (qb[i + 1][lq] <= qb[i][lq]) continue;
                        lo = std::min(lo, (qb[i][lq] + 0.5) / dim[lq] * dim[l] - 0.5);
                        hi = std::max(hi, (qb[i + 1][lq] - 0.5) / dim[lq] * dim[l] - 0.5);
                    }
                    if (lo > hi) { lo = hi = 0; }
                    // 1e-3: the kernel's fp32 pixel coordinates may round across an integer
                    const int a = (int)std::floor(lo + o_lo - 1e-3), b = (int)std::floor(hi + o_hi + 1e-3) + 1;
                    w0[m][i][l] = (short)a;
                    widest = std::max(widest, b - a + 1);
                }
                if (widest > 255) return false;
                (axis ? pm.hl[m][l].WW : pm.hl[m][l].WH) = widest;
            }
    }
    int most = 0;
    for (int i = 0; i < nRy; ++i)
        for (int k = 0; k < nRx; ++k) {
            int c = 0;
            for (int l = 0; l < 4; ++l)
                c += (pm.yb[i + 1][l] - pm.yb[i][l]) * (pm.xb[k + 1][l] - pm.xb[k][l]);
            most = std::max(most, c);
        }
    if (most > p2_waves(cfg) * cfg.max_tasks * 16) return false;
    const int ntasks = (most + 15) / 16;
    pm.tpw = (ntasks + p2_waves(cfg) - 1) / p2_waves(cfg);
    // phases: the same level sets for every head (the largest head decides), greedy in level order
    int rows[kP2Heads][4], worst[4] = {0, 0, 0, 0};
    for (int m = 0; m < M; ++m)
        for (int l = 0; l < 4; ++l) {
            rows[m][l] = (pm.hl[m][l].WH * pm.hl[m][l].WW + 7) & ~7;       // whole 1-KiB LDS-DMA pieces
            worst[l] = std::max(worst[l], rows[m][l]);
        }
    int nph = 0, used = 0;
    memset(pm.ph_mask, 0, sizeof(pm.ph_mask));
    for (int l = 0; l < 4; ++l) {
        if (worst[l] > p2_window_rows(cfg)) return false;
        if (nph == 0 || used + worst[l] > p2_window_rows(cfg)) {
            if (nph == 4) return false;
            ++nph;
            used = 0;
        }
        pm.ph_mask[nph - 1] |= 1 << l;
        used += worst[l];
    }
    pm.nph = nph;
    double fill = 0;
    for (int m = 0; m < M; ++m) {
        for (int p = 0; p < nph; ++p) {
            int r = 0;
            for (int l = 0; l < 4; ++l)
                if (pm.ph_mask[p] >> l & 1) {
                    pm.hl[m][l].row0 = r;
                    r += rows[m][l];
                    fill += rows[m][l] * 128.0;
                }
        }
    }
    if (cost) {
        // fill bytes of all regions of one image plus a fixed charge per workgroup (prologue)
        const double wgs = (double)nRy * nRx * M;
        *cost = fill * nRy * nRx + wgs * 2048.0;
    }
    return true;
}

// Plan for a 4-level pyramid under one launch configuration: the region grid with the least
// estimated cost.  `force` = "RYxRX" (development) pins the grid.
inline bool build_pyr2_meta_cfg(Pyr2Meta &pm, int config, int M, const Pyr2Envelope &env, const char *force) {
    const Pyr2Config &cfg = kP2Configs[config];
    pm.config = config;
    int fy = 0, fx = 0;
    if (force && std::sscanf(force, "%dx%d", &fy, &fx) == 2 && fy >= 1 && fx >= 1 && fy <= kP2MaxR && fx <= kP2MaxR)
        return p2_try_grid(pm, cfg, env, M, fy, fx, nullptr);
    // Measured on the N = 4 call at 1333x800 (profiles/r03_msda_fwd.md, after the location loads lost
    // their non-temporal hint): 8x12 111 us, 7x14 114, 5x12 (two phases) 117, 10x16 127, 12x16 133,
    // 16x16 148 -- the least staged bytes win, an extra phase costs a good 15 %, one task per wave
    // leaves the waves idle through the fills.
    double best = 1e300;
    int by = 0, bx = 0;
    Pyr2Meta trial = pm;
    for (int ry = 1; ry <= kP2MaxR; ++ry)
        for (int rx = 1; rx <= kP2MaxR; ++rx) {
            double c;
            if (!p2_try_grid(trial, cfg, env, M, ry, rx, &c)) continue;
            c *= 1.0 + 0.3 * (trial.nph - 1) + (trial.tpw < 2 ? 0.3 : 0.0);
            if (c < best) { best = c; by = ry; bx = rx; }
        }
    if (!by) return false;
    return p2_try_grid(pm, cfg, env, M, by, bx, nullptr);
}

inline bool build_pyr2_meta(Pyr2Meta &pm, const int64_t *sh, const int64_t *ls, int64_t S, int M,
                            const Pyr2Envelope &env, const char *force = nullptr, int force_config = -1) {
    if (M < 1 || M > kP2Heads) return false;
    long total = 0;
    for (int l = 0; l < 4; ++l) {
        const long H = sh[2 * l], W = sh[2 * l + 1];
        if (H < 1 || W < 1 || H > 4096 || W > 4096 || ls[l] != total) return false;
        pm.H[l] = (int)H; pm.W[l] = (int)W; pm.start[l] = (int)total;
        total += H * W;
    }
    if (total != S || S * (long)M * 128 >= (1L << 31)) return false;
    for (int l = 1; l < 4; ++l)
        if (pm.H[l] > pm.H[l - 1] || pm.W[l] > pm.W[l - 1]) return false;
    // configuration 1 (three 256-thread workgroups per CU) measured level with configuration 0 on
    // the ring envelope (113.6 vs 114.6 us) and behind it for wider ones (158 vs 149 us): development only
    return build_pyr2_meta_cfg(pm, force_config >= 0 ? (force_config & 1) : 0, M, env, force);
}
