// Host/device shared description of the query tiling used by msda_bwd_tiled.hip.
#pragma once
#include <cstdint>

#define DATR_TILE_W 16            // pyramid mode: 16 x 8 queries of one level per workgroup
#define DATR_TILE_H 8
#define DATR_TILE_LINEAR 128      // linear mode: 128 consecutive queries per workgroup
#define DATR_TILED_MAX_LEVELS 8

struct DatrTileLevel {
    int H, W, start;        // geometry (rows, cols, first index)
    int tiles_x, tiles_y;   // number of tile_w x tile_h tiles covering it
    int tile_base;          // index of this level's first tile
};

struct DatrTiledMeta {
    int L;                                   // target (value) levels
    DatrTileLevel lv[DATR_TILED_MAX_LEVELS]; // target level geometry (tiles_* unused)
    // how the Lq queries are cut into workgroup tiles: in pyramid mode (Lq == S) the queries ARE
    // the target pixels, QL == L and qlv == lv geometry; in linear mode QL == 1 and qlv[0] is
    // the 1 x Lq strip of queries
    int QL;
    DatrTileLevel qlv[DATR_TILED_MAX_LEVELS];
    int tile_w, tile_h;
    int total_tiles;
    int Lq;
};

// Entry points one .hip file of the library calls in another: C linkage, NOT exported from
// libdatr_hip.so.  The one exception is declared in include/datr_hip_internal.h (a test hook).
#define DATR_INTERNAL extern "C" __attribute__((visibility("hidden")))

DATR_INTERNAL int datr_internal_msda_bwd_tiled_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const DatrTiledMeta *meta, int64_t N, int64_t S, int64_t M, int64_t P,
    float *grad_value, float *grad_loc, float *grad_attn, void *stream);

DATR_INTERNAL int datr_internal_msda_fwd_pyr_d32(
    const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L,
    int64_t Lq, int64_t P, float *out, void *stream);

// phased all-LDS forward (msda_fwd_pyr2.hip); envelope_host: float[8][4][4] or NULL
extern "C" int datr_internal_msda_fwd_pyr2_d32(   // exported: include/datr_hip_internal.h
   
    const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, const float *envelope_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream);
struct Pyr2Meta;
DATR_INTERNAL int datr_internal_msda_fwd_pyr2_plan(const int64_t *shapes_host, const int64_t *level_start_host,
                                                int64_t S, int64_t M, const float *envelope_host,
                                                Pyr2Meta *pm_out, int32_t *info);
DATR_INTERNAL int datr_internal_msda_bwd_pyr_plan(const int64_t *shapes_host, const int64_t *level_start_host,
                                               int64_t S, int64_t M, int32_t *info);

DATR_INTERNAL int datr_internal_msda_bwd_pyr_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const int64_t *shapes_host, const int64_t *level_start_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, const float *envelope_host, float *grad_value,
    float *grad_loc, float *grad_attn, void *stream, int query_grad = 0);

DATR_INTERNAL int datr_internal_msda_bwd_dots_pyr2_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn, const int64_t *shapes_host,
    const int64_t *level_start_host, const float *envelope_host, int64_t N, int64_t S, int64_t M,
    int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_loc, float *grad_attn, void *stream,
    int query_grad = 0);

DATR_INTERNAL int datr_internal_msda_bwd_owner_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const DatrTiledMeta *meta, int64_t N, int64_t S, int64_t M, int64_t P, int64_t Lq,
    float *grad_value, int64_t grad_value_row_stride, float *grad_loc, float *grad_attn, void *stream);
