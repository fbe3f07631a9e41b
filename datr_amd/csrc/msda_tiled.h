// Host/device shared description of the query tiling used by msda_bwd_tiled.hip.
#pragma once
#include <cstdint>

#define DATR_TILE_W 16
#define DATR_TILE_H 8
#define DATR_TILED_MAX_LEVELS 8

struct DatrTileLevel {
    int H, W, start;        // level geometry (rows, cols, first token index)
    int tiles_x, tiles_y;   // number of DATR_TILE_W x DATR_TILE_H query tiles covering it
    int tile_base;          // index of this level's first tile
};

struct DatrTiledMeta {
    int L;
    int total_tiles;
    DatrTileLevel lv[DATR_TILED_MAX_LEVELS];
};

extern "C" int datr_internal_msda_bwd_tiled_d32(
    const float *grad_out, const float *value, const float *loc, const float *attn,
    const DatrTiledMeta *meta, int64_t N, int64_t S, int64_t M, int64_t P,
    float *grad_value, float *grad_loc, float *grad_attn, void *stream);
