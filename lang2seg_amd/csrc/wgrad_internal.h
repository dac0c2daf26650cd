// shared between conv_wgrad.hip (dispatch of the grouped launches) and conv_wgrad_dma.hip (the LDS-DMA filter-row tile)
#pragma once
#include "../../include/lang2seg_hip.h"
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace l2s {
// stream-K plan of one launch: problem i owns the (tile, slice) units [unit0[i], unit0[i + 1]), S[i] slices per tile; U units in all
// mode 1 (XCD-lockstep, see conv_wgrad_dma.hip): T tiles per (problem, filter row) group, Ng groups, k1 workgroups per tile and XCD
struct wgrad_sk_plan { int n; long U; long unit0[L2S_WGRAD_MAX_GROUP + 1]; int S[L2S_WGRAD_MAX_GROUP]; int mode, T, Ng, k1, Q, r, m; };
bool wgrad_row3_dma_ok(const l2s_wgrad_prob& q);
long wgrad_row3_dma_tiles(int Cin, int Cout);
size_t wgrad_row3_dma_ws_bytes(int G);
// the LDS-DMA 256x256 tile of the large 1x1 problems (conv_wgrad_dma1.hip): one workgroup per tile, problem i owns tiles [tile0[i], tile0[i + 1])
struct wgrad_tile_prefix { int n; int tile0[L2S_WGRAD_MAX_GROUP + 1]; };
bool wgrad_1x1_dma_ok(const l2s_wgrad_prob& q);
long wgrad_1x1_dma_tiles(int Cin, int Cout);
int wgrad_1x1_dma_launch(const l2s_wgrad_prob* tab_dev, const l2s_wgrad_prob* tab_host, int nprob, hipStream_t st);
int wgrad_row3_dma_launch(const l2s_wgrad_prob* tab_dev, const l2s_wgrad_prob* tab_host, int nprob, float* ws, size_t ws_bytes, int G, hipStream_t st);
}
