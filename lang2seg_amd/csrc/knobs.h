// Tunables that the A/B tools vary.  In the product library (the default build, liblang2seg_hip.so) every one of them is a compile-time
// constant: the C ABI carries no process-global state and exports no setter.  tools/build_tools_lib.py compiles the same sources with
// -DL2S_TOOLS into lang2seg_amd/lib/liblang2seg_hip_tools.so, where they are variables behind l2s_tools_set(name, value) - that library
// is loaded only by tools/ab.py and the tools/*bench*.py / *stamps.py scripts, never by the package's default path, bench.py or tests.
#pragma once

#ifdef L2S_TOOLS
#define L2S_KNOB(name, value) extern int name;
#else
#define L2S_KNOB(name, value) constexpr int name = value;
#endif

namespace l2s_knobs {
L2S_KNOB(pdma_wgs, 0)              // resident workgroups of the persistent LDS-DMA tile (0 = one per CU)
L2S_KNOB(dma256_auto, 1)           // wide plain GEMMs (N >= 1024, N % 256 == 0, M >= 4096) take the 256x256 LDS-DMA tile
L2S_KNOB(wgrad_grid_cap, 0)        // > 0: at most this many workgroups per grouped weight-gradient launch (cap | variant mask << 16)
L2S_KNOB(wgrad_row3_dma, 1)        // large 3x3 weight gradients on the LDS-DMA filter-row tile
L2S_KNOB(wgrad_row3_dma_wgs, 160)  // workgroups of its stream-K launch (round 4, 96 / 128 / 160 / 256 -> 190.1 / 189.1 / 189.0 / 185.5 img/s; round 5, with the shorter
                                   // proposal chain and the split tail, 96 / 128 / 160 / 176 / 192 -> 213.2 / 218.3 / 220.8 / 218.3 / 218.6 img/s, same box x 2)
L2S_KNOB(wgrad_row3_wide, 1)       // 512+ channels on both sides: that tile from any pixel count
L2S_KNOB(wgrad_row3_min_m, 8192)   // pixels from which a 3x3 problem takes it
L2S_KNOB(wgrad_1x1_dma, 0)         // the LDS-DMA 256x256 tile for the large 1x1 problems (built, tested, no faster: conv_wgrad_dma1.hip)
L2S_KNOB(row3_form, 0)             // knock-out mask of the filter-row kernel (results are then garbage; tools/wgrad_stamps.py)
L2S_KNOB(row3_plan_mode, 0)        // 1: the XCD-lockstep stream-K plan where it applies (272-281 us against 265-268)
L2S_KNOB(sgd_blocks, 256)          // persistent workgroups of the update: one per CU (profiles/r04_sgd_blocks.txt)
}  // namespace l2s_knobs
#undef L2S_KNOB
