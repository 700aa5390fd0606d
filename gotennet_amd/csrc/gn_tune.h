// gn_tune.h -- per-kernel occupancy targets (waves per SIMD) for the gather kernels.
// The AMDGPU scheduler sizes its load batches by the occupancy it believes it must keep: told "2 waves/SIMD"
// it keeps up to ~23 row loads of an edge in flight (184 VGPRs) instead of ~6 (136 VGPRs), which is what a
// latency-bound gather wants; other kernels are best left alone or run better with MORE, thinner waves.
// Values measured on MI355X with tools/tune_sweep.sh (0 = no hint); -DGN_W_<KERNEL>=n overrides for a sweep.
#pragma once

#define GN_WPE_ATTR(n) __attribute__((amdgpu_waves_per_eu(n, n)))

#ifndef GN_W_K6
#define GN_W_K6 3          // message_aggregate_kernel: 117.5 -> 105.1 us (lmax=2) with 2 in round 2; re-swept on the compile-time-width kernel at the
                           // end of round 5: none 137 us, 2 93.5, 3 90.5, 4 139 (message stage 108.0 -> 105.0 us)
#endif
#ifndef GN_W_K6_G
#define GN_W_K6_G 3        // message_aggregate_group_kernel: 290 -> 250 (2) -> 234 us (3) per layer (lmax=4, three launches)
#endif
#ifndef GN_W_MSG_TGT
#define GN_W_MSG_TGT 0     // 2: 333 -> 348 us (message backward, both passes)
#endif
#ifndef GN_W_MSG_SRC
#define GN_W_MSG_SRC 2     // without it the scheduler serialises every row load to reach 3 waves/SIMD: 393 -> 327 us (both passes)
#endif
#ifndef GN_W_HTR_TGT
#define GN_W_HTR_TGT 0     // 2, 3: within noise; 4: 140 -> 268 us
#endif
#ifndef GN_W_HTR_SRC
#define GN_W_HTR_SRC 0
#endif
#ifndef GN_W_HTR_TGT_G
#define GN_W_HTR_TGT_G 0
#endif
#ifndef GN_W_HTR_SRC_G
#define GN_W_HTR_SRC_G 0
#endif
#ifndef GN_W_HTR_EDGE
#define GN_W_HTR_EDGE 0    // 2: +4 us; 4: lmax=4 104 -> 192 us
#endif
#ifndef GN_W_ATTN
#define GN_W_ATTN 0        // 2: 29 -> 55 us; 4: 37 us
#endif

#ifndef GN_K6_MERGE34
#define GN_K6_MERGE34 1    // lmax = 4: degrees 3 and 4 of the message kernel in ONE launch (16 accumulator rows):
#endif                     // C2 message stage 245.9 -> ~235 us (two launches: 60 + 64 us for 111 MB of t_filter each)
#ifndef GN_W_K6_G34
#define GN_W_K6_G34 2      // ... at 2 waves/SIMD (3: 275.6 us, spills; no hint: 325 us)
#endif
// (lmax = 4 backward: the same {3,4} merge was measured for the four backward passes in round 2 and lost -- message
// backward 676 -> 682 / 716 us per layer for the target / source pass, HTR backward 381 -> 413 / 423 us: 16 extra live
// gradient rows cost more than the saved re-reads; the switches are gone, the passes run one degree per launch.)
#ifndef GN_K6G_CH
#define GN_K6G_CH 16       // message_aggregate_group_kernel: accumulator rows reduced per LDS pass (4 KiB per row): 16 = the
                           // {3,4} group in ONE pass (round 3: 204.5 -> 203.0 us per lmax=4 stage; 5: 209.3; the 64 KiB
                           // cost nothing, the kernel sits at 2 waves/SIMD on registers)
#endif

#ifndef GN_MSGB_MERGED
#define GN_MSGB_MERGED 1   // message backward at lmax <= 2 (general launches): 1 = by-source kernel with the per-edge work merged in
#endif                     // (t_filter read once) + attention backward + g_k; 0 = the by-target / by-source pair
#ifndef GN_MSGB_MERGED_FIRST
#define GN_MSGB_MERGED_FIRST 1   // the first interaction (X_in == 0) through the merged kernel too (scalar + direction-gate blocks only)
#endif
#ifndef GN_HTRB_SRC_ONE
#define GN_HTRB_SRC_ONE 1  // HTR backward at lmax 3 / 4: ONE by-source launch for all degrees (its accumulators are the only rows it keeps)
#endif
#ifndef GN_HTRB_TGT_MODE
#define GN_HTRB_TGT_MODE 1 // ... by-target launches: 0 = {1,2},{3},{4}; 1 = {1,2,3} at lmax 3, {1,2},{3,4} at lmax 4.  Nanotube (lmax 3)
#endif                     // 300 -> 271 us per layer with both, lmax 4 273 either way (the gathered EQ / EK rows, 2 x 24 KiB per edge
                           // through L2, bound it, not the re-read [E,F] streams); own EQ rows in LDS: no gain; a 3-wave hint: spills, 435 / 710 us
#ifndef GN_W_MSG_MRG_G
#define GN_W_MSG_MRG_G 2   // degree-group kernels {scalar,1,2} and {4} ...
#endif
#ifndef GN_W_MSG_MRG_G3
#define GN_W_MSG_MRG_G3 3  // ... and {3} (155 VGPRs): nanotube message backward 440 -> 418 us per layer with 3 for every group, lmax 4 neutral
#endif
#ifndef GN_W_MSG_MRG
#define GN_W_MSG_MRG 0     // merged kernel, general launches: no hint (190 -> 186.5 us per layer at lmax 2 against 2)
#endif
#ifndef GN_W_MSG_MRG_F
#define GN_W_MSG_MRG_F 2   // ... its first-interaction form (no hint: lmax 4 message backward 373 -> 382 us per layer on average)
#endif
#ifndef GN_MSGB_PF3
#define GN_MSGB_PF3 4      // message backward, target pass: trips of the score-backward phase whose rows are requested before
                           // the softmax-backward barriers (0: none)
#endif

// ---- launcher constants (formerly environment variables read inside the extern "C" entry points: the ABI promises
// no hidden process state, so they are build-time constants now; tools/variants.py builds a library per value)
#ifndef GN_GEMM_BIG_MIN
#define GN_GEMM_BIG_MIN 900    // 128 x 128 projection tiles from this many tiles up (below: 64 x 64)
#endif
#ifndef GN_GEMM_BIG_CAP
#define GN_GEMM_BIG_CAP 512    // persistent workgroups of the 128 x 128 slab kernel (two per CU)
#endif
#ifndef GN_GEMM_NT_MB
#define GN_GEMM_NT_MB 100.0    // outputs of this many MiB and more are stored non-temporally ...
#endif
#ifndef GN_GEMM_NT_LO
#define GN_GEMM_NT_LO (-1)     // ... from this column on (-1: the first K columns stay on the normal path)
#endif
#ifndef GN_GEMM_PANEL_MAX
#define GN_GEMM_PANEL_MAX 2048   // f16x2 groups of at most this many 32 x 128 tiles (no prologue; one depth K in {128, 256, 512} or depths
#endif                          // that are multiples of 256) run the K-resident panel kernel (gn_gemm_panel.hip); 0: never.  Inside the C2
                                // step: [2688 x 256 x 256] 14.1 -> 12 us, [2688 x 512 x 256] 15.1 -> 13.4, two [2688 x 1280 x 256] 29.1 ->
                                // 26.6, [8064 + 13440 x 256 x 768] 62 -> 57.9; at 3400-4000 tiles the slab kernel's co-resident workgroups
                                // win (the gated edge product 67 vs 78.7 us, the four-problem X group 47.4 vs 51.7, the K = 1536
                                // input-gradient group 184 vs 243.6)
#ifndef GN_F16_SMALL_WIDE
#define GN_F16_SMALL_WIDE 1      // the small tile of the f16x2 slab kernel: 1 = 32 x 128 (waves 1 x 4) when every N >= 128, 0 = always 64 x 64
                                // (waves 2 x 2).  Stand-alone [54368 x 256 x 256] 50.6 -> 47.2 us, with the gated epilogue 58.5 -> 53.9; in the
                                // step (HBM-bound: res, gate, pre_out) 67 -> 66 us
#endif
#ifndef GN_F16_MID_ROWS
#define GN_F16_MID_ROWS 30000    // f16x2 small-tile groups with a gated residual epilogue and at least this many rows run 64 x 128 tiles
                                 // instead of 32 x 128 (0: never): a weight fragment serves two row tiles
#endif
#ifndef GN_F16_MID_CAP
#define GN_F16_MID_CAP 768       // persistent workgroups of that kernel
#endif
#ifndef GN_F16_SMALL_CAP
#define GN_F16_SMALL_CAP 1024     // persistent workgroups of that kernel
#endif
#ifndef GN_HTR_CLOSED
#define GN_HTR_CLOSED 1        // htr_edge_kernel at lmax = 3: closed form EQ.EK - (2 - r.r)(EQ.r)(EK.r) instead of two rejections
#endif
#ifndef GN_HTR_CLOSED_ALL
#define GN_HTR_CLOSED_ALL 4    // ... and from this lmax up the closed form with EVERY row of the edge requested before the first use: lmax 4
#endif                         // 108.4 -> 90 us per call (degree by degree the closed form lost there, 140 us: the compiler serialised the
                               // row loads); lmax 3 is 90.4 us degree by degree, 92.4 us in this form (0: never)
#ifndef GN_ATTN_WAVE
#define GN_ATTN_WAVE 1         // 1: one wave per target in gn_attn_softmax where the shape allows; 0: workgroup per target
#endif

#define GN_TUNE_CAT_(a, b) a##b
#define GN_TUNE_CAT(a, b) GN_TUNE_CAT_(a, b)
#define GN_WPE_SEL_0
#define GN_WPE_SEL_1 GN_WPE_ATTR(1)
#define GN_WPE_SEL_2 GN_WPE_ATTR(2)
#define GN_WPE_SEL_3 GN_WPE_ATTR(3)
#define GN_WPE_SEL_4 GN_WPE_ATTR(4)
#define GN_WPE_SEL_5 GN_WPE_ATTR(5)
#define GN_WPE_SEL_6 GN_WPE_ATTR(6)
#define GN_WPE_SEL_8 GN_WPE_ATTR(8)
#define GN_WPE(n) GN_TUNE_CAT(GN_WPE_SEL_, n)
