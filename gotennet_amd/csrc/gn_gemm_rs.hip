// gn_gemm_rs.hip -- the K = F = 256 projections with the ACTIVATION ROWS STATIONARY IN REGISTERS ("row-stationary").
//
//   C[r, n] = epi( sum_k A[r, k] W[n, k] + bias[n] ),   K = 256,   2 x fp16-split arithmetic (gn_gemm.h), fp32 accumulate
//
// Replaces, for the products whose reduction length is the feature width (reference layers.py:457-581; call sites
// gotennet.py:400-407 [W_re | W_rs on t, the node projections], 432-441 [W_vq, W_vk_l on X], 611 [gamma_t], 728 [W_vu]
// and the W_t^T input-gradient of the force backward), the LDS-slab kernel of gn_gemm.hip.  Round 2's counters on
// that kernel: 8.5 VALU per MFMA (fp32 -> two fp16 planes, block amax, masks, once per K-slab AND per column tile),
// one barrier per 32-deep slab, matrix pipe 31 % busy.  With K = 256 a whole operand row fits in registers, so:
//
//   * a wave owns 32 rows (one MFMA M-tile) and keeps their 256 k-values as the two fp16 planes in 128 VGPRs, split
//     ONCE per row block and reused for every one of the N / 32 column tiles;
//   * the exponent is PER ROW (amax over the row's 256 values), so a row's result does not depend on its neighbours --
//     the batch-position dependence of the block-exponent kernel is gone for these products;
//   * the main loop has no A traffic at all: per 32-column tile it is 48 MFMAs (16 k-steps x {lo*hi, hi*lo, hi*hi})
//     + 32 ds_read_b128 of the weight fragments, which arrive by LDS-DMA (global_load_lds_dwordx4) -- the packed weight
//     planes are already in fragment-major order, one column tile = one contiguous 32 KiB image -- double-buffered,
//     ONE barrier per column tile (per 48 MFMAs per wave instead of per 24);
//   * the epilogue of tile t - 1 (accumulators -> wave-private LDS patch -> rows of 128 B = one cache line each, with
//     the fused bias / activation / pre-activation copy) is cut into six stages that are issued BETWEEN the MFMA groups
//     of tile t: it runs in the matrix pipe's shadow.  Products with a residual / gate operand (HBM-bound anyway) and
//     ragged row blocks take the same stages back to back before the next tile.
//
// A workgroup is 4 waves = 4 row tiles (128 rows) sharing the weight images; LDS 64 KiB (two images) + 8 KiB (patches):
// two workgroups per CU, i.e. two waves per SIMD.  Work items are (row block, column tile) pairs in row-block-major
// order; a persistent grid cuts the flat list into equal contiguous ranges (XCD-contiguous), so a workgroup re-splits
// its rows at most twice per launch.
#include <cstdlib>
#include <type_traits>
#include "gn_gemm.h"

namespace gn {

constexpr int RS_KS = 16;                         // k-steps of 16
constexpr int RS_BT = RS_KS * 2 * 1024;           // bytes of one weight image: [k-step][plane][lane][16 B]
constexpr int RS_PATCH = 16 * 32;                 // floats of a wave's epilogue patch (16 rows x 32 columns)
constexpr int RS_STG = 32 * 36 * 4 + 256;         // bytes of a wave's row staging piece + its exponent table (NW = 8 only: dedicated)
template <int NW> constexpr int rs_lds() { return 2 * RS_BT + NW * RS_PATCH * 4 + (NW == 8 ? NW * RS_STG : 0); }
constexpr int RS_AP = 36;                         // row pitch (floats) of the 32 x 32 staging piece of load_A

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) unsigned char lds_u8;     // 32-bit LDS pointers: one register instead of a generic pair
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) const f16x8 lds_cf16x8;
typedef __attribute__((address_space(3))) const f32x4_nt lds_cf4;

#define RS_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef GN_RS_PROBE
#define GN_RS_PROBE 0          // probe builds only (tools/gemm_rs_trace.py): per-workgroup residency and per-phase cycle stamps
#endif


// hipcc and LDS-DMA (ROCm 7.2): while a global_load_lds may be in flight hipcc (a) trusts only s_waitcnt vmcnt(0) --
// the first use of ANY loaded register, and any LDS read it cannot prove disjoint from the DMA's destination, drains the
// DMA right there -- and (b) ignores counted waits.  Hiding such loads in inline asm works (it was measured) but is
// fragile: hipcc treats an asm output as available at once, and an unrelated edit that made it copy such a register
// before the data had landed silently corrupted the bias (a probe build caught it).  The kernel therefore keeps every
// load visible and spends ONE explicit vmcnt(0) per tile where it is nearly free: two k-steps into the MFMA phase, when
// the LDS-DMA issued in front of the phase has long landed and the riding epilogue has not stored anything yet.  After
// it hipcc knows that nothing is pending: the bias loaded a tile ago and the epilogue's patch reads need no wait.

#if GN_RS_PROBE
// probe only (GN_RS_ABL bit 128): per-workgroup {start, end} of s_memrealtime (100 MHz) and the hardware id register
__device__ long long rs_trace_buf[1024 * 4];
__device__ long long rs_phase_buf[64 * 2 * 8 * 8];      // probe only (bit 256): [block < 64][wave 0 / 3][tile 8..15][stamp] of s_memtime
#endif
// (SiLU only: with the activation kind as a run-time switch the unrolled stages outgrow hipcc's unroll budget and the
//  operand planes land in scratch; models with another activation stay on the LDS-slab kernel)
// NW = waves per workgroup = 32-row tiles sharing one weight image: 4 (two workgroups per CU) or 8 (one per CU: half
// the LDS-DMA instructions and image traffic per MFMA, one barrier for the CU)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void gemm_rs_f16x2(const GroupArgs ga) {
    constexpr int ROWS = 32 * NW, PPW = 32 / NW;     // rows per workgroup; LDS-DMA pieces per wave and image
    // ONE shared array (a second __shared__ object makes hipcc drain vmcnt before every ds_read of a glds pipeline)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[rs_lds<NW>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int abl = ga.spread;                      // probe runs only (GN_RS_ABL, wrong results): 1 no LDS-DMA, 2 no MFMA phase,
                                                    // 4 no epilogue, 8 rows split once per workgroup, 16 no barrier, 32 never overlap, 64 stores fold onto 128 rows

    // flat tile range of this workgroup; blocks of one XCD (b % 8, observed) get neighbouring ranges: they share rows
    const int G = gridDim.x;
    int wg = blockIdx.x;
    if ((G & 7) == 0) wg = (wg & 7) * (G >> 3) + (wg >> 3);
    int cum[GN_MAX_GROUP];
#pragma unroll
    for (int gi = 0; gi < GN_MAX_GROUP; ++gi) cum[gi] = ga.tile_end[gi < ga.n ? gi : ga.n - 1];
    const long T = cum[GN_MAX_GROUP - 1];
    const int lo = (int)((long)wg * T / G), hi = (int)((long)(wg + 1) * T / G);
    if (lo >= hi) return;
    // probe (GN_RS_ABL >> 16 = start delay in 64-cycle units for the second workgroup of a CU): de-phase the two
    // co-resident workgroups so that one's MFMA phase meets the other's LDS-DMA issue / barrier / epilogue tail
    if ((abl >> 16) && (blockIdx.x >= (unsigned)(G >> 1))) {
        const long long until = __builtin_readcyclecounter() + (long long)(abl >> 16) * 64;
        while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
    }
#if GN_RS_PROBE
    const long long t_start = (abl & 128) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#endif

    // ---- problem / tile bookkeeping (wave-uniform).  A tile is (problem g, row block rb, column tile j); the walk is
    // incremental (no division per tile: the first version spent ~1 us per tile on index arithmetic).
    GemmArgs p = ga.g[0];
    int ewt = 0;
    auto select = [&](int g) {                      // p <- problem g (compile-time indices only: no scratch copy)
#pragma unroll
        for (int gi = 0; gi < GN_MAX_GROUP; ++gi)
            if (gi == g) p = ga.g[gi];
        ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));     // weight tensor exponent (plane header)
    };
    auto weight_of = [&](int g) {                   // packed planes of problem g (header skipped)
        const float* w = ga.g[0].W;
#pragma unroll
        for (int gi = 1; gi < GN_MAX_GROUP; ++gi) w = (gi == g) ? ga.g[gi].W : w;
        return reinterpret_cast<const uint4*>(w) + 16;
    };
    auto ntiles_of = [&](int g) {
        int n = ga.g[0].N;
#pragma unroll
        for (int gi = 1; gi < GN_MAX_GROUP; ++gi) n = (gi == g) ? ga.g[gi].N : n;
        return n >> 5;
    };
    // coordinates of tile lo, then of the tile after it
    int cg = 0, crb, cj;
    {
        int base = 0;
#pragma unroll
        for (int gi = 1; gi < GN_MAX_GROUP; ++gi)
            if (gi < ga.n && lo >= cum[gi - 1]) { cg = gi; base = cum[gi - 1]; }
        const int nt = ntiles_of(cg);
        crb = (lo - base) / nt;
        cj = (lo - base) - crb * nt;
    }
    int tile_no = lo;                               // flat id of the CURRENT tile (cg, crb, cj)
    int ng, nrb, nj, nnt;                           // the tile after it
    const uint4* nw;
    auto successor = [&]() {
        ng = cg; nrb = crb; nj = cj + 1;
        if (nj == nnt) {                            // nnt: column tiles of problem cg on entry
            nj = 0;
            ++nrb;
#pragma unroll
            for (int gi = 0; gi < GN_MAX_GROUP; ++gi)
                if (gi == cg && tile_no + 1 >= cum[gi]) { ng = cg + 1; nrb = 0; }
            if (ng != cg) { nnt = ntiles_of(ng); nw = weight_of(ng); }
        }
    };
    nnt = ntiles_of(cg);
    nw = weight_of(cg);

    // ---- weight image of column tile j of the planes `w` -> LDS buffer `buf` by LDS-DMA: 32 wave-instructions of 1 KiB,
    // 8 per wave (every wave fills ITS 8 KiB slice of the image)
    auto dma_B = [&](const uint4* w, int j, int buf) {
        const uint4* src = w + (size_t)j * (RS_KS * 2 * 64) + wave * (PPW * 64) + lane;
        unsigned char* dst = smem + buf * RS_BT + wave * (PPW * 1024);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(src + i * 64, (lds_void*)(dst + i * 1024), 16, 0, 0);
    };

    // ---- this wave's 32 rows: two fp16 planes of x * 2^-e_row, e_row from the row's own amax.
    // Rows are read COALESCED (an instruction = 8 rows x 128 B = 8 full cache lines: lane -> row 8 i + lane / 8, 16 B at
    // column 4 (lane % 8) of a 32-column piece) and reach the MFMA operand layout (lane -> row lane % 32, 8 k-values)
    // through a 32 x 32 staging piece in this wave's own slice of the image buffer that is free until the next LDS-DMA.
    // (The first version loaded operand-shaped: 32 rows x 32 B per instruction = 32 half-used lines; the texture
    //  addresser made that 14-17 us per row block, 30 us of a 230 us launch.)
    f16x8 ah[RS_KS], al[RS_KS];
    int er[4] = {0, 0, 0, 0};                       // exponents of the rows 8 q + lane / 8 (the rows this lane finishes in the epilogue)
    bool sc_ok = false;                             // 2^(er + ewt) is a normal float for every row of the wave: the finish stage is one fma per value
    int cur_rb = -1, cur_g = -1;
    auto load_A = [&](int rb, int freebuf) __attribute__((always_inline)) {
        float* st = NW == 8 ? reinterpret_cast<float*>(smem + 2 * RS_BT + NW * RS_PATCH * 4 + wave * RS_STG)
                            : reinterpret_cast<float*>(smem + freebuf * RS_BT + wave * (8 * 1024));
        int* ex = reinterpret_cast<int*>(st + 32 * RS_AP);
        const float* rp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = rb * ROWS + wave * 32 + 8 * i + (lane >> 3);
            rp[i] = p.A + (size_t)phys_row(p, gm < p.M ? gm : 0) * p.lda + 4 * (lane & 7);   // rows past M: row 0 (never stored)
        }
        // pass 1: row amax (one batch of 32 independent loads: the planes' registers are free at this point)
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        {
            float4 raw[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) raw[q] = ld4(rp[q & 3] + 32 * (q >> 2));
            RS_SB();
#pragma unroll
            for (int q = 0; q < 32; ++q)
                m[q & 3] = fmaxf(m[q & 3], fmaxf(fmaxf(fabsf(raw[q].x), fabsf(raw[q].y)), fmaxf(fabsf(raw[q].z), fabsf(raw[q].w))));
            RS_SB();
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = m[i];
            v = fmaxf(v, __shfl_xor(v, 1, 64));
            v = fmaxf(v, __shfl_xor(v, 2, 64));
            v = fmaxf(v, __shfl_xor(v, 4, 64));
            int need = (int)((__float_as_uint(v) >> 23) & 0xffu) - 126 - 15;        // |x| < 2^(need + 15)
            need = need < -120 ? -120 : (need > 112 ? 112 : need);                   // (an Inf in the row: the row goes non-finite, alone)
            er[i] = need;
            const int e = need + ewt;
            ok = ok && e >= -126 && e <= 127;
            if ((lane & 7) == 0) ex[8 * i + (lane >> 3)] = need;
        }
        sc_ok = __builtin_amdgcn_readfirstlane(__ballot(ok) == ~0ull ? 1 : 0) != 0;
        __builtin_amdgcn_wave_barrier();
        const float scale = __uint_as_float((unsigned)(127 - ex[lane & 31]) << 23);  // 2^-e of the row this lane feeds to the MFMAs
        // pass 2: the same rows again (L2 hits), piece by piece through the staging patch
        int zero = 0;
        asm volatile("" : "+v"(zero));              // real loads, not 128 live registers
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float4 raw[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) raw[q] = ld4(rp[q & 3] + zero + 32 * (4 * c + (q >> 2)));
            RS_SB();
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) st4(st + (8 * i + (lane >> 3)) * RS_AP + 4 * (lane & 7), raw[4 * pc + i]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float* fp = st + (lane & 31) * RS_AP + 16 * ks + 8 * (lane >> 5);
                    f16x4 h0, l0, h1, l1;
                    split4_f16(ld4(fp), scale, h0, l0);
                    split4_f16(ld4(fp + 4), scale, h1, l1);
                    ah[2 * (4 * c + pc) + ks] = f16x8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    al[2 * (4 * c + pc) + ks] = f16x8{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                }
                __builtin_amdgcn_wave_barrier();
            }
            RS_SB();
        }
    };

    // ---- epilogue of one 32 x 32 tile in six stages: {patch write, patch read (+ operand loads), finish + store} for the
    // two half tiles of 16 rows (lane holds column lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile; after the
    // patch 8 lanes own one row: 128 B).  Tiles alternate between two accumulator / bias register sets (`par`): the
    // finished tile sits in set par while the next one accumulates into set par ^ 1, with no copy between them (a
    // loop-carried copy of a LOADED register -- the bias -- is a use with an LDS-DMA in flight: hipcc drains vmcnt there).
    // `lean`: the form that rides between the MFMAs of the next tile: full row block, identity row map, no operand
    // loads -- ~90 instructions per tile (the general form: 340, as long as the MFMAs).
    f32x16 accs[2];
    float4 biasv[2] = {zero4(), zero4()};
    int e_rb = 0, e_j = 0;                          // coordinates of the pending tile
    float4 erv[2], egv[2];
    f32x4_nt ev[2];
    size_t eoff[2] = {0, 0};
    bool eok[2] = {false, false};
    lds_f32* const patch = (lds_f32*)((lds_u8*)smem + 2 * RS_BT) + wave * RS_PATCH;
    lds_f32* const patch_w = patch + 4 * (lane >> 5) * 32 + (lane & 31);        // this lane's first patch element
    const lds_f32* const patch_r = patch + (lane >> 3) * 32 + (lane & 7) * 4;   // its row segment of the first 8 rows
    auto stage = [&](auto PARC, int k, bool lean) __attribute__((always_inline)) {
        constexpr int par = decltype(PARC)::value;
        const int half = k / 3, what = k % 3;
        const int gn = e_j * 32 + (lane & 7) * 4;
        if (what == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int R = half * 8 + r;
                patch_w[((R & 3) + 8 * ((R >> 2) & 1)) * 32] = accs[par][R];
            }
        } else if (what == 1) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (lean) {
                    ev[it] = *(lds_cf4*)(patch_r + 8 * it * 32);
                } else {
                    const int gm = e_rb * ROWS + wave * 32 + 16 * half + 8 * it + (lane >> 3);
                    eok[it] = gm < p.M;
                    eoff[it] = (size_t)phys_row(p, (eok[it] && !(abl & 64)) ? gm : (abl & 64 ? (gm & 127) : 0)) * p.ldc + gn;   // (probe 64: every store into the first 128 rows: L2-resident)
                    // unconditional loads (a missing operand reads the output's own address, rows past M read row 0):
                    erv[it] = ld4((p.res ? p.res : p.C) + eoff[it]);      // see the note in the finish stage
                    egv[it] = ld4((p.gate ? p.gate : p.C) + eoff[it]);
                    ev[it] = *(lds_cf4*)(patch_r + 8 * it * 32);
                }
            }
        } else {
            const bool act = gn >= p.act_lo && gn < p.act_hi;      // act ranges are multiples of 4
            if (lean) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int gm = e_rb * ROWS + wave * 32 + 16 * half + 8 * it + (lane >> 3);
                    const size_t off = (size_t)(gm * p.row_gstride + p.row_goff) * p.ldc + gn;     // identity row map
                    float4 o;
                    if (sc_ok) {
                        const float s_ = __uint_as_float((unsigned)(er[2 * half + it] + ewt + 127) << 23);
                        o = make_float4(fmaf(ev[it].x, s_, biasv[par].x), fmaf(ev[it].y, s_, biasv[par].y),
                                        fmaf(ev[it].z, s_, biasv[par].z), fmaf(ev[it].w, s_, biasv[par].w));
                    } else {
                        const int e = er[2 * half + it] + ewt;
                        o = make_float4(ldexpf(ev[it].x, e) + biasv[par].x, ldexpf(ev[it].y, e) + biasv[par].y,
                                        ldexpf(ev[it].z, e) + biasv[par].z, ldexpf(ev[it].w, e) + biasv[par].w);
                    }
                    if (p.pre_out) st4(p.pre_out + off, o);
                    if (act) o = act4(o, GN_ACT_SILU);
                    if (p.nt_store) st4_nt(p.C + off, o); else st4(p.C + off, o);
                }
            } else {
                // hipcc's wait insertion is path-insensitive: an operand load issued under one condition and consumed under
                // another counts as "possibly still in flight" ever after, and the first later write to its register (an MFMA
                // of the next tile, with the LDS-DMA of the tile after it in flight) then drains vmcnt.  So the operand loads
                // are unconditional and are consumed HERE on every path.
                asm volatile("" :: "v"(erv[0].x), "v"(erv[1].x), "v"(egv[0].x), "v"(egv[1].x));
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (!eok[it]) continue;
                    const int e = er[2 * half + it] + ewt;       // back from the row / weight exponents (only the RESULT may
                    float4 o = make_float4(ldexpf(ev[it].x, e) + biasv[par].x, ldexpf(ev[it].y, e) + biasv[par].y,   // under- / overflow)
                                           ldexpf(ev[it].z, e) + biasv[par].z, ldexpf(ev[it].w, e) + biasv[par].w);
                    if (p.pre_out) st4(p.pre_out + eoff[it], o);
                    if (act) o = act4(o, GN_ACT_SILU);
                    if (p.gate) o = o * (p.gate_mode ? dact4(egv[it], GN_ACT_SILU) : egv[it]);
                    if (p.res) o = erv[it] + o;
                    if (p.nt_store) st4_nt(p.C + eoff[it], o); else st4(p.C + eoff[it], o);
                }
            }
        }
    };
    auto epilogue_now = [&](auto PARC) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            stage(PARC, k, false);
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---- main loop.  Per tile t:  wait + barrier | [flush the pending epilogue, new rows] | bias of tile t |
    // LDS-DMA of image t + 1 | 48 MFMAs on image t with the epilogue stages of tile t - 1 between them.
    // vmcnt retires in order, so the wait at the top is COUNTED when the only operations younger than this wave's
    // LDS-DMA are the `n_st` stores of an overlapped epilogue of a full row block: the stores stay in flight.
    bool pending = false;                           // the other register set holds a finished tile that is not stored yet
    int n_st = 0;                                   // stores issued behind the last LDS-DMA (0: unknown -> drain)
    auto body = [&](auto PARC) __attribute__((always_inline)) {
        constexpr int par = decltype(PARC)::value;  // register set AND image buffer of the current tile
        using Other = std::integral_constant<int, par ^ 1>;
        // this wave's part of the current image has landed (the BUILTIN wait: hipcc then knows that no LDS-DMA is
        // pending and counts the loads below exactly; behind an asm wait it drained vmcnt after every pair of them)
#if GN_RS_PROBE
#define RS_STAMP(k)                                                                                                      \
    do {                                                                                                                 \
        if ((abl & 256) && blockIdx.x < 64 && (wave == 0 || wave == 3) && lane == 0 && tile_no - lo >= 8 && tile_no - lo < 16) \
            rs_phase_buf[((blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + (tile_no - lo - 8)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define RS_STAMP(k) do {} while (0)
#endif
        RS_STAMP(0);
        if (n_st == 4) __builtin_amdgcn_s_waitcnt(0x0f74);     // vmcnt(4)
        else if (n_st == 8) __builtin_amdgcn_s_waitcnt(0x0f78);    // vmcnt(8)
        else __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
        RS_STAMP(1);
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();         // ... everyone's has; and everyone has left the previous image
        RS_STAMP(2);
        const bool change = cg != cur_g || crb != cur_rb;
        // the pending tile can ride in this tile's MFMA phase when nothing it needs changes and it has no operand loads
        const bool ride = pending && !change && !p.res && !p.gate && p.row_cnt == 1 && (e_rb + 1) * ROWS <= p.M && !(abl & 96);
        if (pending && !ride) {
            if (!(abl & 4)) epilogue_now(Other{});
            pending = false;
        }
        if (change) {
            if (cg != cur_g) { select(cg); cur_g = cg; }
            if (!((abl & 8) && tile_no > lo)) load_A(crb, par ^ 1);
            cur_rb = crb;
        }
        // (order of the memory operations: bias load, the eight LDS-DMA pieces, then -- inside the MFMA phase -- the
        //  stores of the riding epilogue: the only operations younger than the LDS-DMA, which the counted wait at the
        //  top of the next tile leaves in flight.  Issuing the DMA pieces between the MFMA chains instead was measured:
        //  520 instead of 1100 cycles in front of the phase, 4490 instead of 3690 inside it -- no gain, not kept.)
        biasv[par] = p.bias ? ld4(p.bias + cj * 32 + (lane & 7) * 4) : zero4();
        successor();
        if (tile_no + 1 < hi && !(abl & 1)) dma_B(nw, nj, par ^ 1);
        n_st = ride ? (p.pre_out ? 8 : 4) : 0;
        RS_STAMP(3);

        // ---- 48 MFMAs: x = hi + lo per operand: lo*hi, hi*lo, hi*hi per k-step.  The three MFMAs of a k-step are one
        // dependent chain issued back to back (an MFMA that takes the previous one's D whole as its C needs no wait
        // states -- but ANY instruction between two such MFMAs costs ~43 cycles: measured 42 % of the wave time
        // issue-stalled with one chain and fillers everywhere); even and odd k-steps use different accumulators, and
        // everything else -- the two fragment reads of the NEXT k-step, one epilogue stage of the previous tile every
        // other k-step -- sits between the chains, where the accumulator changes.  Order pinned with sched_barrier(0).
        const lds_u8* bb = (const lds_u8*)smem + par * RS_BT + lane * 16;
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
        f16x8 fh[2], fl[2];
        fh[0] = *(lds_cf16x8*)(bb);
        fl[0] = *(lds_cf16x8*)(bb + 1024);
        if (!(abl & 2))
#pragma unroll
        for (int s = 0; s < RS_KS; ++s) {
            const int cur = s & 1;
            if (s + 1 < RS_KS) {
                fh[cur ^ 1] = *(lds_cf16x8*)(bb + (2 * s + 2) * 1024);
                fl[cur ^ 1] = *(lds_cf16x8*)(bb + (2 * s + 3) * 1024);
            }
            if (s == 2) __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): image t + 1 and the bias have landed (see the note at the top)
            if (s >= 2 && s <= 12 && !(s & 1) && ride && !(abl & 4)) stage(Other{}, s / 2 - 1, true);
            RS_SB();
            if (cur == 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], fh[cur], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fl[cur], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fh[cur], acc, 0, 0, 0);
            } else {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], fh[cur], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fl[cur], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fh[cur], acc1, 0, 0, 0);
            }
            RS_SB();
        }
        RS_STAMP(4);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        accs[par] = acc;
        e_rb = crb;
        e_j = cj;
        pending = true;
        cg = ng; crb = nrb; cj = nj;                // on to the next tile
        ++tile_no;
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    dma_B(nw, cj, 0);
    while (tile_no < hi) {
        body(P0{});
        if (tile_no < hi) body(P1{});
    }
    if (pending && !(abl & 4)) {
        if ((hi - 1 - lo) & 1) epilogue_now(P1{}); else epilogue_now(P0{});
    }
#if GN_RS_PROBE
    if ((abl & 128) && tid == 0 && blockIdx.x < 1024) {
        rs_trace_buf[blockIdx.x * 4] = t_start;
        rs_trace_buf[blockIdx.x * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        rs_trace_buf[blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        rs_trace_buf[blockIdx.x * 4 + 3] = hi - lo;
    }
#endif
}

}  // namespace gn

// K = 256, no prologue / K segments, N a multiple of 32: what the row-stationary kernel takes (2 x fp16 arithmetic only)
static int rs_waves() {
    static const int nw = getenv("GN_RS_NW") && atoi(getenv("GN_RS_NW")) == 8 ? 8 : 4;
    return nw;
}
bool gn_gemm_rs_eligible(const gn::GemmArgs* g, int n) {
    static const bool off = getenv("GN_GEMM_RS") && atoi(getenv("GN_GEMM_RS")) == 0;      // A/B switch: the LDS-slab kernel
    if (off) return false;
    // small groups stay on the LDS-slab kernel: a workgroup splits its rows once per ~tiles / 512 column tiles
    // (GN_GEMM_RS_MIN_TILES overrides the measured switch-over for a sweep)
    static const long min_tiles = getenv("GN_GEMM_RS_MIN_TILES") ? atol(getenv("GN_GEMM_RS_MIN_TILES")) : 0;
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (g[i].K != 16 * gn::RS_KS || (g[i].N & 31) || g[i].pro_mode || g[i].a_gate || g[i].a_seg ||
            g[i].act_kind != GN_ACT_SILU)
            return false;
        tiles += (long)((g[i].M + 127) / 128) * (g[i].N / 32);
    }
    return tiles >= min_tiles;
}

int gn_gemm_rs_launch(const gn::GemmArgs* g, int n, hipStream_t st, double nt_min_bytes) {
    gn::GroupArgs ga;
    const int rows = 32 * rs_waves();
    long end = 0;
    for (int i = 0; i < gn::GN_MAX_GROUP; ++i) {
        ga.g[i] = g[i < n ? i : n - 1];
        ga.g[i].nt_store = (double)ga.g[i].M * ga.g[i].N * 4.0 >= nt_min_bytes;
        if (i < n) end += (long)((g[i].M + rows - 1) / rows) * (g[i].N / 32);
        ga.tile_end[i] = (int)end;
    }
    if (end > 0x7fffffffL) return GN_ERR_BAD_ARG;
    ga.n = n;
    static const int abl = getenv("GN_RS_ABL") ? atoi(getenv("GN_RS_ABL")) : 0;
    ga.spread = abl;
    if (end == 0) return GN_OK;
    static const long max_grid = getenv("GN_RS_GRID") ? atol(getenv("GN_RS_GRID")) : (rs_waves() == 8 ? 256 : 512);
    long grid = end < max_grid ? end : max_grid;
    if (grid >= 8) grid &= ~7L;
    if (rs_waves() == 8) hipLaunchKernelGGL(gn::gemm_rs_f16x2<8>, dim3((unsigned)grid), dim3(512), 0, st, ga);
    else hipLaunchKernelGGL(gn::gemm_rs_f16x2<4>, dim3((unsigned)grid), dim3(256), 0, st, ga);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

#if GN_RS_PROBE
// probe builds only (not part of the C ABI in include/): the traces of the last launch
extern "C" int gn_debug_rs_phases(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gn::rs_phase_buf), sizeof(gn::rs_phase_buf));
}
extern "C" int gn_debug_rs_trace(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gn::rs_trace_buf), sizeof(gn::rs_trace_buf));
}
#endif
