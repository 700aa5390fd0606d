// gn_gemm.hip -- fp32 dense projections on the CDNA4 matrix cores.
//
//   C[r, n] = epi( sum_k pro(A)[r, k] * W[n, k] + bias[n] )        W is nn.Linear's [out, in]
//
// Replaces every Dense/MLP on the path (reference layers.py:457-581; call sites
// gotennet.py:400-407, 432-441, 611, 728, 738) and, with transposed weights, their
// input-gradients in the force backward.  Exact fp32: v_mfma_f32_32x32x2_f32 is
// bitwise an fmaf chain, 157.3 TFLOP/s peak on MI355X (no TF32/xf32 on gfx950).
//
// Tiling: a 256-thread workgroup (4 waves as 2x2) owns a (64 TM) x (64 TN) output
// tile; each wave a (32 TM) x (32 TN) patch of 32x32 MFMA tiles.  TM = TN = 2 for
// the edge-sized products, TM = TN = 1 when the problem has too few 128x128 tiles to
// fill 256 CUs (atom-sized products: a wave's chain of 64-cycle MFMAs is the latency).
// (The plane arithmetics below use 1 x 4 waves instead: 128 x 128 big tiles, and in the f16x2
// arithmetic 32 x 128 small tiles when every N >= 128; small f16x2 groups leave this file
// altogether for the K-resident panel kernel of gn_gemm_panel.hip.)
// K in slabs of 32.  A and W slabs are staged through LDS as [rows][36] floats: the
// 36-float pitch keeps ds_read_b128 conflict-free for its 16-lane service groups
// (r*36 mod 64 hits 16 distinct 4-bank slots) and keeps 16-byte alignment.  Within a
// slab the K order is permuted so lanes 0-31 own k in [0,16) and lanes 32-63 own
// k in [16,32): each lane fetches its MFMA operands as contiguous float4s.  The next
// slab is prefetched into registers while the current one is multiplied.
//
// Prologue on A (applied while staging; columns [pro_lo, pro_hi) only):
//   1: A <- SiLU(A)                 (activations are stored pre-activation, consumers apply SiLU)
//   2: A <- A * SiLU'(P)            (backward through an activation; P = stored pre-activation)
//   and, for every column, A <- A * G when a_gate != NULL.
#include <type_traits>
#include "gn_gemm.h"
#include "gn_tune.h"

#ifndef GN_SPLIT_LOOP
#define GN_SPLIT_LOOP 2        // 2: branch-free A fetch two slabs ahead (exact s_waitcnt); 0: conditional loads
#endif
#ifndef GN_SPLIT_MINW
#define GN_SPLIT_MINW 2        // minimum waves per SIMD of the split kernel (register cap 512 / n)
#endif
// (The timing-probe paths of rounds 2-3 -- term / load / barrier ablations that produce wrong results, the no-store
//  build, per-tile cycle stamps, start skew -- are gone from the product translation unit; what they measured is
//  recorded in DESIGN.md 5.3 and profiles/r0[23]_*.)

namespace gn {

// SPLIT = true: the 3 x bf16-split arithmetic (gn_gemm_split.hip has the numerics): A is split into hi/mid/lo bf16
// planes while it is staged into LDS ([3][BM][40] bf16 per slab, double-buffered); the weight arrives pre-split in
// FRAGMENT-MAJOR order (gn_split_bf16x3: [n-tile of 32][k-step of 16][plane][lane][8 bf16], so one wave-wide 16-byte
// load is one contiguous KiB = exactly one MFMA B operand) and goes L2 -> registers, never through LDS: the weights
// are a few MB, L2-resident, and every wave needs a different column block.  Six v_mfma_f32_32x32x16_bf16 per
// 32x32x16 product block, fp32 accumulate; everything else (grouping, persistent tile walk, prologues, epilogue) is
// shared with the exact-fp32 instantiation.
// Wave grid: the 4 waves of a workgroup form WM x WN; each wave owns TM x TN MFMA tiles of 32 x 32, so the workgroup
// tile is (32 TM WM) x (32 TN WN).  Exact fp32: 2 x 2 waves.  SPLIT, 128 x 128 tile: 1 x 4 waves of 4 x 1 tiles -- every
// wave then needs ONE 32-column block of the weight per k-step (3 KiB from L2 through the 64 B/clk L1 path instead of
// 6 KiB with 2 x 2 tiles per wave) and re-reads the whole A slab from LDS, which has the bandwidth to spare.
// MODE: 0 exact fp32 MFMA, 1 three bf16 planes (six MFMA terms), 2 two scaled fp16 planes (three MFMA terms, gn_gemm.h)
template <int TM, int TN, int WM, int WN, bool PRO, int PF, int MODE, bool ASILU>
__device__ __forceinline__ void gemm_body(const GroupArgs ga) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr bool SPLIT = MODE != 0;
    constexpr bool F16 = MODE == 2;
    constexpr int NP = F16 ? 2 : 3;                 // operand planes
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int RA = BM / 32, RB = BN / 32;       // staged float4 rows per thread
    constexpr int STAGE = (BM + BN) * PITCH;        // floats per K-slab buffer (A rows then W rows)
    constexpr int CP = BN + 4;                      // epilogue tile pitch
    constexpr int APL = BM * SPLIT_PB;              // SPLIT: bf16 elements per A plane of a slab
    constexpr int STAGE_S = NP * APL;               // SPLIT: 16-bit elements per slab buffer
    constexpr int MAIN_FLOATS = SPLIT ? (2 * STAGE_S) / 2 : 2 * STAGE;
    constexpr int LDS_FLOATS = (MAIN_FLOATS > BM * CP) ? MAIN_FLOATS : BM * CP;
    // F16: + the block exponents of the two slab buffers, [2][4] signed bytes: wave q stages rows 8q..8q+7 of every
    // 32-row M-tile, and those rows -- accumulator registers 4q..4q+3 of every MFMA tile -- share ONE exponent per slab.
    // One dword per slab buffer, so "did anything change" is a single scalar compare per slab.
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS + (F16 ? 2 : 0)];
    signed char* const exps = reinterpret_cast<signed char*>(smem + LDS_FLOATS);
    int e_run = -120;                               // F16, staging side: running exponent of this wave's rows
    unsigned e_acc = 0x88888888u;                   // F16, MFMA side: the four block exponents (bytes) the accumulators are held in
    int ewt = 0;                                    // F16: the weight tensor's exponent (header of the packed planes)

    // One launch walks the tiles of up to GN_MAX_GROUP independent problems (a "group": the atom-sized products of
    // a layer are too small to fill 256 CUs one at a time).  Inside a problem tiles are row-tile major, so consecutive
    // ids share their A rows.
    // XCD-aware order (block b runs on XCD b % 8, speed only): EVERY problem's tile list is cut into 8 contiguous
    // ranges, XCD x walks range x of problem 0, then range x of problem 1, ... -- an A row tile is pulled through ONE
    // L2 instead of all eight, and a problem with longer tiles (larger K) is spread over all XCDs.
    // grid = 8 * ceil(tiles / 8) capped; the excess exits.  Walk index j in [0, tiles_here).
    const int xcd = blockIdx.x & 7;
    int cum[GN_MAX_GROUP], shift[GN_MAX_GROUP];      // cum: walk indices below cum[g] belong to problems <= g;
    if (ga.spread) {                                 // shift: walk index j -> problem-local tile id j + shift[g]
        int run = 0, prev_end = 0;
#pragma unroll
        for (int gi = 0; gi < GN_MAX_GROUP; ++gi) {
            const int tg = gi < ga.n ? ga.tile_end[gi] - prev_end : 0;
            prev_end = gi < ga.n ? ga.tile_end[gi] : prev_end;
            const int cq = tg >> 3, cr = tg & 7;
            const int base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
            shift[gi] = base - run;
            run += cq + (xcd < cr ? 1 : 0);
            cum[gi] = run;
        }
    } else {
        // problems with equal tile lengths: ONE cut of the concatenated tile list (each XCD then touches one or two
        // weight matrices only: measured 46 vs 67 us on the x / v pair)
        int tiles_all = ga.tile_end[0];
#pragma unroll
        for (int gi = 1; gi < GN_MAX_GROUP; ++gi)
            if (gi < ga.n) tiles_all = ga.tile_end[gi];
        const int xq = tiles_all >> 3, xr = tiles_all & 7;
        const int lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
        const int hi = lo + xq + (xcd < xr ? 1 : 0);
        int prev_end = 0;
#pragma unroll
        for (int gi = 0; gi < GN_MAX_GROUP; ++gi) {
            const int ge = gi < ga.n ? ga.tile_end[gi] : prev_end;
            const int a1 = ge < hi ? ge : hi;        // this XCD's part of problem gi ends at global id a1
            shift[gi] = lo - prev_end;               // walk index j is global id lo + j, problem-local id lo + j - prev_end
            cum[gi] = (a1 > lo ? a1 : lo) - lo;
            prev_end = ge;
        }
    }
    const int tile_stop = cum[GN_MAX_GROUP - 1];
    // the problem a walk index belongs to (indices only grow, so the scan never goes back)
    GemmArgs p = ga.g[0];
    int g_begin = -shift[0], g_end = cum[0], tiles_n = (p.N + BN - 1) / BN;
    auto select = [&](int t) {
#pragma unroll
        for (int gi = 1; gi < GN_MAX_GROUP; ++gi)
            if (t >= cum[gi - 1] && g_end <= cum[gi - 1]) {
                p = ga.g[gi];
                g_begin = -shift[gi];
                g_end = cum[gi];
            }
        tiles_n = (p.N + BN - 1) / BN;
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int c4 = tid & 7;                         // staging map: thread -> (row sr + 32 i, float4 column c4)
    const int sr = tid >> 3;
    const int stride = gridDim.x >> 3;

    // fetch-side context of a tile (set one tile AHEAD at the end of the K loop: the first slab of the next
    // tile is already in flight while the current tile's epilogue runs)
    int prow[RA];
    const float* brow[RB];
    bool aok[RA], bok[RB];
    bool a_full = false;                            // every row of the tile is inside M and K is a multiple of the slab depth: no masks
    auto set_tile = [&](int t_idx) {
        const int m0f = ((t_idx - g_begin) / tiles_n) * BM, n0f = ((t_idx - g_begin) % tiles_n) * BN;
        a_full = m0f + BM <= p.M && (p.K % BK) == 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int gm = m0f + sr + 32 * i;
            aok[i] = gm < p.M;
            prow[i] = phys_row(p, aok[i] ? gm : 0);
        }
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int gn = n0f + sr + 32 * i;
                bok[i] = gn < p.N;
                brow[i] = p.W + (size_t)(bok[i] ? gn : 0) * p.K + 4 * c4;
            }
        }
    };

    f32x16 acc[TM][TN];

    // PF register sets of prefetched slabs: a fetched slab stays in flight for PF compute phases (PF = 1 is
    // what ships: PF = 2/3 measured no gain on MI355X and costs a wave of occupancy)
    float4 pa[PF][RA], pb[PF][RB];
    // L2 (split, no prologue): branch-free A fetch into TWO register sets (slab s lives in set s & 1), so that the
    // compiler can count its loads exactly (exec-masked loads force conservative s_waitcnt: the HBM latency of the A
    // slab was exposed once per slab) and a slab stays in flight for two slab times.
    constexpr bool L2 = SPLIT && !PRO && GN_SPLIT_LOOP == 2;
    float4 qa2[2][RA];
    bool kok2[2] = {true, true};
    auto fetchA = [&](int k0, float4 (&qa)[RA], bool& kflag) {
        const int kraw = k0 + 4 * c4;
        const bool kok = kraw < p.K;
        const int kc = kok ? kraw : 0;               // clamped: any valid column, zeroed by select in stashA
        const int k0c = kok ? k0 : 0;
        const float* Ab = p.A;
        int ka = kc;
        if (p.a_seg) {
            const bool s2 = k0c >= 2 * p.a_seg, s1 = k0c >= p.a_seg;
            Ab = s2 ? p.A3 : (s1 ? p.A2 : p.A);
            ka = kc - (s2 ? 2 * p.a_seg : (s1 ? p.a_seg : 0));
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) qa[i] = ld4(Ab + (size_t)prow[i] * p.lda + ka);   // prow of a row past M is row 0: valid memory
        kflag = kok;
    };
    // F16: scale this wave's 8-row block of every M-tile by 2^-e (e = running maximum of the block's binary exponent
    // over the slabs staged so far, so that |x'| < 2^15), split x' = hi + lo into two fp16 planes, publish e
    auto stash_f16 = [&](int sb, const float4 (&v)[RA]) {
        _Float16* d0 = reinterpret_cast<_Float16*>(smem) + sb * STAGE_S + sr * SPLIT_PB + 4 * c4;
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        int need = (int)((wave_umax_sgpr(__float_as_uint(m)) >> 23) & 0xffu) - 126 - 15;          // |x| < 2^(need + 15)
        if (__builtin_expect(need > 112, 0)) {       // an Inf (a NaN never wins v_max) in the block: scale by its FINITE
            asm volatile("" ::: "memory");           // values, so that only the rows that hold the Inf turn non-finite
            float mf = 0.f;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const float c[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) mf = fmaxf(mf, fabsf(c[t]) <= 3.0e38f ? fabsf(c[t]) : 0.f);
            }
            need = (int)((wave_umax_sgpr(__float_as_uint(mf)) >> 23) & 0xffu) - 126 - 15;
        }
        need = need < -120 ? -120 : need;            // (a signed byte; blocks below 2^-105 keep fewer bits)
        e_run = need > e_run ? need : e_run;
        const float scale = __uint_as_float((unsigned)(127 - e_run) << 23);        // 2^-e_run (e_run in [-120, 113])
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            f16x4 h, l;
            split4_f16(v[i], scale, h, l);
            _Float16* d = d0 + 32 * i * SPLIT_PB;
            *reinterpret_cast<f16x4*>(d) = h;
            *reinterpret_cast<f16x4*>(d + APL) = l;
        }
        if (lane == 0) exps[sb * 4 + wave] = (signed char)e_run;
    };
    auto stashA = [&](int sb, const float4 (&qa)[RA], bool kflag) {
        if constexpr (F16) {
            if (a_full) {                            // interior tile (wave-uniform): 16 selects and their compares less per slab
                stash_f16(sb, qa);
                return;
            }
            float4 v[RA];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = aok[i] && kflag;
                v[i] = make_float4(ok ? qa[i].x : 0.f, ok ? qa[i].y : 0.f, ok ? qa[i].z : 0.f, ok ? qa[i].w : 0.f);
            }
            stash_f16(sb, v);
            return;
        }
        __bf16* d0 = reinterpret_cast<__bf16*>(smem) + sb * STAGE_S + sr * SPLIT_PB + 4 * c4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bf16x4 h, m, l;
            const bool ok = aok[i] && kflag;
            const float4 v = make_float4(ok ? qa[i].x : 0.f, ok ? qa[i].y : 0.f, ok ? qa[i].z : 0.f, ok ? qa[i].w : 0.f);
            split4_trunc(v, h, m, l);
            __bf16* d = d0 + 32 * i * SPLIT_PB;
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + APL) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * APL) = l;
        }
    };
    auto fetch = [&](int k0, float4 (&qa)[RA], float4 (&qb)[RB]) {
        if constexpr (L2) {                           // (the cross-tile prefetch of a tile's first slab: set 0)
            fetchA(k0, qa2[0], kok2[0]);
            return;
        }
        const int kc = k0 + 4 * c4;
        const bool kok = kc < p.K;
        const bool pro = PRO && p.pro_mode && kc >= p.pro_lo && kc < p.pro_hi;
        // K-segmented A (a slab never straddles a segment: a_seg % BK == 0, checked by the launcher)
        const float* Ab = p.A;
        int ka = kc;
        if (p.a_seg) {
            if (k0 >= 2 * p.a_seg) { Ab = p.A3; ka = kc - 2 * p.a_seg; }
            else if (k0 >= p.a_seg) { Ab = p.A2; ka = kc - p.a_seg; }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float4 v = zero4();
            if (aok[i] && kok) {
                v = ld4(Ab + (size_t)prow[i] * p.lda + ka);
                if constexpr (PRO) {
                    if (pro) v = (p.pro_mode == 1) ? act4(v, ASILU ? (int)GN_ACT_SILU : p.act_kind) : v * dact4(ld4(p.a_pre + (size_t)prow[i] * p.ldp + kc), ASILU ? (int)GN_ACT_SILU : p.act_kind);
                    if (p.a_gate) v = v * ld4(p.a_gate + (size_t)prow[i] * p.ldg + kc);
                }
            }
            qa[i] = v;
        }
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < RB; ++i) qb[i] = (bok[i] && kok) ? ld4(brow[i] + k0) : zero4();
        }
    };
    // slab buffer `sb` (0 / 1) of the double buffer
    auto stash = [&](int sb, const float4 (&qa)[RA], const float4 (&qb)[RB]) {
        if constexpr (L2) {
            stashA(sb, qa2[0], kok2[0]);
        } else if constexpr (F16) {
            stash_f16(sb, qa);
        } else if constexpr (SPLIT) {
            __bf16* d0 = reinterpret_cast<__bf16*>(smem) + sb * STAGE_S + sr * SPLIT_PB + 4 * c4;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                bf16x4 h, m, l;
                split4_trunc(qa[i], h, m, l);
                __bf16* d = d0 + 32 * i * SPLIT_PB;
                *reinterpret_cast<bf16x4*>(d) = h;
                *reinterpret_cast<bf16x4*>(d + APL) = m;
                *reinterpret_cast<bf16x4*>(d + 2 * APL) = l;
            }
        } else {
            float* buf = smem + sb * STAGE;
#pragma unroll
            for (int i = 0; i < RA; ++i) st4(&buf[(sr + 32 * i) * PITCH + 4 * c4], qa[i]);
#pragma unroll
            for (int i = 0; i < RB; ++i) st4(&buf[(BM + sr + 32 * i) * PITCH + 4 * c4], qb[i]);
        }
    };
    // SPLIT: B operands of k-step g (16 deep) for this wave's TN column blocks, NP planes each, L2 -> registers
    // (F16: the packed planes start with a 256-byte header whose first int is the weight tensor's exponent)
    const uint4* wfrag = reinterpret_cast<const uint4*>(p.W);
    int ks2 = 0;                                    // k-steps per column block in the fragment-major weight (even)
    size_t nt_off[TN];
    auto set_btile = [&](int n0b) {
        wfrag = reinterpret_cast<const uint4*>(p.W) + (F16 ? 16 : 0);
        if constexpr (F16) ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));
        ks2 = 2 * ((p.K + BK - 1) / BK);
        const int nt_last = (p.N + 31) / 32 - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int nt = n0b / 32 + wn * TN + j;
            nt = nt < nt_last ? nt : nt_last;        // column blocks past N: any valid block (results are never stored)
            nt_off[j] = (size_t)nt * ks2 * (NP * 64) + lane;
        }
    };
    auto load_b = [&](int g, uint4 (&q)[TN][NP]) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s_ = 0; s_ < NP; ++s_) q[j][s_] = wfrag[nt_off[j] + (size_t)(g * NP + s_) * 64];
    };
    // F16: bring the accumulator rows of every 8-row block to the exponent slab buffer `sb` was staged with.  The
    // exponents only grow and settle after the first slabs: the common case is TM scalar compares that all fall through.
    auto rescale = [&](int sb) {
        const unsigned en = (unsigned)__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(exps + sb * 4));
        if (__builtin_expect(en != e_acc, 0)) {
            asm volatile("" ::: "memory");              // a real branch: never if-convert the multiplies
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int eo = (int)(signed char)(e_acc >> (8 * q)), e1 = (int)(signed char)(en >> (8 * q));
                const float f = ldexpf(1.0f, eo - e1);                  // exponents only grow: f <= 1, exact
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] *= f;
            }
            e_acc = en;
        }
    };

    // double-buffered LDS K loop, one barrier per slab; register set s holds slab kt+1 when slab kt is
    // multiplied and is refilled with slab kt+1+PF right after it has been written to LDS
    int idx = blockIdx.x >> 3;
    if (idx >= tile_stop) return;
    select(idx);
    set_tile(idx);
    fetch(0, pa[0], pb[0]);
  // persistent over tiles: workgroup b walks the tiles base + b/8, base + b/8 + gridDim/8, ... of ITS XCD's range
  for (;;) {
    const int nk = (p.K + BK - 1) / BK;
    const int m0 = ((idx - g_begin) / tiles_n) * BM;
    const int n0 = ((idx - g_begin) % tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (F16) {
        e_run = -120;
        e_acc = 0x88888888u;                         // -120 in every byte (the accumulators are zero)
    }
    stash(0, pa[0], pb[0]);
    uint4 bq[2][TN][NP];
    if constexpr (SPLIT) {
        set_btile(n0);
        load_b(0, bq[0]);
    }
    __syncthreads();
    const int khalf = (lane >> 5) * 16;
    const int frow = lane & 31;
    if constexpr (SPLIT) {
        static_assert(PF == 1, "the split main loop prefetches one slab");
        // One slab = two 16-deep k-steps.  While slab kt is multiplied: the weights of the NEXT k-step are in flight
        // (L2 -> registers), slab kt + 1 (already in registers) is split and written to the other LDS buffer, and after
        // the barrier slab kt + 2 goes in flight from HBM.  The last slab is peeled so that the steady-state body has
        // no conditional code: it is ONE scheduling region and the split arithmetic can sit in the MFMAs' shadow.
        if (!L2 && nk > 1) fetch(BK, pa[0], pb[0]);
        auto kstep = [&](const __bf16* Ap, int ks, const uint4 (&bw)[TN][NP]) {
            if constexpr (F16) {
                // x = hi + lo per operand: lo*hi, hi*lo, hi*hi (lo*lo is below 2^-22 of the product)
                f16x8 a[TM][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_)
                        a[i][s_] = *reinterpret_cast<const f16x8*>(Ap + s_ * APL + i * 32 * SPLIT_PB + ks * 16);
                constexpr int TA[3] = {1, 0, 0};
                constexpr int TB[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                a[i][TA[t]], __builtin_bit_cast(f16x8, bw[j][TB[t]]), acc[i][j], 0, 0, 0);
                return;
            } else {
            bf16x8 a[TM][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_)
                    a[i][s_] = *reinterpret_cast<const bf16x8*>(Ap + s_ * APL + i * 32 * SPLIT_PB + ks * 16);
            // smallest terms first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi); consecutive MFMAs rotate over the
            // TM*TN accumulators so that none waits on the one before it
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            a[i][TA[t]], __builtin_bit_cast(bf16x8, bw[j][TB[t]]), acc[i][j], 0, 0, 0);
            }
        };
        const __bf16* Abase = reinterpret_cast<const __bf16*>(smem) + (wm * 32 * TM + frow) * SPLIT_PB + (lane >> 5) * 8;
        if constexpr (L2) {
            // Load order per slab kt (vmcnt retires in order): weights of k-step 2kt+1, THEN the A slab kt+2 (two slab
            // times ahead), mid-slab the weights of k-step 2kt+2: nothing that is needed soon ever queues behind an A
            // load younger than one slab time.  sched_barrier(0) keeps the phases apart (without them the scheduler
            // hoists across the whole region until the 256-register budget spills: measured 15-30 % slower).
            auto slab = [&](int kt, auto SET, auto LAST) {
                constexpr int set = decltype(SET)::value;            // register set of slab kt (= kt & 1)
                constexpr bool last = decltype(LAST)::value;
                const __bf16* Ap = Abase + set * STAGE_S;
                load_b(2 * kt + 1, bq[1]);
                if constexpr (!last) fetchA((kt + 2) * BK, qa2[set], kok2[set]);   // slab kt + 2 -> the set slab kt came from
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) rescale(set);
                kstep(Ap, 0, bq[0]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!last) load_b(2 * kt + 2, bq[0]);
                kstep(Ap, 1, bq[1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!last) stashA(set ^ 1, qa2[set ^ 1], kok2[set ^ 1]);  // slab kt + 1 -> the other LDS buffer
                __syncthreads();
            };
            using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
            using No = std::false_type; using Yes = std::true_type;
            fetchA(BK, qa2[1], kok2[1]);             // slab 1 (zero-filled past K); slab 0 is already staged
            int kt = 0;
            for (; kt + 2 < nk; kt += 2) {
                slab(kt, T0{}, No{});
                slab(kt + 1, T1{}, No{});
            }
            if (nk - kt == 2) {
                slab(kt, T0{}, No{});
                slab(kt + 1, T1{}, Yes{});
            } else {
                slab(kt, T0{}, Yes{});
            }
        } else {
        // (the conditional form: weights, MFMAs and the split each in their own basic block.  A branch-free single
        // block, with or without sched_group_barrier interleaving, measured 15-30 % SLOWER on MI355X: the scheduler
        // hoists until the 256-register budget spills)
        for (int kt = 0; kt < nk; ++kt) {
            const __bf16* Ap = Abase + (kt & 1) * STAGE_S;
            load_b(2 * kt + 1, bq[1]);
            if constexpr (F16) rescale(kt & 1);
            kstep(Ap, 0, bq[0]);
            if (kt + 1 < nk) load_b(2 * kt + 2, bq[0]);
            kstep(Ap, 1, bq[1]);
            if (kt + 1 < nk) stash((kt + 1) & 1, pa[0], pb[0]);
            __syncthreads();
            if (kt + 2 < nk) fetch((kt + 2) * BK, pa[0], pb[0]);
        }
        }
    } else {
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (1 + s < nk) fetch((1 + s) * BK, pa[s], pb[s]);

    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int kt = kt0 + s;
        if (kt >= nk) break;
        const float* As = smem + (kt & 1) * STAGE;
        const float* Bs = As + BM * PITCH;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float* ap = &As[(wm * 32 * TM + i * 32 + frow) * PITCH + khalf + hh * 8];
                const float4 a0 = ld4(ap), a1 = ld4(ap + 4);
                a[i][0] = a0.x; a[i][1] = a0.y; a[i][2] = a0.z; a[i][3] = a0.w;
                a[i][4] = a1.x; a[i][5] = a1.y; a[i][6] = a1.z; a[i][7] = a1.w;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* bp = &Bs[(wn * 32 * TN + j * 32 + frow) * PITCH + khalf + hh * 8];
                const float4 b0 = ld4(bp), b1 = ld4(bp + 4);
                b[j][0] = b0.x; b[j][1] = b0.y; b[j][2] = b0.z; b[j][3] = b0.w;
                b[j][4] = b1.x; b[j][5] = b1.y; b[j][6] = b1.z; b[j][7] = b1.w;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stash((kt + 1) & 1, pa[s], pb[s]);   // other buffer: last read in iteration kt-1
        __syncthreads();
        if (kt + 1 + PF < nk) fetch((kt + 1 + PF) * BK, pa[s], pb[s]);
      }
    }
    }

    // next tile's first slab goes in flight now and lands during the epilogue (same problem only: the epilogue
    // below still needs this problem's arguments)
    const int next = idx + stride;
    const bool has_next = next < tile_stop;
    const bool same = has_next && next < g_end;
    if (same) {
        set_tile(next);
        fetch(0, pa[0], pb[0]);
    }

    // (An epilogue straight from the accumulators -- a lane owns one column of each 32 x 32 tile, so a store instruction
    // would write two full 128-byte row segments with no LDS round trip -- measured 2x SLOWER per tile: 64 dword stores
    // per lane instead of 16 dwordx4; tools/gemm_trace.py.)
    {
    // epilogue through LDS: accumulators -> [BM][BN+4] tile -> coalesced float4 rows
    // (lane holds column (lane & 31), rows (r&3) + 8 (r>>2) + 4 (lane>>5) of each 32x32 tile)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if constexpr (F16) v = ldexpf(v, (int)(signed char)(e_acc >> (8 * (r >> 2))) + ewt);   // back from the block / weight exponents (only the RESULT may under- / overflow)
                smem[row * CP + wn * 32 * TN + j * 32 + (lane & 31)] = v;
            }
    __syncthreads();
    {
        constexpr int C4 = BN / 4;                  // 256 % C4 == 0: a thread keeps ONE column group
        const int cc = (tid % C4) * 4, gn = n0 + cc;
        if (gn < p.N) {
            const float4 bias4 = p.bias ? ld4(p.bias + gn) : zero4();
            const bool act = gn >= p.act_lo && gn < p.act_hi;          // act ranges are multiples of 4
            if (p.res || p.gate) {
                // four rows per pass, all residual / gate loads of the pass issued BEFORE its first store: the
                // stores may alias them as far as the compiler knows (res == C is allowed), and a load behind
                // a store was one exposed round trip per row
                constexpr int ITS = (BM * C4) / 256, UN = ITS < 4 ? ITS : 4;
                for (int it0 = 0; it0 < ITS; it0 += UN) {
                    size_t off[UN];
                    bool ok[UN];
                    float4 rv[UN], gv[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int gm = m0 + (it0 + u) * (256 / C4) + tid / C4;
                        ok[u] = gm < p.M;
                        off[u] = (size_t)phys_row(p, ok[u] ? gm : 0) * p.ldc + gn;
                        rv[u] = (ok[u] && p.res) ? ld4(p.res + off[u]) : zero4();
                        gv[u] = (ok[u] && p.gate) ? ld4(p.gate + off[u]) : zero4();
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        if (!ok[u]) continue;
                        const int row = (it0 + u) * (256 / C4) + tid / C4;
                        float4 v = ld4(&smem[row * CP + cc]) + bias4;
                        if (p.pre_out) st4(p.pre_out + off[u], v);
                        if (act) v = act4(v, ASILU ? (int)GN_ACT_SILU : p.act_kind);
                        if (p.gate) v = v * (p.gate_mode ? dact4(gv[u], ASILU ? (int)GN_ACT_SILU : p.act_kind) : gv[u]);
                        if (p.res) v = rv[u] + v;
                        st4(p.C + off[u], v);
                    }
                }
            } else {
#pragma unroll 4
                for (int it = 0; it < (BM * C4) / 256; ++it) {
                    const int row = it * (256 / C4) + tid / C4;
                    const int gm = m0 + row;
                    if (gm >= p.M) continue;
                    float4 v = ld4(&smem[row * CP + cc]) + bias4;
                    const size_t off = (size_t)phys_row(p, gm) * p.ldc + gn;
                    if (p.pre_out) st4(p.pre_out + off, v);
                    if (act) v = act4(v, ASILU ? (int)GN_ACT_SILU : p.act_kind);
                    if (gn >= p.nt_store) st4_nt(p.C + off, v); else st4(p.C + off, v);
                }
            }
        }
    }
    __syncthreads();                                // LDS is reused by the next tile's first slab
    }
    if (!has_next) break;
    if (!same) {                                    // first tile of the next problem
        select(next);
        set_tile(next);
        fetch(0, pa[0], pb[0]);
    }
    idx = next;
  }
}

// ASILU: every problem of the launch uses SiLU (the reference's default): the activation is a compile-time constant.
// With the kind as a run-time switch the eleven other activations' code cost the 128x128 split kernel a 36-byte spill
// and the 64x64 fp32 kernel a wave of occupancy; models with another activation take the !ASILU instantiations.
template <int TM, int TN, int WM, int WN, bool PRO, int PF, bool ASILU>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GroupArgs ga) {
    gemm_body<TM, TN, WM, WN, PRO, PF, 0, ASILU>(ga);
}

// 3 x bf16-split instantiation: two workgroups per CU (one wave of each per SIMD: while one is in its epilogue or at a
// barrier the other feeds the matrix pipe), so the register budget is capped at 256 per lane.
template <int TM, int TN, int WM, int WN, bool PRO, bool ASILU>
__global__ __launch_bounds__(256, GN_SPLIT_MINW) void gemm_bf16x3_mfma(const GroupArgs ga) {
    gemm_body<TM, TN, WM, WN, PRO, 1, 1, ASILU>(ga);
}

// 2 x fp16-split instantiation (three MFMA terms per product, block exponents): same grid and tile shapes
template <int TM, int TN, int WM, int WN, bool PRO, bool ASILU>
__global__ __launch_bounds__(256, GN_SPLIT_MINW) void gemm_f16x2_mfma(const GroupArgs ga) {
    gemm_body<TM, TN, WM, WN, PRO, 1, 2, ASILU>(ga);
}

}  // namespace gn

extern "C" int gn_gemm_ex(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                          int Mrows, int Nout, int K, int act_lo, int act_hi,
                          int row_cnt, int row_gstride, int row_goff,
                          const float* res, const float* gate, int gate_mode, float* pre_out,
                          int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                          const float* a_gate, int ldg, int act_kind, void* stream) {
    if (act_kind < 0 || act_kind >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (Mrows < 0 || Nout <= 0 || K <= 0 || (K & 3) || (lda & 3) || (Nout & 3) || (ldc & 3) || (act_lo & 3) ||
        (act_hi & 3) || row_cnt <= 0)
        return GN_ERR_BAD_ARG;
    if (gate != nullptr && res == nullptr && gate_mode == 0) return GN_ERR_BAD_ARG;
    if (pro_mode < 0 || pro_mode > 2 || (pro_mode == 2 && (!a_pre || (ldp & 3))) || (a_gate && (ldg & 3)) ||
        (pro_mode && ((pro_lo & 3) || (pro_hi & 3))))
        return GN_ERR_BAD_ARG;
    if (Mrows == 0) return GN_OK;
    gn::GemmArgs p{A, W, bias, C, res, gate, pre_out, a_pre, a_gate, lda, ldc, ldp, ldg, Mrows, Nout, K,
                   act_lo, act_hi, pro_mode, pro_lo, pro_hi, row_cnt, row_gstride, row_goff, gate_mode,
                   nullptr, nullptr, 0, 0, act_kind};
    return gn_gemm_launch(&p, 1, (hipStream_t)stream, 0);
}

static int gemm_args_ok(int Mrows, int Nout, int K, int lda, int ldc, int act_lo, int act_hi, int row_cnt,
                        const float* res, const float* gate, int gate_mode, int pro_mode, int pro_lo, int pro_hi,
                        const float* a_pre, int ldp, const float* a_gate, int ldg) {
    if (Mrows < 0 || Nout <= 0 || K <= 0 || (K & 3) || (lda & 3) || (Nout & 3) || (ldc & 3) || (act_lo & 3) ||
        (act_hi & 3) || row_cnt <= 0)
        return 0;
    if (gate != nullptr && res == nullptr && gate_mode == 0) return 0;
    if (pro_mode < 0 || pro_mode > 2 || (pro_mode == 2 && (!a_pre || (ldp & 3))) || (a_gate && (ldg & 3)) ||
        (pro_mode && ((pro_lo & 3) || (pro_hi & 3))))
        return 0;
    return 1;
}

static int gemm_group_impl(const gn_gemm_desc* d, int n, void* stream, int split) {
    if (n < 0 || n > gn::GN_MAX_GROUP || (n > 0 && !d)) return GN_ERR_BAD_ARG;
    gn::GemmArgs g[gn::GN_MAX_GROUP];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const gn_gemm_desc& q = d[i];
        if (!gemm_args_ok(q.M, q.N, q.K, q.lda, q.ldc, q.act_lo, q.act_hi, q.row_cnt, q.res, q.gate, q.gate_mode,
                          q.pro_mode, q.pro_lo, q.pro_hi, q.a_pre, q.ldp, q.a_gate, q.ldg))
            return GN_ERR_BAD_ARG;
        if (q.a_seg < 0 || (q.a_seg && ((q.a_seg % gn::BK) || q.pro_mode || q.a_gate || q.K > 3 * q.a_seg ||
                                        !q.A2 || (q.K > 2 * q.a_seg && !q.A3))))
            return GN_ERR_BAD_ARG;
        if (q.M == 0) continue;
        g[m++] = gn::GemmArgs{q.A, q.W, q.bias, q.C, q.res, q.gate, q.pre_out, q.a_pre, q.a_gate, q.lda, q.ldc, q.ldp,
                              q.ldg, q.M, q.N, q.K, q.act_lo, q.act_hi, q.pro_mode, q.pro_lo, q.pro_hi, q.row_cnt,
                              q.row_gstride, q.row_goff, q.gate_mode, q.A2, q.A3, q.a_seg};
        if (q.act_kind < 0 || q.act_kind >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
        g[m - 1].act_kind = q.act_kind;
    }
    if (m == 0) return GN_OK;
    return gn_gemm_launch(g, m, (hipStream_t)stream, split);
}

extern "C" int gn_gemm_group(const gn_gemm_desc* d, int n, void* stream) { return gemm_group_impl(d, n, stream, 0); }
// the same group on the 3 x bf16-split path: every desc.W points to the planes written by gn_split_bf16x3
extern "C" int gn_gemm_group_split(const gn_gemm_desc* d, int n, void* stream) { return gemm_group_impl(d, n, stream, 1); }
// ... and on the 2 x fp16-split path: every desc.W points to the buffer written by gn_split_f16x2
extern "C" int gn_gemm_group_f16x2(const gn_gemm_desc* d, int n, void* stream) { return gemm_group_impl(d, n, stream, 2); }

// one launch for n <= GN_MAX_GROUP problems (validated by the callers)
int gn_gemm_launch(const gn::GemmArgs* g, int n, hipStream_t st, int split) {
#ifndef GN_SPLIT_GRID
#define GN_SPLIT_GRID 14
#endif
#if GN_SPLIT_GRID == 12            // 64 x 128 tile, waves 1 x 4 of 2 x 1 MFMA tiles: ~130 registers, three workgroups per CU
    const int BMB = split ? 64 : 128, BNB = 128;
    const long cap_big = split ? 768 : 512;
#else
    const int BMB = 128, BNB = 128;
    const long cap_big = GN_GEMM_BIG_CAP;
#endif
    if (split == 2) {
        const int r = gn_gemm_panel_launch(g, n, st);
        if (r != 0) return r > 0 ? GN_OK : -r;
    }
    gn::GroupArgs ga;
    long big = 0;
    bool pro = false;
    for (int i = 0; i < n; ++i) {
        big += (long)((g[i].M + 127) / 128) * ((g[i].N + 127) / 128);
        pro = pro || g[i].pro_mode != 0 || g[i].a_gate != nullptr;
    }
    // 128x128 tiles from GN_GEMM_BIG_MIN = 900 tiles up (measured switch-over; a build-time constant of gn_tune.h, swept
    // with tools/variants.py): the [E x 256 x K] products (850 tiles = 1.66 rounds on 512 resident workgroups) run
    // 4-20 % faster as 3400 64x64 tiles, the [E x 1536 x 256] product (5100 tiles) and the four-problem X group (1008)
    // want the big tile
    const bool use_big = big >= (long)GN_GEMM_BIG_MIN;
    // the f16x2 small tile is 32 x 128 (waves 1 x 4: a row is split once per 128 columns) unless a problem is narrower than a tile
    bool wide = GN_F16_SMALL_WIDE && split == 2;
    for (int i = 0; i < n; ++i) wide = wide && g[i].N >= 128;
    // f16x2, small-tile groups with many rows: 64 x 128 tiles (waves 1 x 4 of 2 x 1 MFMA tiles) -- a weight fragment serves two
    // row tiles, half the L2 -> CU weight traffic of the 32 x 128 tile per output row (GN_F16_MID_ROWS, gn_tune.h; 0: never)
    // Measured (round 6, MI355X): the gated residual launch [54368 x 256 x 256]g + rider 75 -> 68.8 us in the step (stand-alone 69.3
    // -> 65.7, ungated 42.9 -> 37.5); the ungated launches of the step tie or lose 2 us (their riders have K = 512), so only
    // groups with a gated residual epilogue take it
    long rows_all = 0;
    bool gated = false;
    for (int i = 0; i < n; ++i) {
        rows_all += g[i].M;
        gated = gated || (g[i].gate != nullptr && g[i].res != nullptr);
    }
    const bool mid = wide && !use_big && gated && GN_F16_MID_ROWS > 0 && rows_all >= (long)GN_F16_MID_ROWS;
    long end = 0;
    // outputs of 100 MB and more (the [E, (1+M)F] edge projection) are stored non-temporally: they are consumed by
    // later kernels from HBM anyway and would only evict the node tables (K6 +5 %); GN_GEMM_NT_MB (gn_tune.h)
    const double nt_min = (double)GN_GEMM_NT_MB * 1048576.0;
    for (int i = 0; i < gn::GN_MAX_GROUP; ++i) {
        ga.g[i] = g[i < n ? i : n - 1];
        // (nt_store = first column written non-temporally.  The first K columns of a large output stay on the normal
        //  path: for the edge projection [W_re | W_rs] that is the attention block, which the segment softmax re-reads
        //  right away -- measured 23.3 -> 20.6 us for it at C2, the message kernel unchanged; GN_GEMM_NT_LO >= 0 (gn_tune.h)
        //  fixes the column instead)
        const int nt_lo = GN_GEMM_NT_LO >= 0 ? GN_GEMM_NT_LO : ga.g[i].K;
        ga.g[i].nt_store = (double)ga.g[i].M * ga.g[i].N * 4.0 >= nt_min ? (ga.g[i].N > nt_lo ? nt_lo : 0) : 0x7fffffff;
        if (i < n) end += use_big ? (long)((g[i].M + BMB - 1) / BMB) * ((g[i].N + BNB - 1) / BNB)
                                  : (mid ? (long)((g[i].M + 63) / 64) * ((g[i].N + 127) / 128)
                                  : wide ? (long)((g[i].M + 31) / 32) * ((g[i].N + 127) / 128)
                                                                     : (long)((g[i].M + 63) / 64) * ((g[i].N + 63) / 64));
        ga.tile_end[i] = (int)end;
    }
    ga.n = n;
    if (split == 2) {                                // large prologue-free K = 256 groups with a wide product: the column-loop kernel
        const int r = gn_gemm_colpipe_launch(ga.g, n, st);
        if (r != 0) return r > 0 ? GN_OK : -r;
    }
    // per-problem XCD ranges when the problems are unlike (different tile lengths K, or a small problem riding with
    // a large one: its tiles would otherwise all sit at the end of the last XCD's range)
    ga.spread = 0;
    for (int i = 1; i < n; ++i) {
        const long t0 = ga.tile_end[0], ti = ga.tile_end[i] - ga.tile_end[i - 1];
        ga.spread |= (g[i].K != g[0].K) || ti * 4 < t0 || t0 * 4 < ti;
    }
    if (end == 0) return GN_OK;
    // persistent launch: at most 2 (big tiles) / 4 (small tiles) workgroups per CU walk the tile list (+2 %)
    long grid = 8L * ((end + 7) / 8);
    const long cap = use_big ? cap_big : (mid ? (long)GN_F16_MID_CAP : (wide ? (long)GN_F16_SMALL_CAP : 1024));
    if (grid > cap) grid = cap;
    bool silu = true;
    for (int i = 0; i < n; ++i) silu = silu && g[i].act_kind == GN_ACT_SILU;
#define GN_GEMM_GO_S(TM_, TN_, WM_, WN_, PRO_)                                                                        \
    do {                                                                                                              \
        if (silu) hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<TM_, TN_, WM_, WN_, PRO_, true>), dim3((unsigned)grid), dim3(256), 0, st, ga); \
        else hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<TM_, TN_, WM_, WN_, PRO_, false>), dim3((unsigned)grid), dim3(256), 0, st, ga);     \
    } while (0)
#define GN_GEMM_GO_H(TM_, TN_, WM_, WN_, PRO_)                                                                        \
    do {                                                                                                              \
        if (silu) hipLaunchKernelGGL((gn::gemm_f16x2_mfma<TM_, TN_, WM_, WN_, PRO_, true>), dim3((unsigned)grid), dim3(256), 0, st, ga); \
        else hipLaunchKernelGGL((gn::gemm_f16x2_mfma<TM_, TN_, WM_, WN_, PRO_, false>), dim3((unsigned)grid), dim3(256), 0, st, ga);     \
    } while (0)
#define GN_GEMM_GO_F(TM_, TN_, WM_, WN_, PRO_)                                                                        \
    do {                                                                                                              \
        if (silu) hipLaunchKernelGGL((gn::gemm_f32_mfma<TM_, TN_, WM_, WN_, PRO_, 1, true>), dim3((unsigned)grid), dim3(256), 0, st, ga); \
        else hipLaunchKernelGGL((gn::gemm_f32_mfma<TM_, TN_, WM_, WN_, PRO_, 1, false>), dim3((unsigned)grid), dim3(256), 0, st, ga);     \
    } while (0)
#if GN_SPLIT_GRID == 14
#define GN_SPLIT_BIG(PRO_) GN_GEMM_GO_S(4, 1, 1, 4, PRO_)
#elif GN_SPLIT_GRID == 12
#define GN_SPLIT_BIG(PRO_) GN_GEMM_GO_S(2, 1, 1, 4, PRO_)
#else
#define GN_SPLIT_BIG(PRO_) GN_GEMM_GO_S(2, 2, 2, 2, PRO_)
#endif
    if (split == 2) {
        if (use_big) { if (pro) GN_GEMM_GO_H(4, 1, 1, 4, true); else GN_GEMM_GO_H(4, 1, 1, 4, false); }
        else if (mid) { if (pro) GN_GEMM_GO_H(2, 1, 1, 4, true); else GN_GEMM_GO_H(2, 1, 1, 4, false); }
        else if (wide) { if (pro) GN_GEMM_GO_H(1, 1, 1, 4, true); else GN_GEMM_GO_H(1, 1, 1, 4, false); }
        else { if (pro) GN_GEMM_GO_H(1, 1, 2, 2, true); else GN_GEMM_GO_H(1, 1, 2, 2, false); }
    } else if (split) {
        if (use_big) { if (pro) GN_SPLIT_BIG(true); else GN_SPLIT_BIG(false); }
        else { if (pro) GN_GEMM_GO_S(1, 1, 2, 2, true); else GN_GEMM_GO_S(1, 1, 2, 2, false); }
    } else {
        if (use_big) { if (pro) GN_GEMM_GO_F(2, 2, 2, 2, true); else GN_GEMM_GO_F(2, 2, 2, 2, false); }
        else { if (pro) GN_GEMM_GO_F(1, 1, 2, 2, true); else GN_GEMM_GO_F(1, 1, 2, 2, false); }
    }
#undef GN_SPLIT_BIG
#undef GN_GEMM_GO_S
#undef GN_GEMM_GO_H
#undef GN_GEMM_GO_F
    GN_LAUNCH_CHECK();
    return GN_OK;
}

static int gemm_planes_single(const float* A, int lda, const unsigned short* W3, const float* bias, float* C, int ldc,
                              int Mrows, int Nout, int K, int act_lo, int act_hi,
                              int row_cnt, int row_gstride, int row_goff,
                              const float* res, const float* gate, int gate_mode, float* pre_out,
                              int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                              const float* a_gate, int ldg, int act_kind, void* stream, int mode) {
    if (!gemm_args_ok(Mrows, Nout, K, lda, ldc, act_lo, act_hi, row_cnt, res, gate, gate_mode, pro_mode, pro_lo, pro_hi,
                      a_pre, ldp, a_gate, ldg) || !W3 || act_kind < 0 || act_kind >= GN_ACT_COUNT)
        return GN_ERR_BAD_ARG;
    if (Mrows == 0) return GN_OK;
    gn::GemmArgs p{A, reinterpret_cast<const float*>(W3), bias, C, res, gate, pre_out, a_pre, a_gate, lda, ldc, ldp, ldg,
                   Mrows, Nout, K, act_lo, act_hi, pro_mode, pro_lo, pro_hi, row_cnt, row_gstride, row_goff, gate_mode,
                   nullptr, nullptr, 0, 0, act_kind};
    return gn_gemm_launch(&p, 1, (hipStream_t)stream, mode);
}

extern "C" int gn_gemm_split(const float* A, int lda, const unsigned short* W3, const float* bias, float* C, int ldc,
                             int Mrows, int Nout, int K, int act_lo, int act_hi,
                             int row_cnt, int row_gstride, int row_goff,
                             const float* res, const float* gate, int gate_mode, float* pre_out,
                             int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                             const float* a_gate, int ldg, int act_kind, void* stream) {
    return gemm_planes_single(A, lda, W3, bias, C, ldc, Mrows, Nout, K, act_lo, act_hi, row_cnt, row_gstride, row_goff, res,
                              gate, gate_mode, pre_out, pro_mode, pro_lo, pro_hi, a_pre, ldp, a_gate, ldg, act_kind, stream, 1);
}

extern "C" int gn_gemm_f16x2(const float* A, int lda, const unsigned short* W2, const float* bias, float* C, int ldc,
                             int Mrows, int Nout, int K, int act_lo, int act_hi,
                             int row_cnt, int row_gstride, int row_goff,
                             const float* res, const float* gate, int gate_mode, float* pre_out,
                             int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                             const float* a_gate, int ldg, int act_kind, void* stream) {
    return gemm_planes_single(A, lda, W2, bias, C, ldc, Mrows, Nout, K, act_lo, act_hi, row_cnt, row_gstride, row_goff, res,
                              gate, gate_mode, pre_out, pro_mode, pro_lo, pro_hi, a_pre, ldp, a_gate, ldg, act_kind, stream, 2);
}

extern "C" int gn_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                       int Mrows, int Nout, int K, int act_lo, int act_hi,
                       int row_cnt, int row_gstride, int row_goff,
                       const float* res, const float* gate, void* stream) {
    return gn_gemm_ex(A, lda, W, bias, C, ldc, Mrows, Nout, K, act_lo, act_hi, row_cnt, row_gstride, row_goff,
                      res, gate, 0, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, GN_ACT_SILU, stream);
}
