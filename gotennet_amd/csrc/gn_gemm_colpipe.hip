// gn_gemm_colpipe.hip -- the 2 x fp16-split projection for WIDE K = 256 products: an A row panel is converted ONCE and
// stays in LDS while the workgroup walks the output columns.
//
//   C[r, n] = epi( sum_k A[r, k] * W[n, k] + bias[n] ),   K = 256,   W is nn.Linear's [out, in]
//
// The edge projection [E x (1+M)F x F] (reference gotennet.py:406-407) and its riders.  The slab kernel of gn_gemm.hip
// re-fetches, re-scales and re-splits the A rows of a tile for each of the 12 (lmax 2) or 20 (lmax 4) column tiles of that
// product (round-4 counters: 7.5 VALU per MFMA).  Here a 256-thread workgroup (4 waves side by side) owns 64 rows:
//   * the [64 x 256] fp32 panel is requested once (a wave per row: 1 KiB requests), every ROW gets its own exponent e_r
//     (|x 2^-e_r| < 2^15: row-wise arithmetic, results do not depend on batch-mates), is split into the two fp16 planes and
//     parked in LDS, XOR-swizzled so that the 8-byte plane stores and the 16-byte MFMA fragment reads are bank-conflict
//     free with NO padding (64 KiB exactly);
//   * the workgroup then walks column passes of 128 columns over the resident panel (wave tile 64 x 32): fragment reads
//     (LDS), weight fragments (fragment-major planes of gn_split_f16x2, L2 -> registers, GN_CP_NB k-steps ahead), 6 MFMAs
//     per k-step and wave -- no barrier, no conversion;
//   * the product is issued TRANSPOSED (weights as the MFMA's first operand): a lane holds ONE output row (lane & 31) and
//     4 x 4 consecutive columns in its 16 accumulator registers, so the epilogue is a per-WAVE affair -- each wave
//     transposes its own 32 x 32 tiles through a private 4 KiB LDS scratch (ds_write_b128 -> ds_read_b128, swizzled, no
//     workgroup barrier) into rows of 128 contiguous bytes and applies 2^(e_r + e_w) / bias / activation range / residual /
//     gate / pre_out / row map as gn_gemm.hip does;
//   * two accumulator sets: the epilogue of pass p - 1 is cut into slices that ride between the MFMA groups of pass p
//     (k-step 0 / 8: tile 0 / 1 -> scratch; 2 / 10: rows back, residual / gate rows requested; 4, 6 / 12, 14: two rows
//     each finished and stored), so a wave's MFMA stream is continuous from the first pass of a row panel to its last and
//     its stores are spread over the pass instead of arriving in a burst (vector-memory operations retire in order: a
//     weight fragment requested behind sixteen stores waits for all of them);
//   * units (row panel, column pass) are dealt in contiguous, equal ranges to 512 persistent workgroups: equal MFMA work to
//     within one pass, ~1.6 conversions per workgroup; LDS 64 + 16 KiB = 80 KiB, two workgroups per CU.
// What it buys, and what it does not (profiles/r05_colloop_experiment.txt): [54368 x 1536 x 256] 205 -> 184 us, [54368 x
// 2560 x 256] 308 -> 268 us.  Not more, because the launch is bound by the chip's POWER budget, not by a pipe: the bare MFMA
// stream of this product (no loads, no LDS, no stores) takes 91 us on random data and 73 us on zeros, rocm-smi reads 1 319 W
// of the 1 400 W cap under it, and every ablation is additive in time (stores +40 us, conversion +18, operand traffic +15)
// whether or not it overlaps the MFMAs in the schedule -- the first form of this kernel (epilogue after each pass, s_memtime:
// 10 000 cycles of MFMAs then 12 700 of epilogue per wave and pass) and this one run the same 185-192 us.
#include <type_traits>
#include "gn_gemm.h"
#include "gn_tune.h"

#ifndef GN_CP_NB
#define GN_CP_NB 2              // weight fragments requested this many k-steps ahead (4: 22 spilled registers, the N = 256 products 45 -> 60 us)
#endif
#ifndef GN_CP_GRID
#define GN_CP_GRID 512          // persistent workgroups (two per CU)
#endif
#ifndef GN_CP_MIN_N
#define GN_CP_MIN_N 1024        // the widest product of the group has at least this many columns (8 column passes per conversion) ...
#endif
#ifndef GN_CP_MIN_TILES
#define GN_CP_MIN_TILES 2048    // ... and the group this many 64 x 128 output tiles (four per persistent workgroup); -1: never
#endif

namespace gn {

__global__ __launch_bounds__(256, 2) void gemm_f16x2_colpipe(const ClArgs ca) {
    constexpr int TM = 2, BM = 64, BN = 128, KC = CL_KC;
    constexpr int NS = KC / 16;                     // 16 k-steps
    constexpr int NB = GN_CP_NB;
    constexpr int RPW = BM / 4;                     // rows a wave converts
    constexpr int APL = BM * KC;                    // fp16 per plane
    static_assert(NS % NB == 0, "ring slots are compile-time");
    __shared__ __attribute__((aligned(16))) _Float16 panel[2 * APL];
    __shared__ __attribute__((aligned(16))) float scratch[4 * 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const scr = scratch + wave * 1024;       // this wave's transposition tile; its first 64 ints carry the row exponents across the conversion barrier
    const int lm = lane & 31, lh = lane >> 5;
    const int x0 = lh ^ (lm & 15);                  // fragment reads: row lm (+ 32 i), 16-byte chunk (2 ks + lh) ^ (lm & 15)
    const _Float16* const Arow = panel + lm * KC;
    const int c4 = lane & 7, rq = lane >> 3;        // read-back of a transposed tile: row 8 q + rq, columns 4 c4 .. 4 c4 + 3

    long T = ca.wend[0];
#pragma unroll
    for (int gi = 1; gi < GN_MAX_GROUP; ++gi)
        if (gi < ca.n) T = ca.wend[gi];
    const long w_lo = T * (long)blockIdx.x / (long)gridDim.x, w_hi = T * ((long)blockIdx.x + 1) / (long)gridDim.x;

    for (int gi = 0; gi < ca.n; ++gi) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const __attribute__((address_space(4))) ClArgs* KargPtr;
        const GemmArgs& p = ((KargPtr)__builtin_amdgcn_kernarg_segment_ptr())->g[gi];
        const long base = gi ? ((KargPtr)__builtin_amdgcn_kernarg_segment_ptr())->wend[gi - 1] : 0;
#else
        const GemmArgs& p = ca.g[gi];
        const long base = gi ? ca.wend[gi - 1] : 0;
#endif
        const int M = p.M, N = p.N;
        const int npass = (N + BN - 1) / BN;         // column passes per row panel
        const long nun = (long)((M + BM - 1) / BM) * npass;
        long k0l = w_lo - base, k1l = w_hi - base;
        k0l = k0l < 0 ? 0 : (k0l < nun ? k0l : nun);
        k1l = k1l < 0 ? 0 : (k1l < nun ? k1l : nun);
        int k = (int)k0l;
        const int k1 = (int)k1l;
        if (k >= k1) continue;
        const uint4* const wfrag = reinterpret_cast<const uint4*>(p.W) + 16 + lane;   // 256-byte header: the weight's exponent
        const int ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));
        const int nt_last = (N + 31) / 32 - 1;
        auto boff = [&](int ps) -> unsigned {        // weight fragments of this wave's column block in pass ps
            int nt = ps * 4 + wave;
            nt = nt < nt_last ? nt : nt_last;        // a column block past N: any valid block (never stored)
            return (unsigned)nt * (NS * 128u);
        };
        uint4 bq[NB][2];
        auto load_b = [&](unsigned bo, int g, uint4 (&q)[2]) __attribute__((always_inline)) {
            q[0] = wfrag[bo + (unsigned)g * 128u];
            q[1] = wfrag[bo + (unsigned)g * 128u + 64u];
        };
        const int kind = GN_ACT_SILU;
        const float* const res = p.res; const float* const gate = p.gate; float* const pre_out = p.pre_out; float* const C = p.C;
        const float* const bias = p.bias;
        const int ldc = p.ldc, gate_mode = p.gate_mode, nt_store = p.nt_store, act_lo = p.act_lo, act_hi = p.act_hi;

        while (k < k1) {                             // one row panel, column passes [ps0, ps1)
            const int m0 = (k / npass) * BM;
            const int ps0 = k % npass;
            const int ps1 = ps0 + (k1 - k) < npass ? ps0 + (k1 - k) : npass;
            k += ps1 - ps0;
            // physical row (row map) of panel row `lane`: ONE division per lane and panel
            const int prow_v = phys_row(p, m0 + lane < M ? m0 + lane : 0);
            // ---- the A rows: wave w converts rows 4 u + w, a full wave per row (1 KiB requests)
            {
                float4 va[RPW];
                const float* Ak = p.A + 4 * lane;
                const int lda = p.lda;
#pragma unroll
                for (int u = 0; u < RPW; ++u)
                    va[u] = ld4(Ak + (size_t)__builtin_amdgcn_readlane(prow_v, 4 * u + wave) * lda);   // (a row past M: row 0, zeroed below)
                const unsigned bo0 = boff(ps0);
#pragma unroll
                for (int g = 0; g < NB; ++g) load_b(bo0, g, bq[g]);
#pragma unroll
                for (int u = 0; u < RPW; ++u) {
                    const int r = 4 * u + wave;
                    const bool inside = m0 + r < M;
                    const float4 v = make_float4(inside ? va[u].x : 0.f, inside ? va[u].y : 0.f, inside ? va[u].z : 0.f, inside ? va[u].w : 0.f);
                    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                    int need = (int)((wave_umax_sgpr(__float_as_uint(m)) >> 23) & 0xffu) - 126 - 15;   // |x| < 2^(need + 15)
                    need = need < -120 ? -120 : need;
                    const float scale = __uint_as_float((unsigned)(127 - need) << 23);          // 2^-e (e in [-120, 114])
                    f16x4 h, l;
                    split4_f16(v, scale, h, l);
                    _Float16* d = panel + r * KC + (((lane >> 1) ^ (r & 15)) << 3) + ((lane & 1) << 2);
                    *reinterpret_cast<f16x4*>(d) = h;
                    *reinterpret_cast<f16x4*>(d + APL) = l;
                    if (lane < 4) reinterpret_cast<int*>(scratch + lane * 1024)[r] = need;
                }
            }
            lds_barrier();
            // exponent and physical row of the rows this lane STORES (rows 32 i + 8 q + rq)
            int e_r[TM][4], prow_e[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e_r[i][q] = reinterpret_cast<const int*>(scr)[i * 32 + 8 * q + rq] + ewt;
                    prow_e[i][q] = __shfl(prow_v, i * 32 + 8 * q + rq, 64);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the exponents are in registers before the first tile overwrites them

            f32x16 acc[2][TM];
            float4 t[4], rv[4], gv[4], bias4 = zero4();
            auto read_a = [&](int g, f16x8 (&a)[TM][2]) __attribute__((always_inline)) {
                const int ch8 = ((2 * g) ^ x0) * 8;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a[i][0] = *reinterpret_cast<const f16x8*>(Arow + i * 32 * KC + ch8);
                    a[i][1] = *reinterpret_cast<const f16x8*>(Arow + APL + i * 32 * KC + ch8);
                }
            };
            // slice `g` (0..15) of the epilogue of pass `ps` whose accumulators are set P
            auto epi_slice = [&](auto PP, int g, int ps) __attribute__((always_inline)) {
                constexpr int P = decltype(PP)::value;
                const int i = g >> 3, sub = g & 7;
                const int gn = ps * BN + wave * 32 + 4 * c4;
                const bool col_ok = gn < N;
                if (sub == 0) {
                    if (i == 0) bias4 = (bias && col_ok) ? ld4(bias + gn) : zero4();
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        st4(scr + lm * 32 + (((2 * gg + lh) ^ (lm & 7)) << 2),
                            make_float4(acc[P][i][4 * gg + 0], acc[P][i][4 * gg + 1], acc[P][i][4 * gg + 2], acc[P][i][4 * gg + 3]));
                    __builtin_amdgcn_wave_barrier();
                } else if (sub == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = 8 * q + rq;
                        t[q] = ld4(scr + rr * 32 + ((c4 ^ (rr & 7)) << 2));
                    }
                    if (res || gate) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool ok = m0 + i * 32 + 8 * q + rq < M && col_ok;
                            const size_t off = (size_t)prow_e[i][q] * ldc + gn;
                            rv[q] = (ok && res) ? ld4(res + off) : zero4();
                            gv[q] = (ok && gate) ? ld4(gate + off) : zero4();
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                } else if (sub == 4 || sub == 6) {
                    const bool act = gn >= act_lo && gn < act_hi;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = (sub == 4 ? 0 : 2) + qq;
                        const bool ok = m0 + i * 32 + 8 * q + rq < M && col_ok;
                        if (!ok) continue;
                        const size_t off = (size_t)prow_e[i][q] * ldc + gn;
                        const int e = e_r[i][q];
                        float4 v = make_float4(ldexpf(t[q].x, e), ldexpf(t[q].y, e), ldexpf(t[q].z, e), ldexpf(t[q].w, e)) + bias4;
                        if (pre_out) st4(pre_out + off, v);
                        if (act) v = act4(v, kind);
                        if (gate) v = v * (gate_mode ? dact4(gv[q], kind) : gv[q]);
                        if (res) v = rv[q] + v;
                        if (gn >= nt_store) st4_nt(C + off, v); else st4(C + off, v);
                    }
                }
            };
            // one column pass: 16 k-steps off the resident panel into accumulator set P; EPI: the slices of pass ps - 1 (set
            // P ^ 1) ride between the MFMA groups.  The weight ring continues into pass `ps + 1` when there is one.
            auto pass = [&](auto PP, auto EE, int ps, bool has_next) __attribute__((always_inline)) {
                constexpr int P = decltype(PP)::value;
                constexpr bool EPI = decltype(EE)::value;
                const unsigned bo = boff(ps), bo_next = boff(has_next ? ps + 1 : ps);
                f16x8 ab[2][TM][2];
                read_a(0, ab[0]);
#pragma unroll
                for (int g = 0; g < NS; ++g) {
                    if (g + 1 < NS) read_a(g + 1, ab[(g + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // x = hi + lo per operand: lo*hi, hi*lo, hi*hi (lo*lo is below 2^-22 of the product); weights first: D^T
                    constexpr int TA[3] = {1, 0, 0};
                    constexpr int TB[3] = {0, 1, 0};
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            if (g == 0 && tt == 0) {
                                f32x16 z;
#pragma unroll
                                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                                acc[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(f16x8, bq[g % NB][TB[tt]]), ab[g & 1][i][TA[tt]], z, 0, 0, 0);
                            } else {
                                acc[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(f16x8, bq[g % NB][TB[tt]]), ab[g & 1][i][TA[tt]], acc[P][i], 0, 0, 0);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + NB < NS) load_b(bo, g + NB, bq[g % NB]);
                    else if (has_next) load_b(bo_next, g + NB - NS, bq[g % NB]);
                    if constexpr (EPI) {
                        epi_slice(std::integral_constant<int, P ^ 1>{}, g, ps - 1);
                    }
                }
            };
            using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
            using No = std::false_type; using Yes = std::true_type;
            int ps = ps0;
            pass(P0{}, No{}, ps, ps + 1 < ps1);
            ++ps;
            while (ps + 1 < ps1) {
                pass(P1{}, Yes{}, ps, true);
                pass(P0{}, Yes{}, ps + 1, ps + 2 < ps1);
                ps += 2;
            }
            if (ps < ps1) {
                pass(P1{}, Yes{}, ps, false);
#pragma unroll
                for (int g = 0; g < NS; g += 2) epi_slice(P1{}, g, ps);
            } else {
#pragma unroll
                for (int g = 0; g < NS; g += 2) epi_slice(P0{}, g, ps - 1);
            }
            lds_barrier();                           // the next panel overwrites the planes and the exponents
        }
    }
}

}  // namespace gn

// Launch the column-loop kernel for a validated f16x2 group when it applies: 1 = launched, 0 = the caller keeps the panel /
// slab kernels, < 0 = -hipError_t.  Applies to: SiLU, no prologue, K = 256 for every problem, and a product at least
// GN_CP_MIN_N columns wide with at least GN_CP_MIN_TILES output tiles of 64 x 128 in the group -- where one conversion of a
// row panel serves many column passes.  Measured (MI355X, stand-alone / in the C2 step): [54368 x 1536 x 256] 205 -> 184 us,
// [54368 x 2560 x 256] 308 -> 268 us; a single-pass product (N = 256) ties the slab kernel and stays there.
int gn_gemm_colpipe_launch(const gn::GemmArgs* g, int n, hipStream_t st) {
    if (GN_CP_MIN_TILES < 0) return 0;
    gn::ClArgs ca;
    int n_max = 0;
    long end = 0;
    for (int i = 0; i < n; ++i) {
        if (g[i].pro_mode != 0 || g[i].a_gate != nullptr || g[i].K != gn::CL_KC || g[i].a_seg != 0 || g[i].act_kind != GN_ACT_SILU) return 0;
        n_max = g[i].N > n_max ? g[i].N : n_max;
    }
    for (int i = 0; i < gn::GN_MAX_GROUP; ++i) {
        ca.g[i] = g[i < n ? i : n - 1];
        if (i < n) end += (long)((g[i].M + 63) / 64) * ((g[i].N + 127) / 128);
        ca.wend[i] = end;
    }
    ca.n = n;
    if (n_max < GN_CP_MIN_N || end < (long)GN_CP_MIN_TILES) return 0;
    const unsigned grid = (unsigned)(end < (long)GN_CP_GRID ? end : (long)GN_CP_GRID);
    hipLaunchKernelGGL(gn::gemm_f16x2_colpipe, dim3(grid), dim3(256), 0, st, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    return 1;
}
