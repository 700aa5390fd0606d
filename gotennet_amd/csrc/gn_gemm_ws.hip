// gn_gemm_ws.hip -- 3 x bf16-split projections, wave-specialised persistent kernel (the large [E x N x K] products).
//
// Why: in the 4-wave split kernel (gn_gemm.hip, SPLIT) every wave alternates between 48 MFMAs and the work that feeds
// them (fetch the next A slab, split it into bf16 planes: ~100 VALU ops, write LDS, wait at the barrier) and then runs
// the whole epilogue.  Two such workgroups per CU overlap only by chance: tools/gemm_trace.py + the ablation builds
// (GN_SPLIT_ABL) put the matrix pipe at ~42 % busy on [54368 x 1536 x 256]; the split alone costs 18-25 % of the time,
// the epilogue another 21 %.  Here the roles are separate waves of ONE 512-thread workgroup per CU:
//
//   waves 0-3  consumers: each owns a 128 x 32 column strip of the 128 x 128 tile (4 accumulators of 32 x 32).  Per
//              32-deep slab: A fragments from LDS, weights L2 -> registers (fragment-major planes of gn_split_bf16x3,
//              one k-step ahead), 48 MFMAs, one barrier.  After the last slab the accumulators go to an LDS staging
//              tile and the wave starts the next tile at once.
//   waves 4-7  producers: fetch the A slab two slabs ahead (HBM -> registers, branch-free so that s_waitcnt counts are
//              exact), split + write the next slab's planes to the other LDS buffer, and -- spread over the slabs of
//              the NEXT tile -- drain the staged output tile: bias / SiLU / gate / residual, 16-byte coalesced stores.
//
// so the VALU-heavy split and the store-heavy epilogue run in the matrix pipe's shadow by construction.  LDS: 2 x 30 KiB
// A planes + 66 KiB staging tile = 126 KiB, one workgroup per CU.  Same arithmetic (term order, k order) as the 4-wave
// split kernel: results are bit-identical to it.  Used for groups whose problems all have K % 64 == 0, K >= 256, no A
// prologue; everything else stays on gn_gemm.hip.
#include <type_traits>
#include "gn_gemm.h"

#ifndef GN_SPLIT_TRACE
#define GN_SPLIT_TRACE 0
#endif
#ifndef GN_WS_ABL
#define GN_WS_ABL 0            // probe builds only (wrong results): 1 no split/stash, 2 no A fetch, 4 no plain drain, 8 no MFMAs
#endif
#ifndef GN_WS_PRIO
#define GN_WS_PRIO 2           // s_setprio of the producer waves (0-3)
#endif
#if GN_SPLIT_TRACE
__device__ long long gn_ws_trace_buf[64 * 16 * 8];
#define GN_WTR(slot)                                                                                   \
    do {                                                                                               \
        if ((tid & 255) == 0 && blockIdx.x < 64 && tile_no < 16)                                        \
            gn_ws_trace_buf[(blockIdx.x * 16 + tile_no) * 8 + (slot)] = __builtin_readcyclecounter();   \
    } while (0)
extern "C" int gn_debug_trace_ws(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gn_ws_trace_buf), sizeof(gn_ws_trace_buf));
}
#else
#define GN_WTR(slot) do {} while (0)
#endif

namespace gn {

constexpr int WS_BM = 128, WS_BN = 128, WS_CP = WS_BN + 4;
constexpr int WS_APL = WS_BM * SPLIT_PB;            // 16-bit elements per A plane of a slab
constexpr int WS_NBUF = 3;                          // A slab buffers in LDS (the producers run two slabs ahead)

// F16 = false: three bf16 planes, six MFMA terms.  F16 = true: two fp16 planes scaled by running block exponents, three
// terms (MODE 2 of gn_gemm.hip): producer wave q owns rows 8q..8q+7 of every 32-row MFMA tile and publishes their
// exponent per slab buffer as one byte; the consumers rescale accumulator registers 4q..4q+3 when it grows.  Same block
// structure, same term order as the 4-wave kernels: bit-identical results in both arithmetics.
template <bool F16>
__device__ __forceinline__ void gemm_ws_body(const GroupArgs ga) {
    constexpr int NP = F16 ? 2 : 3;
    constexpr int WS_STAGE = NP * WS_APL;           // 16-bit elements per slab buffer
    constexpr int WS_LDS_BYTES = WS_NBUF * WS_STAGE * 2 + WS_BM * WS_CP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[WS_LDS_BYTES + 16];
    __bf16* Abuf = reinterpret_cast<__bf16*>(smem_raw);
    float* Cst = reinterpret_cast<float*>(smem_raw + WS_NBUF * WS_STAGE * 2);
    signed char* exps = reinterpret_cast<signed char*>(smem_raw + WS_LDS_BYTES);       // F16: [WS_NBUF][4] block exponents

    // the XCD-aware tile walk of gn_gemm.hip (same order: an A row tile is pulled through one L2); scalars, no arrays
    // indexed at run time (those would live in scratch memory)
    const int xcd = blockIdx.x & 7;
    int cum0, cum1, cum2, cum3, sh0, sh1, sh2, sh3;
    {
        int cum[GN_MAX_GROUP], shift[GN_MAX_GROUP];
        if (ga.spread) {
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
                const int a1 = ge < hi ? ge : hi;
                shift[gi] = lo - prev_end;
                cum[gi] = (a1 > lo ? a1 : lo) - lo;
                prev_end = ge;
            }
        }
        cum0 = cum[0]; cum1 = cum[1]; cum2 = cum[2]; cum3 = cum[3];
        sh0 = shift[0]; sh1 = shift[1]; sh2 = shift[2]; sh3 = shift[3];
    }
    struct { int stop; } w{cum3};
    // walk index t (< w.stop) -> problem gi and its problem-local tile id
    auto ws_locate = [&](int, int t, int& gi, int& local) {
        gi = t >= cum2 ? 3 : (t >= cum1 ? 2 : (t >= cum0 ? 1 : 0));
        local = t + (gi == 3 ? sh3 : (gi == 2 ? sh2 : (gi == 1 ? sh1 : sh0)));
    };
    // problem gi of the group, read straight from the kernel-argument segment with a wave-uniform index (scalar loads).
    // Selecting among ga.g[0..3] by value made the compiler copy the whole argument block to scratch memory and index it
    // per lane.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) GroupArgs* KargPtr;
    const KargPtr karg = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
#define GN_WS_PROBLEM(dst, gi_) dst = karg->g[__builtin_amdgcn_readfirstlane(gi_)]
#else
#define GN_WS_PROBLEM(dst, gi_) dst = ga.g[gi_]      /* host pass of the single-source compile: never executed */
#endif
    const int stride = gridDim.x >> 3;
    int idx = blockIdx.x >> 3;
    if (idx >= w.stop) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    if (wave < 4) {
        // ============================================================ consumers
        // One wave per SIMD issues all the MFMAs, so nothing else covers its latencies; the stream is software-pipelined by
        // hand.  A slab = 4 groups (k-step 0 / 1 x row pairs {0,1} / {2,3}); a group = 6 ds_read_b128 (2 row tiles x 3
        // planes) + 12 MFMAs on two accumulators.  The fragments of group g + 1 are read while group g's MFMAs run --
        // across the slab barrier too: the A planes are triple-buffered in LDS and the producers stage two slabs ahead,
        // so slab kt + 1 is already visible during slab kt.  The weights of a whole slab are loaded one slab ahead.
        const int frow = lane & 31;
        const __bf16* Abase = Abuf + frow * SPLIT_PB + (lane >> 5) * 8;
        f32x16 acc[4];
        uint4 bw[2][2][NP];                                        // [slab parity][k-step][plane]
        bf16x8 fa[2][NP], fb[2][NP];                               // two fragment groups in flight (16-bit x 8: bf16 or fp16 bits)
        unsigned e_acc = 0x88888888u;                              // F16: the four block exponents (bytes) the accumulators are held in
        int ewt = 0;                                               // F16: exponent of the weight tensor
        auto load_g = [&](const __bf16* Ap, int ks, int pair, bf16x8 (&f)[2][NP]) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s_ = 0; s_ < NP; ++s_)
                    f[i][s_] = *reinterpret_cast<const bf16x8*>(Ap + s_ * WS_APL + (2 * pair + i) * 32 * SPLIT_PB + ks * 16);
        };
        auto rescale = [&](int buf) {                              // F16: accumulators to the exponents slab buffer `buf` was staged with
            const unsigned en = (unsigned)__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(exps + buf * 4));
            if (__builtin_expect(en != e_acc, 0)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float f = ldexpf(1.0f, (int)(signed char)(e_acc >> (8 * q)) - (int)(signed char)(en >> (8 * q)));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][4 * q + r] *= f;
                }
                e_acc = en;
            }
        };
        auto mfma12 = [&](const bf16x8 (&f)[2][NP], const uint4 (&w3)[NP], f32x16& c0, f32x16& c1) {
            if constexpr (F16) {
                constexpr int TA[3] = {1, 0, 0};                   // lo*hi, hi*lo, hi*hi
                constexpr int TB[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[0][TA[t]]), __builtin_bit_cast(f16x8, w3[TB[t]]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[1][TA[t]]), __builtin_bit_cast(f16x8, w3[TB[t]]), c1, 0, 0, 0);
                }
                return;
            } else {
            // smallest terms first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi): the same order per accumulator as
            // the 4-wave kernel, so the results are bit-identical to it
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (!(GN_WS_ABL & 8)) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0][TA[t]], __builtin_bit_cast(bf16x8, w3[TB[t]]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1][TA[t]], __builtin_bit_cast(bf16x8, w3[TB[t]]), c1, 0, 0, 0);
                }
            }
            }
        };
        int rb = 0;                                                // LDS buffer of the current slab (global slab count mod 3)
        bool first = true;
        __syncthreads();                                           // P: slabs 0 and 1 of the first tile are staged
        load_g(Abase, 0, 0, fa);
        for (int tile_no = 0;; ++tile_no) {
            GN_WTR(0);
            int gi, local;
            ws_locate(0, idx, gi, local);
            GemmArgs p;
            GN_WS_PROBLEM(p, gi);
            const int nk = p.K / BK, ks2 = 2 * nk;
            auto weights_of = [&](const GemmArgs& q, int loc) -> const uint4* {
                const int tiles_n = (q.N + WS_BN - 1) / WS_BN;
                const int n0 = (loc % tiles_n) * WS_BN;
                const int nt_last = (q.N + 31) / 32 - 1;
                int nt = n0 / 32 + wave;
                nt = nt < nt_last ? nt : nt_last;                  // column block past N: any valid block (never stored)
                return reinterpret_cast<const uint4*>(q.W) + (F16 ? 16 : 0) + (size_t)nt * (2 * (q.K / BK)) * (NP * 64) + lane;
            };
            const uint4* wf = weights_of(p, local);
            if constexpr (F16) {
                ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));
                e_acc = 0x88888888u;
            }
            const int next = idx + stride;
            const bool has_next = next < w.stop;
            const uint4* wf_n = wf;                                // weights of the next tile's first slab (prefetched in the last slab)
            if (has_next) {
                int gn_, ln_;
                ws_locate(0, next, gn_, ln_);
                GemmArgs pn;
                GN_WS_PROBLEM(pn, gn_);
                wf_n = weights_of(pn, ln_);
            }
            auto load_b = [&](const uint4* wq, int g, uint4 (&q)[NP]) {
#pragma unroll
                for (int s_ = 0; s_ < NP; ++s_) q[s_] = wq[(size_t)(g * NP + s_) * 64];
            };
            if (first) {
                load_b(wf, 0, bw[0][0]);
                load_b(wf, 1, bw[0][1]);
                first = false;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            auto slab = [&](int kt, auto PAR) {
                constexpr int par = decltype(PAR)::value;          // kt & 1 (nk is even: the parity runs on across tiles)
                const int rb1 = rb == WS_NBUF - 1 ? 0 : rb + 1;
                const __bf16* Ap = Abase + rb * WS_STAGE;
                const __bf16* An = Abase + rb1 * WS_STAGE;
                const bool last = kt + 1 == nk;
                const uint4* wq = last ? wf_n : wf;                // next slab's weights: a slab time to arrive
                const int g0 = last ? 0 : 2 * kt + 2;
                load_b(wq, g0, bw[par ^ 1][0]);
                load_b(wq, g0 + 1, bw[par ^ 1][1]);
                load_g(Ap, 0, 1, fb);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) rescale(rb);
                mfma12(fa, bw[par][0], acc[0], acc[1]);
                load_g(Ap, 1, 0, fa);
                __builtin_amdgcn_sched_barrier(0);
                mfma12(fb, bw[par][0], acc[2], acc[3]);
                load_g(Ap, 1, 1, fb);
                __builtin_amdgcn_sched_barrier(0);
                mfma12(fa, bw[par][1], acc[0], acc[1]);
                load_g(An, 0, 0, fa);                              // first group of the NEXT slab (staged two slabs ahead)
                __builtin_amdgcn_sched_barrier(0);
                mfma12(fb, bw[par][1], acc[2], acc[3]);
                if (kt == 3) GN_WTR(3);
                __syncthreads();                                   // S_kt
                if (kt == 3) GN_WTR(5);
                rb = rb1;
            };
            for (int kt = 0; kt < nk; kt += 2) {                   // nk is even
                slab(kt, std::integral_constant<int, 0>{});
                slab(kt + 1, std::integral_constant<int, 1>{});
            }
            GN_WTR(1);
            // accumulators -> staging tile (the producers finished draining the previous tile before S_{nk-1})
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][r];
                    if constexpr (F16) v *= ldexpf(1.0f, (int)(signed char)(e_acc >> (8 * (r >> 2))) + ewt);
                    Cst[row * WS_CP + wave * 32 + (lane & 31)] = v;
                }
            __syncthreads();                                       // E: the tile is staged
            GN_WTR(2);
            if (!has_next) break;
            idx = next;
        }
        return;
    }

    // ================================================================ producers
    __builtin_amdgcn_s_setprio(GN_WS_PRIO);                       // issue ahead of the MFMA stream of the partner wave
    const int ptid = tid - 256;
    const int sr = ptid >> 3, c4 = ptid & 7;                      // A staging: (row sr + 32 i, float4 column c4)
    const int drow = ptid >> 5, dcc = (ptid & 31) * 4;            // drain: (row drow + 8 it, float4 column group dcc)

    // ---- fetch context: the tile whose A slabs are being fetched (the consumers' tile, or the one after it).  Rows
    // past M are clamped to a valid row instead of being zeroed: their accumulator rows are never stored, and K % 32 == 0
    // here, so the split needs no masking at all -- the producers' instruction count is what bounds this kernel (they
    // share each SIMD's issue port with a wave that issues MFMAs back to back).
    const float *fA = nullptr, *fA2 = nullptr, *fA3 = nullptr;     // (K-segmented A: segment s comes from fA / fA2 / fA3)
    size_t f_off[4] = {0, 0, 0, 0};                                // element offset of [row sr + 32 i][4 c4] of the fetch tile
    int f_seg = 0;
    auto set_fetch_tile = [&](int t) {
        int gi, local;
        ws_locate(0, t, gi, local);
        GemmArgs p;
        GN_WS_PROBLEM(p, gi);
        const int tiles_n = (p.N + WS_BN - 1) / WS_BN;
        const int m0 = (local / tiles_n) * WS_BM;
        f_seg = p.a_seg; fA = p.A; fA2 = p.A2; fA3 = p.A3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + sr + 32 * i;
            const int row = (gm < p.M ? gm : 0) * p.row_gstride + p.row_goff;     // identity row map (launcher)
            f_off[i] = (size_t)row * p.lda + 4 * c4;
        }
    };
    float4 qa[2][4];
    auto fetchA = [&](int k0, float4 (&q)[4]) {                    // branch-free: exact s_waitcnt counts
        const float* Ab = fA;
        if (f_seg) {                                               // K-segmented A (a slab never straddles a segment)
            const bool s2 = k0 >= 2 * f_seg, s1 = k0 >= f_seg;
            Ab = s2 ? fA3 : (s1 ? fA2 : fA);
            k0 -= s2 ? 2 * f_seg : (s1 ? f_seg : 0);
        }
        Ab += k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = ld4(Ab + f_off[i]);
    };
    int pb = 2;                                                    // LDS buffer the next staged slab goes to (global slab count mod 3)
    int e_run = -120;                                              // F16: running exponent of this wave's rows in the tile being staged
    auto stashA = [&](int buf, const float4 (&q)[4]) {
        if constexpr (F16) {
            _Float16* d0 = reinterpret_cast<_Float16*>(Abuf) + buf * WS_STAGE + sr * SPLIT_PB + 4 * c4;
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(q[i].x), fabsf(q[i].y)), fmaxf(fabsf(q[i].z), fabsf(q[i].w))));
            int need = (int)((wave_umax_sgpr(__float_as_uint(m)) >> 23) & 0xffu) - 126 - 15;
            if (__builtin_expect(need > 112, 0)) {                 // an Inf in the block: scale by its finite values
                asm volatile("" ::: "memory");
                float mf = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) mf = fmaxf(mf, fabsf(c[t]) <= 3.0e38f ? fabsf(c[t]) : 0.f);
                }
                need = (int)((wave_umax_sgpr(__float_as_uint(mf)) >> 23) & 0xffu) - 126 - 15;
            }
            need = need < -120 ? -120 : need;
            e_run = need > e_run ? need : e_run;
            const float scale = __uint_as_float((unsigned)(127 - e_run) << 23);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x4 h, l;
                split4_f16(q[i], scale, h, l);
                _Float16* d = d0 + 32 * i * SPLIT_PB;
                *reinterpret_cast<f16x4*>(d) = h;
                *reinterpret_cast<f16x4*>(d + WS_APL) = l;
            }
            if ((ptid & 63) == 0) exps[buf * 4 + (ptid >> 6)] = (signed char)e_run;
        } else {
        __bf16* d0 = Abuf + buf * WS_STAGE + sr * SPLIT_PB + 4 * c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 h, m, l;
            split4_trunc(q[i], h, m, l);
            __bf16* d = d0 + 32 * i * SPLIT_PB;
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + WS_APL) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * WS_APL) = l;
        }
        }
    };

    // ---- drain context: the tile sitting in the staging buffer.  Everything a stored row needs is precomputed per tile
    // (identity row map only -- checked by the launcher -- so a row's offset is linear in its index): the per-slab drain
    // is a handful of instructions, and the slab bodies below are small LOOPS (a fully unrolled producer was ~25 k
    // instructions of straight-line code that never repeats inside a tile: instruction-fetch bound).
    float* d_C = nullptr; float* d_pre = nullptr;
    const float *d_res = nullptr, *d_gate = nullptr;
    size_t d_step = 0;                                             // elements between this thread's consecutive rows (8 rows)
    int d_rows = 0;                                                // rows it with it * 8 < d_rows are stored (0: none)
    float4 d_bias = zero4();
    bool d_act = false, d_nt = false, d_dsilu = false;
    int d_kind = 0;
    auto set_drain_tile = [&](int t) {
        int gi, local;
        ws_locate(0, t, gi, local);
        GemmArgs p;
        GN_WS_PROBLEM(p, gi);
        const int tiles_n = (p.N + WS_BN - 1) / WS_BN;
        const int m0 = (local / tiles_n) * WS_BM, n0 = (local % tiles_n) * WS_BN;
        const int gn = n0 + dcc, gm = m0 + drow;
        const bool cok = gn < p.N;
        const size_t off = ((size_t)gm * p.row_gstride + p.row_goff) * p.ldc + (cok ? gn : 0);
        d_C = p.C + off;
        d_pre = p.pre_out ? p.pre_out + off : nullptr;
        d_res = p.res ? p.res + off : nullptr;
        d_gate = p.gate ? p.gate + off : nullptr;
        d_step = (size_t)8 * p.row_gstride * p.ldc;
        d_rows = cok ? p.M - gm : 0;
        d_bias = (p.bias && cok) ? ld4(p.bias + gn) : zero4();
        d_act = gn >= p.act_lo && gn < p.act_hi;
        d_nt = p.nt_store != 0;
        d_dsilu = p.gate_mode != 0;
        d_kind = p.act_kind;
    };
    auto drain_store = [&](int it, float4 rv, float4 gv) {
        if (it * 8 >= d_rows) return;
        float4 v = ld4(&Cst[(it * 8 + drow) * WS_CP + dcc]) + d_bias;
        const size_t o = (size_t)it * d_step;
        if (d_pre) st4(d_pre + o, v);
        if (d_act) v = act4(v, d_kind);
        if (d_gate) v = v * (d_dsilu ? dact4(gv, d_kind) : gv);
        if (d_res) v = rv + v;
        if (d_nt) st4_nt(d_C + o, v); else st4(d_C + o, v);
    };
    // residual / gate rows of chunk j (4 rows) -- loaded one slab before they are used; clamped to a valid row
    auto drain_load = [&](int j, float4 (&r)[4], float4 (&g)[4]) {
        const float* rb = d_res ? d_res : (d_gate ? d_gate : d_C);
        const float* gb = d_gate ? d_gate : rb;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = j * 4 + u;
            const size_t o = (it * 8 < d_rows) ? (size_t)it * d_step : 0;
            r[u] = ld4(rb + o);
            g[u] = ld4(gb + o);
        }
    };

    // one slab of producer work: slab kt + 2 (of this tile, or slab 0 / 1 of the next) -> LDS, slab kt + 4 -> its register
    // set (SET = kt & 1), plus this slab's share of the drain.  DK 0: no drain; 1: rows 2 kt, 2 kt + 1 (plain tile, slabs 0-7);
    // 2: chunk kt of a residual / gate tile (slabs 0-3): its rows were loaded during the previous slab.
    float4 rres[4], rgat[4];
    auto slab = [&](int kt, int nk, int next_tile, auto SET, auto DK) {
        constexpr int set = decltype(SET)::value, dk = decltype(DK)::value;
        float4 nr[4], ng[4];
        if constexpr (dk == 2) drain_load(kt + 1 < 4 ? kt + 1 : 3, nr, ng);
        if constexpr (F16) { if (kt + 2 == nk) e_run = -120; }     // the slab staged now is slab 0 of the NEXT tile
        if (!(GN_WS_ABL & 1)) stashA(pb, qa[set]);                 // slab kt + 2 (of this tile, or slab 0 / 1 of the next)
        pb = pb == WS_NBUF - 1 ? 0 : pb + 1;
        const int v = kt + 4;                                      // ... and its register set takes slab kt + 4
        if (v == nk) set_fetch_tile(next_tile);                    // the last four fetches of a tile belong to the next one
        if (!(GN_WS_ABL & 2)) fetchA((v >= nk ? v - nk : v) * BK, qa[set]);
        if constexpr (dk == 1 && !(GN_WS_ABL & 4)) {
            drain_store(2 * kt, zero4(), zero4());
            drain_store(2 * kt + 1, zero4(), zero4());
        } else if constexpr (dk == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) drain_store(kt * 4 + u, rres[u], rgat[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) { rres[u] = nr[u]; rgat[u] = ng[u]; }
        }
        __syncthreads();                                           // S_kt
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>;

    // per tile: the slabs that carry drain work as one loop of pairs, the rest as another (nk even, >= 8)
    auto tile = [&](int nk, int next_tile, auto DK) {
        constexpr int dk = decltype(DK)::value;
        constexpr int LA = dk == 1 ? 8 : (dk == 2 ? 4 : 0);
        if constexpr (dk == 2) drain_load(0, rres, rgat);
        for (int kt = 0; kt < LA; kt += 2) {
            slab(kt, nk, next_tile, S0{}, DK);
            slab(kt + 1, nk, next_tile, S1{}, DK);
        }
        for (int kt = LA; kt < nk; kt += 2) {
            slab(kt, nk, next_tile, S0{}, K0{});
            slab(kt + 1, nk, next_tile, S1{}, K0{});
        }
    };

    // ---- prologue: slabs 0 and 1 staged, slabs 2 and 3 in flight
    set_fetch_tile(idx);
    fetchA(0, qa[0]);
    fetchA(BK, qa[1]);
    stashA(0, qa[0]);
    stashA(1, qa[1]);
    fetchA(2 * BK, qa[0]);
    fetchA(3 * BK, qa[1]);
    __syncthreads();                                               // P
    int drain_kind = 0;                                            // of the tile in the staging buffer (0: none yet)
    for (;;) {
        int gi, local;
        ws_locate(0, idx, gi, local);
        GemmArgs p;
        GN_WS_PROBLEM(p, gi);
        const int nk = p.K / BK;
        const int next = idx + stride;
        const bool has_next = next < w.stop;
        const int next_tile = has_next ? next : idx;               // no next tile: harmless re-fetch of this one
        if (drain_kind == 0) tile(nk, next_tile, std::integral_constant<int, 0>{});
        else if (drain_kind == 1) tile(nk, next_tile, std::integral_constant<int, 1>{});
        else tile(nk, next_tile, std::integral_constant<int, 2>{});
        __syncthreads();                                           // E: this tile is in the staging buffer
        set_drain_tile(idx);
        drain_kind = (p.res || p.gate) ? 2 : 1;
        if (!has_next) break;
        idx = next;
    }
    // the last tile: nothing left to overlap with
    for (int j = 0; j < 4; ++j) {
        if (drain_kind == 2) drain_load(j, rres, rgat);
#pragma unroll
        for (int u = 0; u < 4; ++u) drain_store(j * 4 + u, rres[u], rgat[u]);
    }
}

__global__ __launch_bounds__(512) void gemm_bf16x3_ws(const GroupArgs ga) { gemm_ws_body<false>(ga); }
__global__ __launch_bounds__(512) void gemm_f16x2_ws(const GroupArgs ga) { gemm_ws_body<true>(ga); }

}  // namespace gn

// host side: may this group run on the wave-specialised kernel?
bool gn_gemm_ws_eligible(const gn::GemmArgs* g, int n) {
    for (int i = 0; i < n; ++i) {
        if (g[i].K % 64 || g[i].K < 256 || g[i].pro_mode || g[i].a_gate || g[i].row_cnt != 1) return false;
    }
    return n > 0;
}

int gn_gemm_ws_launch(const gn::GroupArgs& ga, long tiles, hipStream_t st, int mode) {
    long grid = 8L * ((tiles + 7) / 8);
    if (grid > 256) grid = 256;                                    // one 512-thread workgroup per CU
    if (mode == 2) hipLaunchKernelGGL(gn::gemm_f16x2_ws, dim3((unsigned)grid), dim3(512), 0, st, ga);
    else hipLaunchKernelGGL(gn::gemm_bf16x3_ws, dim3((unsigned)grid), dim3(512), 0, st, ga);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
