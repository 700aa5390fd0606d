// gn_gemm_panel.hip -- the 2 x fp16-split projection for problems too small to hide their own latency.
//
// The slab kernel of gn_gemm.hip walks K in 32-deep slabs: fetch -> split -> LDS -> barrier -> 2 k-steps.  With hundreds of
// tiles per CU two to four co-resident workgroups hide that chain behind each other; an atom-sized product (N = 2688 rows)
// has ONE tile per CU and pays the chain once per slab: measured 0.9 us per slab, 14-21 us per launch for 0.7 GFLOP
// (reference call sites gotennet.py:400-405, 432-441, 728, 738 and their input-gradients).
// Here a workgroup owns a 32 x 128 tile (four waves side by side, one 32 x 32 MFMA tile each): the whole K extent of its 32
// A rows (K = 128 / 256 / 512) is requested at once, split into the two fp16 planes with ONE block exponent per wave (its
// eight rows) and tile, parked in LDS as a [32][K] panel, and the k-steps run back to back off the panel with the weight
// fragments (fragment-major planes of gn_split_f16x2, L2 -> registers) requested eight k-steps ahead: one HBM round trip,
// one barrier, then 3 MFMAs per k-step and wave until the epilogue.  (A 64 x 64 tile with 2 x 2 waves was built first and is
// slower at every size: twice the rows to request, split and park per workgroup for the same MFMA work per wave --
// [2688 x 512 x 256] 9.1 vs 7.9 us, [2688 x 256 x 512] 10.3 vs 8.6, the one-molecule [429 x 256 x 1536] 22.2 vs 18.5.)
// Deeper products (K a multiple of 256, e.g. the K = 1536 input-gradient of the edge projection in a one-molecule call)
// walk 256-deep chunks of the same panel: the next chunk's rows are in flight under the current chunk's k-steps, the block
// exponent only grows and the accumulator rows follow it (exact: powers of two), as in the slab kernel.
//
// Same arguments, prologue-free subset: pro_mode = 0 and a_gate = NULL (the launcher keeps the slab kernel for the rest);
// K-segmented A, row maps, bias / activation range / residual / gate / pre_out epilogues as in gn_gemm.hip.  The block
// exponent covers a whole panel (the slab kernel's covers the slabs staged so far): results agree with the slab kernel to
// the arithmetic's bound (<= 3e-7 of the fp64 product), not bitwise.
#include "gn_gemm.h"
#include "gn_tune.h"

namespace gn {

template <int NK, bool MULTI, bool ASILU>
__global__ __launch_bounds__(256) void gemm_f16x2_panel(const GroupArgs ga) {
    constexpr int K = 32 * NK;                      // depth of a panel: all of K, or (MULTI) one chunk of a K that is a multiple of it
    constexpr int KP = K + 8;                       // fp16 per panel row: (K + 8) / 2 dwords = 4 mod 64, so the 16-lane groups
                                                    // of a ds_read_b128 (16 rows, 16 bytes each) hit 16 distinct 4-bank slots
    constexpr int BM = 32, BN = 128, CP = BN + 4;
    constexpr int NPS = 4;                          // staging passes of a wave: two rows each (its 8-row block)
    constexpr int APL = BM * KP;                    // fp16 per plane
    constexpr int KC = K / 128;                     // 128-column chunks of a row: one half-wave reads 512 contiguous bytes
    constexpr int NS = 2 * NK;                      // k-steps of 16
    constexpr int NB = MULTI ? NS : (NS < 8 ? NS : 8);   // weight fragments in flight (k-steps ahead; MULTI: a whole chunk, see the loop)
    static_assert(2 * APL * 2 >= BM * CP * 4, "the epilogue tile overlays the panel");
    __shared__ __attribute__((aligned(16))) _Float16 panel[2 * APL];
    __shared__ int exps_w;                          // the four waves' block exponents (signed bytes)

    // tile of this workgroup: the concatenated 32 x 128 tile list is cut into 8 contiguous ranges, XCD x (= blockIdx % 8)
    // takes range x -- neighbouring column tiles share their A rows through one L2
    int tiles_all = ga.tile_end[0];
#pragma unroll
    for (int gi = 1; gi < GN_MAX_GROUP; ++gi)
        if (gi < ga.n) tiles_all = ga.tile_end[gi];
    const int xcd = blockIdx.x & 7;
    const int xq = tiles_all >> 3, xr = tiles_all & 7;
    const int lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int cnt = xq + (xcd < xr ? 1 : 0);
    if ((int)(blockIdx.x >> 3) >= cnt) return;
    const int t = lo + (int)(blockIdx.x >> 3);
    // the problem this tile belongs to, read from the kernarg segment with ONE indexed scalar load (copying all four
    // descriptors and selecting cost ~500 scalar instructions before the first global load)
    int gi = 0, first = 0;
#pragma unroll
    for (int q = 1; q < GN_MAX_GROUP; ++q)
        if (q < ga.n && t >= ga.tile_end[q - 1]) {
            gi = q;
            first = ga.tile_end[q - 1];
        }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) GroupArgs* KargPtr;
    const GemmArgs p = ((KargPtr)__builtin_amdgcn_kernarg_segment_ptr())->g[__builtin_amdgcn_readfirstlane(gi)];
#else
    const GemmArgs p = ga.g[gi];                     // host pass of the single-source compile: never executed
#endif
    // (scalars of their own: a select between p.A / p.A2 / p.A3 is otherwise turned into an indexed load from a private
    //  copy of the descriptor, which then lives in scratch)
    const float* const seg_ptr[3] = {p.A, p.A2, p.A3};
    const int a_seg = p.a_seg;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = ((t - first) / tiles_n) * BM, n0 = ((t - first) % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave;                            // four waves side by side

    // MULTI: K is a run-time multiple of the chunk depth; the panel is refilled per chunk, the accumulators carry on
    const int nch = MULTI ? p.K / K : 1;
    const int ks_all = nch * NS;                     // k-steps of the whole product

    // ---- weight fragments of the first NB k-steps (L2 -> registers; nothing depends on A yet)
    const uint4* wfrag = reinterpret_cast<const uint4*>(p.W) + 16;        // 256-byte header: the weight tensor's exponent
    const int ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));
    {
        const int nt_last = (p.N + 31) / 32 - 1;
        int nt = n0 / 32 + wn;
        nt = nt < nt_last ? nt : nt_last;            // a column block past N: any valid block (never stored)
        wfrag += (size_t)nt * ks_all * 128 + lane;
    }
    uint4 bq[NB][2];
    auto load_b = [&](int g, uint4 (&q)[2]) __attribute__((always_inline)) {
        q[0] = wfrag[(size_t)(2 * g) * 64];
        q[1] = wfrag[(size_t)(2 * g + 1) * 64];
    };

    // ---- the A rows: wave w stages rows 8w..8w+7 (accumulator registers 4w..4w+3 of every wave's MFMA tile), two rows
    // per pass, a half-wave per row
    const int c = lane & 31, lr = lane >> 5;
    float4 va[NPS][KC];
    bool aok[NPS];
    size_t ro[NPS];
    if (p.row_cnt == 1) {                            // identity row map (wave-uniform branch: no per-row division code)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int r = m0 + 8 * wave + ps * 2 + lr;
            aok[ps] = r < p.M;
            ro[ps] = (size_t)((aok[ps] ? r : 0) * p.row_gstride + p.row_goff) * p.lda;       // a row past M: row 0, zeroed below
        }
    } else {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int r = m0 + 8 * wave + ps * 2 + lr;
            aok[ps] = r < p.M;
            ro[ps] = (size_t)phys_row(p, aok[ps] ? r : 0) * p.lda;
        }
    }
    // K-segmented A: a 128-column piece never straddles a segment (a_seg % 128 == 0: launcher)
    auto fetch_chunk = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int k = ch * K + kc * 128;
            const bool s2 = a_seg && k >= 2 * a_seg, s1 = a_seg && k >= a_seg;
            const unsigned long long a0 = (unsigned long long)seg_ptr[0], a1 = (unsigned long long)seg_ptr[1], a2 = (unsigned long long)seg_ptr[2];
            const float* Ak = reinterpret_cast<const float*>(s2 ? a2 : (s1 ? a1 : a0)) + (k - (s2 ? 2 * a_seg : (s1 ? a_seg : 0))) + 4 * c;
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) va[ps][kc] = ld4(Ak + ro[ps]);
        }
    };
    fetch_chunk(0);
#pragma unroll
    for (int g = 0; g < NB; ++g) load_b(g, bq[g]);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int e_run = -120;                                // staging side: this wave's block exponent (only grows over the chunks)
    unsigned e_acc = 0x88888888u;                    // MFMA side: the exponents (bytes) the accumulator rows are held in
    const _Float16* Ap = panel + (lane & 31) * KP + (lane >> 5) * 8;

    for (int ch = 0; ch < nch; ++ch) {
        // ---- block exponent of this wave's 16 rows over the chunk, planes -> LDS
        float m = 0.f;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                float4& v = va[ps][kc];
                if (!aok[ps]) v = zero4();
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        int need = (int)((wave_umax_sgpr(__float_as_uint(m)) >> 23) & 0xffu) - 126 - 15;         // |x| < 2^(need + 15)
        if (__builtin_expect(need > 112, 0)) {       // an Inf in the block: scale by the finite values, so that only the
            float mf = 0.f;                          // rows that hold it turn non-finite (as in gn_gemm.hip)
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const float cv[4] = {va[ps][kc].x, va[ps][kc].y, va[ps][kc].z, va[ps][kc].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) mf = fmaxf(mf, fabsf(cv[q]) <= 3.0e38f ? fabsf(cv[q]) : 0.f);
                }
            need = (int)((wave_umax_sgpr(__float_as_uint(mf)) >> 23) & 0xffu) - 126 - 15;
        }
        need = need < -120 ? -120 : need;
        e_run = need > e_run ? need : e_run;
        const float scale = __uint_as_float((unsigned)(127 - e_run) << 23);     // 2^-e_run
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int r = 8 * wave + ps * 2 + lr;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                f16x4 h, l;
                split4_f16(va[ps][kc], scale, h, l);
                _Float16* d = panel + r * KP + kc * 128 + 4 * c;
                *reinterpret_cast<f16x4*>(d) = h;
                *reinterpret_cast<f16x4*>(d + APL) = l;
            }
        }
        if (lane == 0) reinterpret_cast<signed char*>(&exps_w)[wave] = (signed char)e_run;
        __syncthreads();
        // the next chunk's rows go in flight now: they are older than every weight fragment requested below, so no
        // fragment wait of THIS chunk's k-steps (ring of a whole chunk when MULTI) ever queues behind them.  (Two chunks
        // ahead in a second register set: slower at every size -- 254 VGPRs, [429 x 256 x 1536] 18.5 -> 19.4 us.)
        if (MULTI && ch + 1 < nch) fetch_chunk(ch + 1);
        const unsigned en = (unsigned)__builtin_amdgcn_readfirstlane(exps_w);
        if (MULTI && __builtin_expect(en != e_acc && ch > 0, 0)) {               // the block exponents grew: bring the
            asm volatile("" ::: "memory");                                     // accumulator rows along (exact: powers of two)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float f = ldexpf(1.0f, (int)(signed char)(e_acc >> (8 * q)) - (int)(signed char)(en >> (8 * q)));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[4 * q + r] *= f;
            }
        }
        e_acc = en;

        // ---- k-steps off the panel: x = hi + lo per operand; lo*hi, hi*lo, hi*hi (lo*lo is below 2^-22 of the product)
        const int g0 = ch * NS;
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(Ap + g * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(Ap + APL + g * 16);
            const f16x8 bh = __builtin_bit_cast(f16x8, bq[g % NB][0]), bl = __builtin_bit_cast(f16x8, bq[g % NB][1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            if (MULTI) { if (g0 + g + NB < ks_all) load_b(g0 + g + NB, bq[g % NB]); }
            else if (g + NB < NS) load_b(g + NB, bq[g % NB]);
        }
        __syncthreads();                             // the next chunk's planes / the epilogue tile overlay the panel
    }

    // ---- epilogue through LDS: accumulators -> [64][68] floats -> coalesced float4 rows
    // (a lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of its 32 x 32 tile)
    float* const sc = reinterpret_cast<float*>(panel);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sc[row * CP + wn * 32 + (lane & 31)] = ldexpf(acc[r], (int)(signed char)(e_acc >> (8 * (r >> 2))) + ewt);
    }
    __syncthreads();
    constexpr int C4 = BN / 4, RP = 256 / C4;       // column groups (a thread keeps one) and rows per pass: four passes either way
    const int cc = (tid % C4) * 4, gn = n0 + cc;
    if (gn >= p.N) return;
    const float4 bias4 = p.bias ? ld4(p.bias + gn) : zero4();
    const bool act = gn >= p.act_lo && gn < p.act_hi;
    const int kind = ASILU ? (int)GN_ACT_SILU : p.act_kind;
    size_t off[4];
    bool ok[4];
    float4 rv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                    // every residual / gate row requested before the first store
        const int gm = m0 + u * RP + tid / C4;
        ok[u] = gm < p.M;
        off[u] = (size_t)phys_row(p, ok[u] ? gm : 0) * p.ldc + gn;
        rv[u] = (ok[u] && p.res) ? ld4(p.res + off[u]) : zero4();
        gv[u] = (ok[u] && p.gate) ? ld4(p.gate + off[u]) : zero4();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        float4 v = ld4(&sc[(u * RP + tid / C4) * CP + cc]) + bias4;
        if (p.pre_out) st4(p.pre_out + off[u], v);
        if (act) v = act4(v, kind);
        if (p.gate) v = v * (p.gate_mode ? dact4(gv[u], kind) : gv[u]);
        if (p.res) v = rv[u] + v;
        st4(p.C + off[u], v);
    }
}

}  // namespace gn

// Launch the panel kernel for a validated group when it applies; returns 0 when the caller must use the slab kernel,
// 1 after a launch, < 0 on a launch error.  Applies to: no prologue, K-segments in whole 128-column pieces, at most
// GN_GEMM_PANEL_MAX tiles of 32 x 128, and either ONE depth K in {128, 256, 512} for the whole group (one panel per tile)
// or depths that are all multiples of 256 (K = 768 ... 1536 and mixed groups: 256-deep chunks -- what a one-molecule
// call's K-heavy input-gradient products need: 48 slab chains in a row are 27 us for 28 tiles).
int gn_gemm_panel_launch(const gn::GemmArgs* g, int n, hipStream_t st) {
    if (GN_GEMM_PANEL_MAX <= 0) return 0;
    const int K = g[0].K;
    gn::GroupArgs ga;
    bool silu = true, same = true, m256 = true;
    for (int i = 0; i < n; ++i) {
        if (g[i].pro_mode != 0 || g[i].a_gate != nullptr || (g[i].a_seg % 128) != 0) return 0;
        if (g[i].N < 128 && g[i].M > 4096) return 0;    // a long narrow product ([E x 32 x 512]: 33 us on 64 x 64 slab tiles, 52 us here: three of four waves idle)
        same = same && g[i].K == K;
        m256 = m256 && (g[i].K % 256) == 0;
        silu = silu && g[i].act_kind == GN_ACT_SILU;
    }
    const bool single = same && (K == 128 || K == 256 || K == 512);
    if (!single && !m256) return 0;
    long end = 0;
    for (int i = 0; i < gn::GN_MAX_GROUP; ++i) {
        ga.g[i] = g[i < n ? i : n - 1];
        ga.g[i].nt_store = 0x7fffffff;
        if (i < n) end += (long)((g[i].M + 31) / 32) * ((g[i].N + 127) / 128);
        ga.tile_end[i] = (int)end;
    }
    ga.n = n;
    ga.spread = 0;
    if (end == 0 || end > (long)GN_GEMM_PANEL_MAX) return 0;
    const unsigned grid = (unsigned)(8L * ((end + 7) / 8));
#define GN_PANEL_GO(NK_, MULTI_)                                                                                       \
    do {                                                                                                              \
        if (silu) hipLaunchKernelGGL((gn::gemm_f16x2_panel<NK_, MULTI_, true>), dim3(grid), dim3(256), 0, st, ga);     \
        else hipLaunchKernelGGL((gn::gemm_f16x2_panel<NK_, MULTI_, false>), dim3(grid), dim3(256), 0, st, ga);         \
    } while (0)
    if (!single) GN_PANEL_GO(8, true);
    else if (K == 128) GN_PANEL_GO(4, false);
    else if (K == 256) GN_PANEL_GO(8, false);
    else GN_PANEL_GO(16, false);
#undef GN_PANEL_GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    return 1;
}
