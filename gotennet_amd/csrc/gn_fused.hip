// gn_fused.hip -- the GATA message stage WITHOUT the [E, (1+M)F] edge-projection stream (second half).
//
// The reference materialises  t_attn | t_filter = [W_re; W_rs] t_ij + b  for every edge (gotennet.py:406-407) and
// consumes it in `message` (452-559) and `aggregate` (613-640).  On the inference path (nothing saved for a backward)
// that stream -- 334 MB per layer at C2, written by the edge projection and read back by the softmax and the message
// kernel -- need not exist: this kernel is the edge projection with the message stage as its epilogue.
//
//   * Work unit: a TILE of <= 128 consecutive edge rows cut on CSR target boundaries (gn_edge_tiles: ~6 targets at C2),
//     so every target's incoming edges -- its softmax segment and its aggregation segment -- live in ONE workgroup.
//   * Column passes of 128 over the projection's output columns.  Each pass is one 128 x 128 x F MFMA tile (the same
//     operand path as gn_gemm: A rows staged through LDS as fp16 / bf16 planes, fragment-major weight planes L2 ->
//     registers, 1 x 4 wave grid), whose accumulators go to an LDS tile T[128][132] instead of HBM.
//   * Passes over W_re (columns [0, F)):  scores s[e,h] = sum_c q_i k_j SiLU(T)  ->  LDS;  then the per-target segment
//     softmax in LDS (the arithmetic of gn_attn_softmax).
//   * Passes over W_rs (one F-wide block of the value vector at a time):  o = T * x_j * cut + a * v_j consumed in place:
//     scalar block -> h, direction gate of degree l -> rl (x) o, tensor gate of degree l -> X_j * o, reduced over the
//     target's rows by ONE wave (its two half-waves take alternate rows, combined by one cross-half add: fixed order,
//     no atomics) and written straight to h_out / X_out.
//   * A target with more than 128 incoming edges is a tile of its own, walked in chunks of 128 rows with its scores in
//     the caller's [E, H] scratch (correct, not fast: one wave consumes).
// Measured on MI355X (round 4, tools/fused_ab.py): per layer 297 us vs 293 us for the three kernels at C2,
// 2-10 % SLOWER on every workload -- the three-kernel sequence is bound by L2 -> CU operand traffic (weight fragments, A
// slabs, gathered source rows: the same bytes here), not by the HBM stream this kernel removes, and the two phases of a
// tile do not overlap with the co-resident workgroup's (ablations: MFMA passes alone 153 us, consumers alone 140 us).
// So it is built, held to the fixtures, and OPT-IN (GotenNet.fuse_message = True); the default inference path stays the
// three-kernel sequence.
// Arithmetics: the two plane modes of gn_gemm (2 x fp16 with block exponents, 3 x bf16); row-wise in the bf16 mode, so
// there t_attn / t_filter are the bits gn_gemm_split would have written.  The reduction order over a target's edges
// differs from gn_message_aggregate's four slots, so (h, X) agree with the two-kernel path to fp32 rounding, not bits.
#include <type_traits>
#include "gn_gemm.h"
#include "gn_tune.h"

namespace gn {

constexpr int FT_ROWS = 128;        // edge rows per tile / chunk (four 32-row MFMA tiles per wave)
constexpr int FT_COLS = 128;        // projection columns per pass (one 32-column MFMA tile per wave)
constexpr int FT_CP = FT_COLS + 4;  // pitch of the LDS tile
constexpr int FT_MAXT = 128;        // targets per tile
constexpr int FT_CHUNK = 256;       // targets per packing chunk of gn_edge_tiles (tiles never straddle a chunk)
constexpr int FT_MAXH = 16;         // heads (LDS strip of scores: 128 rows x H)

struct FusedArgs {
    const float* t;                 // [E, F] edge state
    const void* W;                  // packed planes of [W_re; W_rs] (or its prefix without the tensor-gate blocks)
    const float* bias;              // [(1 + blocks) F] or NULL
    const float* q; const float* k; int ldqk;
    const float* x; const float* v; int ldxv;
    const float* X_in; const float* h_in; float* h_out; float* X_out;
    const float* rl; const float* cut;
    const int* rowptr; const int* src; const int* outdeg;
    const int* tile_first; const int* n_tiles;
    float* attn_ws;                 // [E, H] scratch (scores / weights of targets with more than FT_ROWS incoming edges)
    int N, F, H, lmax, M, sep_dir, sep_tensor;
    float inv_sqrt_f;
};

// ------------------------------------------------------------------------------------------ tile list
// Greedy packing of consecutive targets into tiles of <= FT_ROWS edge rows and <= FT_MAXT targets, independently per
// chunk of FT_CHUNK targets (one thread per chunk: the packing is a serial scan).  A target with more rows than a tile
// holds becomes a tile of its own.  tile_first[k] = first target of tile k, tile_first[n_tiles] = N.
__device__ __forceinline__ int pack_chunk(const int* __restrict__ rowptr, int N, int c, int* __restrict__ out) {
    const int i_end = min(N, (c + 1) * FT_CHUNK);
    int i = c * FT_CHUNK, n = 0;
    while (i < i_end) {
        if (out) out[n] = i;
        ++n;
        const int r0 = rowptr[i];
        int cnt = 1;
        ++i;
        while (i < i_end && cnt < FT_MAXT && rowptr[i + 1] - r0 <= FT_ROWS) { ++i; ++cnt; }
    }
    return n;
}

__global__ __launch_bounds__(1024) void edge_tiles_kernel(const int* __restrict__ rowptr, int N, int cap,
                                                          int* __restrict__ tile_first, int* __restrict__ n_tiles) {
    __shared__ int total;
    const int nchunks = (N + FT_CHUNK - 1) / FT_CHUNK;
    // offsets: thread c needs the tile count of every chunk before c.  Chunks are few (N / 256) and this runs once per
    // topology: every thread recounts its predecessors' chunks through a strided two-level sum.
    extern __shared__ int cnt[];                    // [nchunks]
    for (int c = threadIdx.x; c < nchunks; c += blockDim.x) cnt[c] = pack_chunk(rowptr, N, c, nullptr);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < nchunks; ++c) { const int n = cnt[c]; cnt[c] = run; run += n; }
        total = run;
    }
    __syncthreads();
    const bool fits = total <= cap;                 // (cap = gn_edge_tiles_cap is an upper bound of the greedy packing)
    for (int c = threadIdx.x; c < nchunks; c += blockDim.x)
        if (fits) pack_chunk(rowptr, N, c, tile_first + cnt[c]);
    if (threadIdx.x == 0) {
        const int n = fits ? total : 0;
        n_tiles[0] = n;
        tile_first[n] = N;
    }
}

// ------------------------------------------------------------------------------------------ the fused kernel
// per-(target, head) softmax over a strip S[(row - lo) * H + h], rows lo..hi of ONE target, by one wave: the second half
// of attn_softmax_wave_body (same arithmetic and order).  `S` is LDS or, for a long target, its rows of the global scratch.
__device__ __forceinline__ void softmax_strip(float* S, int n, int H, const int* __restrict__ src_rows,
                                              const int* __restrict__ outdeg, float inv_sqrt_f, int lane) {
    float mx = -INFINITY;
    for (int idx = lane; idx < n; idx += 64) mx = fmaxf(mx, S[idx]);
    for (int o = H; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sm = 0.f;
    for (int idx = lane; idx < n; idx += 64) {
        const float ex = fast_exp(S[idx] - mx);
        S[idx] = ex;
        sm += ex;
    }
    for (int o = H; o < 64; o <<= 1) sm += __shfl_xor(sm, o, 64);
    const float rsm = __builtin_amdgcn_rcpf(sm + 1e-16f);
    for (int idx = lane; idx < n; idx += 64) {
        const float nrm = outdeg ? sqrtf((float)outdeg[src_rows[idx / H]]) * inv_sqrt_f : inv_sqrt_f;
        S[idx] = S[idx] * rsm * nrm;
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void message_fused_kernel(const FusedArgs p) {
    static_assert(MODE == 1 || MODE == 2, "plane modes only");
    constexpr bool F16 = MODE == 2;
    constexpr int NP = F16 ? 2 : 3;
    constexpr int TM = 4;
    constexpr int APL = FT_ROWS * SPLIT_PB;         // 16-bit elements per A plane of a slab
    constexpr int STAGE_S = NP * APL;
    constexpr int MAIN_FLOATS = (2 * STAGE_S) / 2;
    constexpr int LDS_FLOATS = MAIN_FLOATS > FT_ROWS * FT_CP ? MAIN_FLOATS : FT_ROWS * FT_CP;
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS + 2];
    __shared__ int srcL[FT_ROWS];                   // source atom of each row of the chunk
    __shared__ int tgtL[FT_ROWS];                   // target atom of each row
    __shared__ float cutL[FT_ROWS];
    __shared__ int trow[FT_MAXT + 1];               // chunk-local first row of each target of the tile
    __shared__ __attribute__((aligned(16))) float Sl[FT_ROWS * FT_MAXH];   // scores, then attention weights a[row][h]
    signed char* const exps = reinterpret_cast<signed char*>(smem + LDS_FLOATS);

    // XCD-aware tile map over the DEVICE-side tile count (the grid is a host-side upper bound)
    const int n_tiles = p.n_tiles[0];
    const int per = (n_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= n_tiles) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31, c0l = 4 * l32;
    const int c4 = tid & 7, sr = tid >> 3;
    const int F = p.F, H = p.H, K = F, nk = K / BK;
    const int D = (p.lmax + 1) * (p.lmax + 1) - 1;
    const int ND = p.sep_dir ? p.lmax : 1, NT = p.sep_tensor ? p.lmax : 1;
    const bool first = p.X_in == nullptr;
    const int nblocks = first ? 1 + ND : 1 + ND + NT;            // value blocks computed (the projection has F more columns)
    const int per_head = (p.M * F) / H;
    const int ppb = F / FT_COLS;                                  // passes per F-wide block
    const int chd = F / H, lph = chd >> 2;                        // channels / lanes per attention head

    const int i0 = p.tile_first[tile], i1 = p.tile_first[tile + 1];
    const int r0 = p.rowptr[i0], R = p.rowptr[i1] - r0;
    const int nchunks = R > FT_ROWS ? (R + FT_ROWS - 1) / FT_ROWS : 1;
    const bool longt = nchunks > 1;                               // a single target with more rows than a tile holds
    const int nt_tile = i1 - i0;

    // ---- per-chunk row metadata -> LDS
    auto load_meta = [&](int chunk) {
        const int base = r0 + chunk * FT_ROWS;
        const int nr = min(FT_ROWS, R - chunk * FT_ROWS);
        for (int r = tid; r < FT_ROWS; r += 256) {
            const int e = base + (r < nr ? r : 0);
            srcL[r] = nr > 0 ? p.src[e] : 0;
            cutL[r] = nr > 0 ? p.cut[e] : 0.f;
        }
        if (!longt) {
            for (int s = tid; s <= nt_tile; s += 256) trow[s] = p.rowptr[i0 + s] - r0;
            __syncthreads();
            // target of each row: every target's (few) rows are stamped by one thread
            for (int s = tid; s < nt_tile; s += 256)
                for (int r = trow[s]; r < trow[s + 1]; ++r) tgtL[r] = i0 + s;
        } else {
            for (int r = tid; r < FT_ROWS; r += 256) tgtL[r] = i0;
        }
        __syncthreads();
        return nr;
    };

    // ---- one 128 x 128 x F MFMA tile: rows [rbase, rbase + nr) of t, projection columns [n0, n0 + 128) -> T (+ bias)
    f32x16 acc[TM];
    int e_run = -120;
    unsigned e_acc = 0x88888888u;
    int ewt = 0;
    float4 qa2[2][TM];
    int prow[TM];
    const uint4* wfrag = nullptr;
    size_t nt_off = 0;
    const int ks2 = 2 * nk;

    auto fetchA = [&](int k0, float4 (&qa)[TM]) {
        const int kraw = k0 + 4 * c4;
        const int kc = kraw < K ? kraw : 0;          // past K (the peeled tail's dummy fetch): any valid column, never staged
#pragma unroll
        for (int i = 0; i < TM; ++i) qa[i] = ld4(p.t + (size_t)prow[i] * F + kc);
    };
    auto stashA = [&](int sb, const float4 (&v)[TM]) {
        if constexpr (F16) {
            _Float16* d0 = reinterpret_cast<_Float16*>(smem) + sb * STAGE_S + sr * SPLIT_PB + 4 * c4;
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
            int need = (int)((wave_umax_sgpr(__float_as_uint(m)) >> 23) & 0xffu) - 126 - 15;
            if (__builtin_expect(need > 112, 0)) {   // an Inf in the block: scale by its finite values (gn_gemm.hip)
                asm volatile("" ::: "memory");
                float mf = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float c[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int t_ = 0; t_ < 4; ++t_) mf = fmaxf(mf, fabsf(c[t_]) <= 3.0e38f ? fabsf(c[t_]) : 0.f);
                }
                need = (int)((wave_umax_sgpr(__float_as_uint(mf)) >> 23) & 0xffu) - 126 - 15;
            }
            need = need < -120 ? -120 : need;
            e_run = need > e_run ? need : e_run;
            const float scale = __uint_as_float((unsigned)(127 - e_run) << 23);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f16x4 h, l;
                split4_f16(v[i], scale, h, l);
                _Float16* d = d0 + 32 * i * SPLIT_PB;
                *reinterpret_cast<f16x4*>(d) = h;
                *reinterpret_cast<f16x4*>(d + APL) = l;
            }
            if (lane == 0) exps[sb * 4 + wave] = (signed char)e_run;
        } else {
            __bf16* d0 = reinterpret_cast<__bf16*>(smem) + sb * STAGE_S + sr * SPLIT_PB + 4 * c4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x4 h, m, l;
                split4_trunc(v[i], h, m, l);
                __bf16* d = d0 + 32 * i * SPLIT_PB;
                *reinterpret_cast<bf16x4*>(d) = h;
                *reinterpret_cast<bf16x4*>(d + APL) = m;
                *reinterpret_cast<bf16x4*>(d + 2 * APL) = l;
            }
        }
    };
    auto load_b = [&](int g, uint4 (&q)[NP]) {
#pragma unroll
        for (int s_ = 0; s_ < NP; ++s_) q[s_] = wfrag[nt_off + (size_t)(g * NP + s_) * 64];
    };
    auto rescale = [&](int sb) {
        const unsigned en = (unsigned)__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(exps + sb * 4));
        if (__builtin_expect(en != e_acc, 0)) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int eo = (int)(signed char)(e_acc >> (8 * q)), e1 = (int)(signed char)(en >> (8 * q));
                const float f = ldexpf(1.0f, eo - e1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][4 * q + r] *= f;
            }
            e_acc = en;
        }
    };
    auto kstep = [&](const __bf16* Ap, int ks, const uint4 (&bw)[NP]) {
        if constexpr (F16) {
            f16x8 a[TM][2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_)
                    a[i][s_] = *reinterpret_cast<const f16x8*>(Ap + s_ * APL + i * 32 * SPLIT_PB + ks * 16);
            constexpr int TA[3] = {1, 0, 0};
            constexpr int TB[3] = {0, 1, 0};
#pragma unroll
            for (int t_ = 0; t_ < 3; ++t_)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t_]], __builtin_bit_cast(f16x8, bw[TB[t_]]),
                                                                    acc[i], 0, 0, 0);
        } else {
            bf16x8 a[TM][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_)
                    a[i][s_] = *reinterpret_cast<const bf16x8*>(Ap + s_ * APL + i * 32 * SPLIT_PB + ks * 16);
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t_ = 0; t_ < 6; ++t_)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t_]], __builtin_bit_cast(bf16x8, bw[TB[t_]]),
                                                                     acc[i], 0, 0, 0);
        }
    };

    // ---- one 128 x 128 x F MFMA tile in two halves: `gemm_prefetch` puts the first two A slabs and the first weight
    // fragments of the NEXT pass in flight (they land while the current pass is consumed), `gemm_run` does the K loop
    // and leaves accumulators + bias in T.
    uint4 bq[2][NP];
    auto gemm_rows = [&](int rbase, int nr) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = sr + 32 * i;
            prow[i] = rbase + (r < nr ? r : nr - 1);             // rows past the chunk: a valid row (their T rows are never read)
        }
    };
    auto gemm_prefetch = [&](int n0) {
        wfrag = reinterpret_cast<const uint4*>(p.W) + (F16 ? 16 : 0);
        nt_off = (size_t)(n0 / 32 + wave) * ks2 * (NP * 64) + lane;
        fetchA(0, qa2[0]);
        load_b(0, bq[0]);
        fetchA(BK, qa2[1]);
    };
    auto gemm_run = [&](int n0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[i][r_] = 0.f;
        if constexpr (F16) {
            e_run = -120;
            e_acc = 0x88888888u;
        }
        stashA(0, qa2[0]);
        __syncthreads();
        const __bf16* Abase = reinterpret_cast<const __bf16*>(smem) + (lane & 31) * SPLIT_PB + (lane >> 5) * 8;
        auto slab = [&](int kt, auto SET, auto LAST) {
            constexpr int set = decltype(SET)::value;
            constexpr bool last = decltype(LAST)::value;
            const __bf16* Ap = Abase + set * STAGE_S;
            load_b(2 * kt + 1, bq[1]);
            if constexpr (!last) fetchA((kt + 2) * BK, qa2[set]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (F16) rescale(set);
            kstep(Ap, 0, bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!last) load_b(2 * kt + 2, bq[0]);
            kstep(Ap, 1, bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!last) stashA(set ^ 1, qa2[set ^ 1]);
            __syncthreads();
        };
        using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
        using No = std::false_type; using Yes = std::true_type;
        int kt = 0;
        for (; kt + 2 < nk; kt += 2) {               // nk = F / 32 is even (F % 128 == 0)
            slab(kt, T0{}, No{});
            slab(kt + 1, T1{}, No{});
        }
        slab(kt, T0{}, No{});
        slab(kt + 1, T1{}, Yes{});
        // accumulators (+ bias) -> T.  The slab buffers alias T: the K loop's last barrier is behind us.
        const int col = wave * 32 + (lane & 31);
        const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][r];
                if constexpr (F16) v = ldexpf(v, (int)(signed char)(e_acc >> (8 * (r >> 2))) + ewt);
                smem[row * FT_CP + col] = v + bv;
            }
        __syncthreads();
    };
    if constexpr (F16) ewt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.W));

    // ---- scores of the rows of one attention pass: eight half-wave slots stride over the rows (a half-wave covers the
    // pass's 128 columns of one row), UA rows in flight per slot
    auto attention_rows = [&](int nr, int rbase, int pp) {
        constexpr int UA = 4;
        const int colg = pp * FT_COLS + c0l;
        const int hh = colg / chd;
        const bool writer = (l32 & (lph - 1)) == 0;
        for (int rb = 2 * wave + half; rb < nr; rb += 8 * UA) {
            float4 t4[UA], qi[UA], kj[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int r = rb + 8 * u < nr ? rb + 8 * u : nr - 1;
                t4[u] = ld4(&smem[r * FT_CP + c0l]);
                qi[u] = ld4(p.q + (size_t)tgtL[r] * p.ldqk + colg);
                kj[u] = ld4(p.k + (size_t)srcL[r] * p.ldqk + colg);
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const float4 a4 = act4(t4[u], GN_ACT_SILU);
                float s = qi[u].x * kj[u].x * a4.x;
                s += qi[u].y * kj[u].y * a4.y;
                s += qi[u].z * kj[u].z * a4.z;
                s += qi[u].w * kj[u].w * a4.w;
                s = group_sum(s, lph);
                const int r = rb + 8 * u;
                if (writer && r < nr) {
                    if (longt) p.attn_ws[(size_t)(rbase + r) * H + hh] = s;
                    else Sl[r * H + hh] = s;
                }
            }
        }
    };

    // ---- value blocks.  item = (block b, ROLE, degree l): ROLE 0 scalar -> h; 1 direction gate -> rl (x) o; 2 tensor gate
    // -> X_j * o.  A wave owns the targets wave, wave + 4, ... of the tile; its half-waves take alternate rows, U rows in
    // flight each; rows past the target's range are clamped to its last row with zero weight.
    auto consume = [&](auto ROLEc, auto NRc, int m0, int b, int colb, int lo, int hi, int rbase,
                       float4 (&accv)[decltype(NRc)::value]) {
        constexpr int ROLE = decltype(ROLEc)::value;
        constexpr int NR = decltype(NRc)::value;
        constexpr int U = ROLE == 2 ? (NR > 5 ? 2 : 4) : 5;
        const int col = colb + c0l;                 // column inside the F-wide block
        const int hh = (b * F + col) / per_head;
        const float* xb = p.x + (size_t)b * F + col;
        const float* vb = p.v + (size_t)b * F + col;
        for (int rb = lo + half; rb < hi; rb += 2 * U) {
            int rr[U], jj[U];
            float4 o[U];
            {
                float4 tf[U], x4[U], v4[U];
                float ab[U], ce[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = rb + 2 * u < hi;
                    rr[u] = ok ? rb + 2 * u : hi - 1;
                    jj[u] = srcL[rr[u]];
                    ce[u] = ok ? cutL[rr[u]] : 0.f;
                    ab[u] = longt ? p.attn_ws[(size_t)(rbase + rr[u]) * H + hh] : Sl[rr[u] * H + hh];
                    ab[u] = ok ? ab[u] : 0.f;
                    tf[u] = ld4(&smem[rr[u] * FT_CP + c0l]);
                    x4[u] = ld4(xb + (size_t)jj[u] * p.ldxv);
                    v4[u] = ld4(vb + (size_t)jj[u] * p.ldxv);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) o[u] = fma4(ab[u], v4[u], (tf[u] * x4[u]) * ce[u]);    // gotennet.py:516-529
            }
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) accv[0] = accv[0] + o[u];
            } else if constexpr (ROLE == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float* re = p.rl + (size_t)(rbase + rr[u]) * D + m0;
#pragma unroll
                    for (int mm = 0; mm < NR; ++mm) accv[mm] = fma4(re[mm], o[u], accv[mm]);
                }
            } else {
#pragma unroll
                for (int mm = 0; mm < NR; ++mm) {
                    float4 xj[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) xj[u] = ld4(p.X_in + ((size_t)jj[u] * D + m0 + mm) * F + col);
#pragma unroll
                    for (int u = 0; u < U; ++u) accv[mm] = fma4(xj[u], o[u], accv[mm]);
                }
            }
        }
    };
    auto finish = [&](auto ROLEc, auto NRc, int m0, int colb, int i, float4 (&accv)[decltype(NRc)::value]) {
        constexpr int ROLE = decltype(ROLEc)::value;
        constexpr int NR = decltype(NRc)::value;
        const int col = colb + c0l;
#pragma unroll
        for (int mm = 0; mm < NR; ++mm) {
            accv[mm].x += __shfl_xor(accv[mm].x, 32, 64);
            accv[mm].y += __shfl_xor(accv[mm].y, 32, 64);
            accv[mm].z += __shfl_xor(accv[mm].z, 32, 64);
            accv[mm].w += __shfl_xor(accv[mm].w, 32, 64);
        }
        if (half) return;
        if constexpr (ROLE == 0) {
            st4(p.h_out + (size_t)i * F + col, ld4(p.h_in + (size_t)i * F + col) + accv[0]);
        } else {
#pragma unroll
            for (int mm = 0; mm < NR; ++mm) {
                const size_t off = ((size_t)i * D + m0 + mm) * F + col;
                // the direction pass of a degree is the first writer of its rows, the tensor pass adds to them (same lane:
                // program order); the first interaction has no tensor pass and no X_in
                if constexpr (ROLE == 1) st4(p.X_out + off, first ? accv[mm] : ld4(p.X_in + off) + accv[mm]);
                else st4(p.X_out + off, ld4(p.X_out + off) + accv[mm]);
            }
        }
    };
    auto item = [&](auto ROLEc, auto NRc, int l, int b, int colb) {
        constexpr int NR = decltype(NRc)::value;
        const int m0 = l * l - 1;                    // first row of degree l (role 0: unused)
        for (int s = wave; s < nt_tile; s += 4) {
            float4 accv[NR];
#pragma unroll
            for (int mm = 0; mm < NR; ++mm) accv[mm] = zero4();
            consume(ROLEc, NRc, m0, b, colb, trow[s], trow[s + 1], r0, accv);
            finish(ROLEc, NRc, m0, colb, i0 + s, accv);
        }
    };
    using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
    auto items_of_block = [&](int b, int colb) {
        // roles of block b: 0 scalar; [1, 1 + ND) direction gates; [1 + ND, ...) tensor gates
        if (b == 0) { item(R0{}, std::integral_constant<int, 1>{}, 0, b, colb); return; }
        const bool dir = b < 1 + ND;
        const bool sep = dir ? p.sep_dir : p.sep_tensor;
        const int l_lo = sep ? (dir ? b : b - ND) : 1, l_hi = sep ? l_lo : p.lmax;
        for (int l = l_lo; l <= l_hi; ++l) {
            if (dir) {
                switch (l) {
                    case 1: item(R1{}, std::integral_constant<int, 3>{}, l, b, colb); break;
                    case 2: item(R1{}, std::integral_constant<int, 5>{}, l, b, colb); break;
                    case 3: item(R1{}, std::integral_constant<int, 7>{}, l, b, colb); break;
                    default: item(R1{}, std::integral_constant<int, 9>{}, l, b, colb); break;
                }
            } else {
                switch (l) {
                    case 1: item(R2{}, std::integral_constant<int, 3>{}, l, b, colb); break;
                    case 2: item(R2{}, std::integral_constant<int, 5>{}, l, b, colb); break;
                    case 3: item(R2{}, std::integral_constant<int, 7>{}, l, b, colb); break;
                    default: item(R2{}, std::integral_constant<int, 9>{}, l, b, colb); break;
                }
            }
        }
    };

    if (!longt) {
        // ============================================================= the common case: the tile is one chunk
        load_meta(0);
        const int npass = (1 + nblocks) * ppb;       // projection columns [q * 128, q * 128 + 128): attention first, then the blocks
        if (R > 0) {
            gemm_rows(r0, R);
            gemm_prefetch(0);
        }
        for (int q = 0; q < npass; ++q) {
            if (R > 0) {
                gemm_run(q * FT_COLS);
                if (q + 1 < npass) gemm_prefetch((q + 1) * FT_COLS);
            }
            if (q < ppb) {
                attention_rows(R, r0, q);
                if (q == ppb - 1) {                  // all scores are in: per-target segment softmax, one wave per target
                    __syncthreads();
                    for (int s = wave; s < nt_tile; s += 4) {
                        const int lo = trow[s], hi = trow[s + 1];
                        if (hi > lo) softmax_strip(Sl + lo * H, (hi - lo) * H, H, srcL + lo, p.outdeg, p.inv_sqrt_f, lane);
                    }
                }
            } else {
                items_of_block(q / ppb - 1, (q % ppb) * FT_COLS);
            }
            __syncthreads();                         // T is rewritten by the next pass
        }
        return;
    }

    // ================================================================= a target with more rows than a tile holds
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int rbase = r0 + chunk * FT_ROWS;
        const int nr = load_meta(chunk);
        gemm_rows(rbase, nr);
        for (int pp = 0; pp < ppb; ++pp) {
            gemm_prefetch(pp * FT_COLS);
            gemm_run(pp * FT_COLS);
            attention_rows(nr, rbase, pp);
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    if (wave == 0) softmax_strip(p.attn_ws + (size_t)r0 * H, R * H, H, p.src + r0, p.outdeg, p.inv_sqrt_f, lane);
    __threadfence_block();
    __syncthreads();

    // ---- a long target: every (block, pass, degree) item accumulates over the chunks; wave 0 consumes
    for (int b = 0; b < nblocks; ++b)
        for (int pp = 0; pp < ppb; ++pp) {
            const int colb = pp * FT_COLS;
            const bool dir = b >= 1 && b < 1 + ND;
            const int role = b == 0 ? 0 : (dir ? 1 : 2);
            const bool sep = b == 0 ? true : (dir ? p.sep_dir : p.sep_tensor);
            const int l_lo = b == 0 ? 0 : (sep ? (dir ? b : b - ND) : 1), l_hi = b == 0 ? 0 : (sep ? l_lo : p.lmax);
            for (int l = l_lo; l <= l_hi; ++l) {
                float4 accv[9];
#pragma unroll
                for (int mm = 0; mm < 9; ++mm) accv[mm] = zero4();
                const int m0 = l * l - 1;
                for (int chunk = 0; chunk < nchunks; ++chunk) {
                    const int rbase = r0 + chunk * FT_ROWS;
                    const int nr = load_meta(chunk);
                    gemm_rows(rbase, nr);
                    gemm_prefetch((1 + b) * F + colb);
                    gemm_run((1 + b) * F + colb);
                    if (wave == 0) {
                        // (NR = 9 with the rows past 2l+1 masked by re-reading row 0: a rare, correctness-only path)
                        const int col = colb + c0l;
                        const int hh = (b * F + col) / per_head;
                        const int nrow = b == 0 ? 1 : 2 * l + 1;
                        for (int r = half; r < nr; r += 2) {
                            const int j = srcL[r];
                            const float4 tf = ld4(&smem[r * FT_CP + c0l]);
                            const float4 x4 = ld4(p.x + (size_t)j * p.ldxv + (size_t)b * F + col);
                            const float4 v4 = ld4(p.v + (size_t)j * p.ldxv + (size_t)b * F + col);
                            const float ab = p.attn_ws[(size_t)(rbase + r) * H + hh];
                            const float4 o = fma4(ab, v4, (tf * x4) * cutL[r]);
#pragma unroll
                            for (int mm = 0; mm < 9; ++mm) {
                                if (mm >= nrow) continue;
                                if (role == 0) accv[mm] = accv[mm] + o;
                                else if (role == 1) accv[mm] = fma4(p.rl[(size_t)(rbase + r) * D + m0 + mm], o, accv[mm]);
                                else accv[mm] = fma4(ld4(p.X_in + ((size_t)j * D + m0 + mm) * F + col), o, accv[mm]);
                            }
                        }
                    }
                    __syncthreads();
                }
                if (wave == 0) {
                    const int col = colb + c0l;
                    const int nrow = b == 0 ? 1 : 2 * l + 1;
#pragma unroll
                    for (int mm = 0; mm < 9; ++mm) {
                        accv[mm].x += __shfl_xor(accv[mm].x, 32, 64);
                        accv[mm].y += __shfl_xor(accv[mm].y, 32, 64);
                        accv[mm].z += __shfl_xor(accv[mm].z, 32, 64);
                        accv[mm].w += __shfl_xor(accv[mm].w, 32, 64);
                        if (half || mm >= nrow) continue;
                        if (role == 0) {
                            st4(p.h_out + (size_t)i0 * F + col, ld4(p.h_in + (size_t)i0 * F + col) + accv[mm]);
                        } else {
                            const size_t off = ((size_t)i0 * D + m0 + mm) * F + col;
                            if (role == 1) st4(p.X_out + off, first ? accv[mm] : ld4(p.X_in + off) + accv[mm]);
                            else st4(p.X_out + off, ld4(p.X_out + off) + accv[mm]);
                        }
                    }
                }
            }
        }
}

}  // namespace gn

// ====================================================================================== C ABI
extern "C" long gn_edge_tiles_cap(int N, long E) {
    if (N < 0 || E < 0) return 0;
    const long nchunks = (N + gn::FT_CHUNK - 1) / gn::FT_CHUNK;
    return 2 * (E / gn::FT_ROWS + 1) + N / gn::FT_MAXT + nchunks + 2;
}

extern "C" int gn_edge_tiles(const int* rowptr, int N, int cap, int* tile_first, int* n_tiles, void* stream) {
    if (N < 0 || cap < 1 || !rowptr || !tile_first || !n_tiles) return GN_ERR_BAD_ARG;
    const int nchunks = (N + gn::FT_CHUNK - 1) / gn::FT_CHUNK;
    if ((size_t)nchunks * sizeof(int) > 60000) return GN_ERR_BAD_ARG;      // 15 k chunks = 3.9 M atoms per call
    hipLaunchKernelGGL(gn::edge_tiles_kernel, dim3(1), dim3(1024), (size_t)(nchunks > 0 ? nchunks : 1) * sizeof(int),
                       (hipStream_t)stream, rowptr, N, cap, tile_first, n_tiles);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_message_fused_supported(int F, int H, int lmax, int M, int act, int arith) {
    if (F < 128 || F > 1024 || (F % 128) || !gn::is_pow2(F)) return 0;
    if (H <= 0 || H > gn::FT_MAXH || !gn::is_pow2(H) || (F % H) || ((F / H) % 4) || (F / H) > 128) return 0;
    if (lmax < 1 || lmax > 4 || act != GN_ACT_SILU) return 0;
    if (M < 3 || (M * F) % H || ((M * F) / H) % 4) return 0;
    return arith == 1 || arith == 2;
}

extern "C" int gn_message_fused(const gn_fused_desc* d, int arith, void* stream) {
    if (!d) return GN_ERR_BAD_ARG;
    const int M = 1 + (d->sep_dir ? d->lmax : 1) + (d->sep_tensor ? d->lmax : 1);
    if (!gn_message_fused_supported(d->F, d->H, d->lmax, M, GN_ACT_SILU, arith) || d->N < 0 || d->tile_cap < 1 ||
        (d->ldqk & 3) || (d->ldxv & 3) || !d->t || !d->W || !d->tile_first || !d->n_tiles || !d->attn_ws ||
        d->X_in == d->X_out)
        return GN_ERR_BAD_ARG;
    if (d->N == 0) return GN_OK;
    gn::FusedArgs p{d->t, d->W, d->bias, d->q, d->k, d->ldqk, d->x, d->v, d->ldxv, d->X_in, d->h_in, d->h_out, d->X_out,
                    d->rl, d->cut, d->rowptr, d->src, d->outdeg, d->tile_first, d->n_tiles, d->attn_ws,
                    d->N, d->F, d->H, d->lmax, M, d->sep_dir ? 1 : 0, d->sep_tensor ? 1 : 0,
                    (float)(1.0 / sqrt((double)d->F))};
    const dim3 grid(8 * (unsigned)((d->tile_cap + 7) / 8)), block(256);
    if (arith == 2) hipLaunchKernelGGL(gn::message_fused_kernel<2>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gn::message_fused_kernel<1>, grid, block, 0, (hipStream_t)stream, p);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
