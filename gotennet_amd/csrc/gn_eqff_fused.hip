// gn_eqff_fused.hip -- the node-local EQFF chains as ONE kernel each (forward and input-gradient).
//
// Reference: EQFF.forward, gotennet.py:716-748 --  X_p = X W_vu^T (stays a projection launch: N D rows),
//   n = sqrt(sum_D X_p^2 + eps);  [m1 | m2] = W_1 SiLU(W_0 [h | n] + b_0) + b_1;  h += m1;  X += m2 * X_p.
// Round 3 ran everything after X_p as context kernel -> gamma_m.0 (riding in an edge-sized launch) -> gamma_m.1 (a 15 us
// launch of 168 tiles) -> update kernel, and the backward as bwd_a -> W_1^T (rider) -> W_0^T (15 us) -> bwd_b: four
// dependent launches each way whose cost is launch + one tile latency, plus three extra passes over [N, D, F] tables.
// Here a workgroup owns 8 atoms end to end:
//   prologue (element-wise, fills the [8 x 2F] operand tile)  ->  product 1 ([8 x 2F] x [2F -> F], epilogue functor)
//   ->  product 2 ([8 x F] x [F -> 2F])  ->  epilogue (element-wise, writes the [8, D, F] rows once).
// The two products are MFMA (v_mfma_f32_32x32x16_f16 / _bf16, the rows of the A fragment past the tile repeat its rows) in the
// plane arithmetics of gn_gemm: A tile as fp16 planes with ONE exponent per tile (forward: [h | n], backward: [g_h | g_n]:
// a few decades at most) or exact bf16 triples; fragment-major weight planes L2 -> registers.  The exact-fp32 arithmetic,
// other activations and F outside {128, 256} keep the launch sequence (gn_eqff_fused_supported).
#include "gn_gemm.h"

namespace gn {

constexpr int EQ_ATOMS = 8;          // atoms per workgroup: 336 workgroups at C2, two or more per CU (the chain is latency-bound)

struct EqffArgs {
    // forward: h, X in/out; backward: gh, gX in
    const float* Xp;
    const void* W0; const float* b0;      // product 1: [F out, 2F in] planes (+ bias)
    const void* W1; const float* b1;      // product 2: [2F out, F in] planes (+ bias)
    float eps;
    int N, F, D;
    float* h; float* X;                    // forward in/out
    float* ctx_out; float* pre_out; float* mm_out;     // forward: kept for the backward (or NULL)
    const float* gh; const float* gX; const float* mm; const float* ctx; const float* pre_g1;   // backward inputs
    float* gXp; float* gh1;                // backward outputs
};

template <int MODE>
struct Planes {
    static constexpr int NP = MODE == 2 ? 2 : 3;
};

// fp32 tile [16][K] (LDS, pitch ldt) -> operand planes [NP][16][K + 8] (16-bit).  MODE 2: one exponent for the tile.
template <int MODE>
__device__ __forceinline__ int tile_to_planes(const float* T, int ldt, int K, unsigned short* P, float* red, int tid) {
    constexpr int NP = Planes<MODE>::NP;
    const int pitch = K + 8, plane = EQ_ATOMS * pitch;
    int e = 0;
    float scale = 1.f;
    if constexpr (MODE == 2) {
        float m = 0.f;
        for (int idx = tid; idx < EQ_ATOMS * K / 4; idx += 256) {
            const int r = idx / (K / 4), c = (idx % (K / 4)) * 4;
            const float4 v = ld4(T + r * ldt + c);
            const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
            // (an Inf / NaN must not poison the exponent: its own row turns non-finite, nothing else)
            m = fmaxf(m, fmaxf(fmaxf(a0 <= 3.0e38f ? a0 : 0.f, a1 <= 3.0e38f ? a1 : 0.f),
                               fmaxf(a2 <= 3.0e38f ? a2 : 0.f, a3 <= 3.0e38f ? a3 : 0.f)));
        }
        m = wave_max(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126 - 15;       // |x| < 2^(e + 15)
        e = e < -120 ? -120 : e;
        scale = __uint_as_float((unsigned)(127 - e) << 23);
        __syncthreads();
    }
    for (int idx = tid; idx < EQ_ATOMS * K / 4; idx += 256) {
        const int r = idx / (K / 4), c = (idx % (K / 4)) * 4;
        const float4 v = ld4(T + r * ldt + c);
        unsigned short* d = P + r * pitch + c;
        if constexpr (MODE == 2) {
            f16x4 hi, lo;
            split4_f16(v, scale, hi, lo);
            *reinterpret_cast<f16x4*>(d) = hi;
            *reinterpret_cast<f16x4*>(d + plane) = lo;
        } else {
            bf16x4 hi, mid, lo;
            split4_trunc(v, hi, mid, lo);
            *reinterpret_cast<bf16x4*>(d) = hi;
            *reinterpret_cast<bf16x4*>(d + plane) = mid;
            *reinterpret_cast<bf16x4*>(d + 2 * plane) = lo;
        }
    }
    __syncthreads();
    return e;
}

// [16 x K] (planes P in LDS) x W^T (fragment-major planes, [Nout][K]) -> fp32 tile Out[16][Nout] (LDS, pitch ldo), bias
// added.  Wave w owns the 32-column blocks w, w + 4, ... (at most NTW of them).
template <int MODE, int NTW>
__device__ __forceinline__ void chain_product(const unsigned short* P, int K, const void* Wp, const float* bias, int Nout,
                                              int e_a, float* Out, int ldo, int tid) {
    constexpr int NP = Planes<MODE>::NP;
    constexpr bool F16 = MODE == 2;
    const int lane = tid & 63, wave = tid >> 6;
    const int pitch = K + 8, plane = EQ_ATOMS * pitch;
    const int nk = K / 16, ks2 = 2 * ((K + BK - 1) / BK);
    const int ntiles = Nout / 32, ntw = (ntiles - wave + 3) / 4;       // blocks of this wave
    const uint4* wfrag = reinterpret_cast<const uint4*>(Wp) + (F16 ? 16 : 0);
    const int ewt = F16 ? __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(Wp)) : 0;
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    size_t off[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = j < ntw ? wave + 4 * j : wave;                   // idle slots re-read a valid block (never stored)
        off[j] = (size_t)nt * ks2 * (NP * 64) + lane;
    }
    const unsigned short* Ap = P + (lane & (EQ_ATOMS - 1)) * pitch + (lane >> 5) * 8;
    // CH k-steps of weight fragments per register set, TWO sets: while one set is multiplied the other is in flight.  (With
    // one step of look-ahead every k-step waited a full L2 round trip for 6 MFMAs of work, 14 us for the 32 steps of
    // product 1; one set of 2 CH steps, loaded and then multiplied, still paid that round trip once per set.)
    constexpr int CH = NP == 2 ? (NTW <= 2 ? 4 : 2) : (NTW <= 2 ? 2 : 1);
    uint4 bq0[CH][NTW][NP], bq1[CH][NTW][NP];
    auto load_b = [&](int g, uint4 (&q)[NTW][NP]) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int s = 0; s < NP; ++s) q[j][s] = wfrag[off[j] + (size_t)(g * NP + s) * 64];
    };
    auto kstep = [&](int g, const uint4 (&bw)[NTW][NP]) {
        if constexpr (F16) {
            f16x8 a[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = *reinterpret_cast<const f16x8*>(Ap + s * plane + g * 16);
            constexpr int TA[3] = {1, 0, 0};
            constexpr int TB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TA[t]], __builtin_bit_cast(f16x8, bw[j][TB[t]]), acc[j], 0, 0, 0);
        } else {
            bf16x8 a[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) a[s] = *reinterpret_cast<const bf16x8*>(Ap + s * plane + g * 16);
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], __builtin_bit_cast(bf16x8, bw[j][TB[t]]), acc[j], 0, 0, 0);
        }
    };
#pragma unroll
    for (int c = 0; c < CH; ++c) load_b(c, bq0[c]);
    for (int g0 = 0; g0 < nk; g0 += 2 * CH) {        // nk = K / 16 is a multiple of 8 (K % 128 == 0), 2 CH divides 8
#pragma unroll
        for (int c = 0; c < CH; ++c) load_b(g0 + CH + c, bq1[c]);
        __builtin_amdgcn_sched_barrier(0);           // a set's loads together: left alone the scheduler interleaves them with
#pragma unroll                                       // the MFMAs three at a time and every group waits on L2
        for (int c = 0; c < CH; ++c) kstep(g0 + c, bq0[c]);
        __builtin_amdgcn_sched_barrier(0);
        if (g0 + 2 * CH < nk) {
#pragma unroll
            for (int c = 0; c < CH; ++c) load_b(g0 + 2 * CH + c, bq0[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CH; ++c) kstep(g0 + CH + c, bq1[c]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // rows 0..EQ_ATOMS-1 of the 32 x 32 tile: lane (l >> 5) holds rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (j >= ntw) continue;
        const int col = (wave + 4 * j) * 32 + (lane & 31);
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < EQ_ATOMS / 2; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[j][r];
            if constexpr (F16) v = ldexpf(v, e_a + ewt);
            Out[row * ldo + col] = v + bv;
        }
    }
    __syncthreads();
}

// BWD = false: EQFF forward after X_p; BWD = true: its input-gradient
template <int MODE, bool BWD, int FT>
__global__ __launch_bounds__(256) void eqff_fused_kernel(const EqffArgs p) {
    constexpr int NP = Planes<MODE>::NP;
    __shared__ __attribute__((aligned(16))) float T[EQ_ATOMS * (2 * FT + 4)];              // fp32 tile [8][2F + 4]
    __shared__ __attribute__((aligned(16))) unsigned short P0[NP * EQ_ATOMS * (2 * FT + 8)];     // operand planes
    __shared__ float red[4];
    unsigned short* const P1 = P0;                   // the [16 x F] operand of product 2 reuses them (product 1 is done)
    constexpr int F = FT, K0 = 2 * F;
    constexpr int ldt = K0 + 4;
    const int D = p.D;
    const int tid = threadIdx.x;
    const int a0 = blockIdx.x * EQ_ATOMS;
    constexpr int DU = 8;                            // rows of the [D, F] tables in flight per atom
    constexpr int f4 = F >> 2, apr = 256 / f4;       // float4 column groups per row, atoms per pass of the workgroup
    const int c0 = (tid % f4) * 4, ar = tid / f4;

    // ---- prologue: the [16 x 2F] operand tile
#pragma unroll
    for (int a = ar; a < EQ_ATOMS; a += apr) {
        const int n = a0 + a;
        float4 lo4 = zero4(), hi4 = zero4();
        if (n < p.N) {
            if constexpr (!BWD) {                    // [h | sqrt(sum_D X_p^2 + eps)]          (gotennet.py:731-735)
                float4 s = zero4();
                for (int m0 = 0; m0 < D; m0 += DU) {     // DU rows in flight (D is a run-time value: a plain loop
                    float4 q[DU];                        // compiles to one exposed HBM round trip per row)
#pragma unroll
                    for (int u = 0; u < DU; ++u) q[u] = ld4(p.Xp + ((size_t)n * D + (m0 + u < D ? m0 + u : D - 1)) * F + c0);
#pragma unroll
                    for (int u = 0; u < DU; ++u)
                        if (m0 + u < D) s = fma4(q[u], q[u], s);
                }
                lo4 = ld4(p.h + (size_t)n * F + c0);
                hi4 = make_float4(sqrtf(s.x + p.eps), sqrtf(s.y + p.eps), sqrtf(s.z + p.eps), sqrtf(s.w + p.eps));
                if (p.ctx_out) {
                    st4(p.ctx_out + (size_t)n * K0 + c0, lo4);
                    st4(p.ctx_out + (size_t)n * K0 + F + c0, hi4);
                }
            } else {                                 // g_m = [g_h | sum_D g_X X_p]
                float4 s = zero4();
                for (int m0 = 0; m0 < D; m0 += DU) {
                    float4 g[DU], q[DU];
#pragma unroll
                    for (int u = 0; u < DU; ++u) {
                        const size_t off = ((size_t)n * D + (m0 + u < D ? m0 + u : D - 1)) * F + c0;
                        g[u] = ld4(p.gX + off);
                        q[u] = ld4(p.Xp + off);
                    }
#pragma unroll
                    for (int u = 0; u < DU; ++u)
                        if (m0 + u < D) s = fma4(g[u], q[u], s);
                }
                lo4 = ld4(p.gh + (size_t)n * F + c0);
                hi4 = s;
            }
        }
        st4(T + a * ldt + c0, lo4);
        st4(T + a * ldt + F + c0, hi4);
    }
    __syncthreads();
    const int e0 = tile_to_planes<MODE>(T, ldt, K0, P0, red, tid);
    // ---- product 1: [16 x 2F] -> [16 x F]
    chain_product<MODE, 2>(P0, K0, p.W0, p.b0, F, e0, T, ldt, tid);
#pragma unroll
    for (int a = ar; a < EQ_ATOMS; a += apr) {
        const int n = a0 + a;
        float4 v = ld4(T + a * ldt + c0);
        if constexpr (!BWD) {                        // hidden = SiLU(pre); the backward keeps the pre-activation
            if (n < p.N && p.pre_out) st4(p.pre_out + (size_t)n * F + c0, v);
            v = act4(v, GN_ACT_SILU);
        } else {                                     // g_pre = (g_m W_1) * SiLU'(pre)
            const float4 pre = n < p.N ? ld4(p.pre_g1 + (size_t)n * F + c0) : zero4();
            v = v * dact4(pre, GN_ACT_SILU);
        }
        st4(T + a * ldt + c0, v);
    }
    __syncthreads();
    const int e1 = tile_to_planes<MODE>(T, ldt, F, P1, red, tid);
    // ---- product 2: [16 x F] -> [16 x 2F]
    chain_product<MODE, 4>(P1, F, p.W1, p.b1, K0, e1, T, ldt, tid);
    // ---- epilogue: every row load of the thread's atoms first, then the stores (a store to X / g_Xp may alias the
    // next atom's loads as far as the compiler knows: atom by atom the row loads of atom 2 waited for atom 1's stores)
    constexpr int NA = EQ_ATOMS / apr;               // atoms per thread
    for (int m0 = 0; m0 < D; m0 += DU) {
        float4 ra[NA][DU], rb[NA][DU];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int n = a0 + ar + k * apr < p.N ? a0 + ar + k * apr : p.N - 1;
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const size_t off = ((size_t)n * D + (m0 + u < D ? m0 + u : D - 1)) * F + c0;
                ra[k][u] = ld4((BWD ? p.gX : p.X) + off);
                rb[k][u] = ld4(p.Xp + off);
            }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int a = ar + k * apr, n = a0 + a;
            if (n >= p.N) continue;
            const float4 u2 = ld4(T + a * ldt + F + c0);
            if constexpr (!BWD) {                    // X += m2 * X_p                             (gotennet.py:745-746)
#pragma unroll
                for (int u = 0; u < DU; ++u)
                    if (m0 + u < D) st4(p.X + ((size_t)n * D + m0 + u) * F + c0, fma4(u2, rb[k][u], ra[k][u]));
            } else {                                 // g_Xp = g_X m2 + (g_n / n) X_p
                const float4 nn = ld4(p.ctx + (size_t)n * K0 + F + c0);
                const float4 sc = make_float4(u2.x / nn.x, u2.y / nn.y, u2.z / nn.z, u2.w / nn.w);
                const float4 m2 = ld4(p.mm + (size_t)n * K0 + F + c0);
#pragma unroll
                for (int u = 0; u < DU; ++u)
                    if (m0 + u < D) st4(p.gXp + ((size_t)n * D + m0 + u) * F + c0, fma4(sc, rb[k][u], ra[k][u] * m2));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int a = ar + k * apr, n = a0 + a;
        if (n >= p.N) continue;
        const float4 u1 = ld4(T + a * ldt + c0);
        if constexpr (!BWD) {                        // h += m1
            if (p.mm_out) {
                st4(p.mm_out + (size_t)n * K0 + c0, u1);
                st4(p.mm_out + (size_t)n * K0 + F + c0, ld4(T + a * ldt + F + c0));
            }
            float* hp = p.h + (size_t)n * F + c0;
            st4(hp, ld4(hp) + u1);
        } else {                                     // g_h1 = g_h + g_ctx[:, :F]
            st4(p.gh1 + (size_t)n * F + c0, ld4(p.gh + (size_t)n * F + c0) + u1);
        }
    }
}

}  // namespace gn

extern "C" int gn_eqff_fused_supported(int F, int act, int arith) {
    return (F == 128 || F == 256) && act == GN_ACT_SILU && (arith == 1 || arith == 2);
}

template <bool BWD>
static int eqff_fused_launch(const gn::EqffArgs& p, int arith, hipStream_t st) {
    const dim3 grid((unsigned)((p.N + gn::EQ_ATOMS - 1) / gn::EQ_ATOMS)), block(256);
    if (p.F == 256) {
        if (arith == 2) hipLaunchKernelGGL((gn::eqff_fused_kernel<2, BWD, 256>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gn::eqff_fused_kernel<1, BWD, 256>), grid, block, 0, st, p);
    } else {
        if (arith == 2) hipLaunchKernelGGL((gn::eqff_fused_kernel<2, BWD, 128>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gn::eqff_fused_kernel<1, BWD, 128>), grid, block, 0, st, p);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_eqff_fused_forward(const float* Xp, const void* W0p, const float* b0, const void* W1p, const float* b1,
                                     float eps, int N, int F, int D, float* h, float* X,
                                     float* ctx_out, float* pre_out, float* mm_out, int arith, void* stream) {
    if (!gn_eqff_fused_supported(F, GN_ACT_SILU, arith) || N < 0 || D <= 0 || !Xp || !W0p || !W1p || !h || !X)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    gn::EqffArgs p{};
    p.Xp = Xp; p.W0 = W0p; p.b0 = b0; p.W1 = W1p; p.b1 = b1; p.eps = eps; p.N = N; p.F = F; p.D = D;
    p.h = h; p.X = X; p.ctx_out = ctx_out; p.pre_out = pre_out; p.mm_out = mm_out;
    return eqff_fused_launch<false>(p, arith, (hipStream_t)stream);
}

extern "C" int gn_eqff_fused_backward(const float* gh, const float* gX, const float* mm, const float* Xp, const float* ctx,
                                      const float* pre_g1, const void* W1Tp, const void* W0Tp, int N, int F, int D,
                                      float* gXp, float* gh1, int arith, void* stream) {
    if (!gn_eqff_fused_supported(F, GN_ACT_SILU, arith) || N < 0 || D <= 0 || !gh || !gX || !mm || !Xp || !ctx || !pre_g1 ||
        !W1Tp || !W0Tp || !gXp || !gh1 || gXp == gX)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    gn::EqffArgs p{};
    p.Xp = Xp; p.W0 = W1Tp; p.b0 = nullptr; p.W1 = W0Tp; p.b1 = nullptr; p.eps = 0.f; p.N = N; p.F = F; p.D = D;
    p.gh = gh; p.gX = gX; p.mm = mm; p.ctx = ctx; p.pre_g1 = pre_g1; p.gXp = gXp; p.gh1 = gh1;
    return eqff_fused_launch<true>(p, arith, (hipStream_t)stream);
}
