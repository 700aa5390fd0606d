// gn_backward.hip -- hand-written input-gradient kernels for the force path
// F = -dE/dpos (reference: torch.autograd.grad through GotenNet.forward,
// outputs.py:365-375).  Only gradients w.r.t. activations are needed (no weight
// gradients), so the backward costs about one forward of GEMMs plus two gather
// passes per stage: a by-target pass over the CSR rows (gradients of target-side
// operands q_i, EQ_i and of per-edge quantities) and a by-source pass over the CSC
// columns (gradients of the gathered source rows x_j, v_j, k_j, X_j, EK_j).  Both
// recompute the per-edge message from saved pre-activations instead of storing
// [E, D, F] tensors (the reference's autograd OOMs at 62 GB on this workload).
// Same slot layout as the forward (gn_edge.hip); every sum is a fixed-order
// register/LDS reduction: no float atomics, bit-reproducible forces.
//
// Forward equations being differentiated: gotennet.py:452-559 (message),
// 351-364 + 561-611 (HTR), 716-748 (EQFF), layers.py:1658-1714 (init),
// layers.py:133-152, 744-746, 805-902 (cutoff, RBF, harmonics).
#include "gn_common.h"
#include "gn_tune.h"
#include "gn_sh.h"
#include "gn_highl.h"

namespace gn {



// The register-tiled kernels of this file are compiled for SiLU (the reference's default activation, and what every
// benchmarked configuration uses): with the activation kind as a run-time switch the other eleven kinds' code cost
// 4..22 VGPRs and a spill in msg_bwd_source (+4.5 % on the whole step).  Models with another activation take the
// degree-sliced kernels of gn_highl.hip, which carry the kind as a run-time value.
// =========================================================================== HTR backward
// w = sum_l [ A.B - (2 - r.r)(A.r)(B.r) ],  A = EQ_i block, B = EK_j block, r = rl block
template <int LMAX, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_HTR_TGT) void htr_bwd_target_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w,
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F_rt,
    float* __restrict__ gEQ, float* __restrict__ g_rl, float* __restrict__ g_pre_t, int /* act: GN_ACT_SILU on this path, see gn_htr_backward */) {
    constexpr int act = GN_ACT_SILU;
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int KP = D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : 32));
    constexpr int CH = D < 9 ? D : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int F = FC ? FC : F_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 eq[D], acc[D];
#pragma unroll
    for (int m = 0; m < D; ++m) { eq[m] = ld4(EQ + ((size_t)i * D + m) * F + c0); acc[m] = zero4(); }
    for (int e = e0 + slot; e < e1; e += ns) {
        const float4 gte = ld4(gtp + (size_t)e * F + c0), pte = ld4(pre_t + (size_t)e * F + c0);
        float4 a_pt, d_pt;
        act_pair4(pte, act, a_pt, d_pt);
        const float4 gw = gte * a_pt;
        // t' = t + SiLU(pre_t) * w:  d/d pre_t, ready for the plain W_t^T product that follows
        // (w is read here and nowhere else after K7 wrote it: non-temporal; step 7.555 / 7.534 -> 7.532 / 7.513 ms.  A non-temporal
        //  g_pre_t store measured nothing: the W_t^T product reads it next)
        st4(g_pre_t + (size_t)e * F + c0, gte * ld4_nt(w + (size_t)e * F + c0) * d_pt);
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float part[KP];
#pragma unroll
        for (int m = D; m < KP; ++m) part[m] = 0.f;
        int m0 = 0;
#pragma unroll
        for (int l = 1; l <= LMAX; ++l) {
            float4 ek[2 * LMAX + 1];
            float r[2 * LMAX + 1];
            float4 pa = zero4(), pb = zero4();
            float rr = 0.f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                ek[mm] = ld4(kj + (size_t)(m0 + mm) * F);
                r[mm] = re[m0 + mm];
                pa = fma4(r[mm], eq[m0 + mm], pa);
                pb = fma4(r[mm], ek[mm], pb);
                rr = fmaf(r[mm], r[mm], rr);
            }
            // gw factored out of the per-row terms: u = -c gw pb, v = -c gw pa, s = 2 gw pa pb
            const float c = 2.0f - rr;
            const float4 u = gw * pb * (-c), v = gw * pa * (-c), s2 = gw * (pa * pb) * 2.0f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                acc[m0 + mm] = fma4(r[mm], u, fma4(gw, ek[mm], acc[m0 + mm]));
                part[m0 + mm] = hsum4(fma4(r[mm], s2, fma4(v, ek[mm], u * eq[m0 + mm])));
            }
            m0 += 2 * l + 1;
        }
        if (lps >= KP) {                             // one value-halving butterfly for all D sums
            multi_group_sum<KP>(part, lps, lp);
            const int stride = lps / KP;
            if ((lp & (stride - 1)) == 0 && lp / stride < D) g_rl[(size_t)e * D + lp / stride] = part[0];
        } else {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float s = group_sum(part[m], lps);
                if (lp == 0) g_rl[(size_t)e * D + m] = s;
            }
        }
    }
    reduce_rows<D>(acc, red, slot, c0, F, ns, [&](int row, float4 s) { st4(gEQ + ((size_t)i * D + row) * F + c0, s); });
}

template <int LMAX, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_HTR_SRC) void htr_bwd_source_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t,
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ colptr, const int* __restrict__ perm, const int* __restrict__ dst, int N, int F_rt,
    float* __restrict__ gEK, int /* act: GN_ACT_SILU on this path, see gn_htr_backward */) {
    constexpr int act = GN_ACT_SILU;
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int CH = D < 9 ? D : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int F = FC ? FC : F_rt;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int p0 = colptr[j], p1 = colptr[j + 1];
    float4 ek[D], acc[D];
#pragma unroll
    for (int m = 0; m < D; ++m) { ek[m] = ld4(EK + ((size_t)j * D + m) * F + c0); acc[m] = zero4(); }
    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = perm[pp];
        const float4 gw = ld4(gtp + (size_t)e * F + c0) * act4(ld4(pre_t + (size_t)e * F + c0), act);
        const float* qi = EQ + (size_t)dst[pp] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        int m0 = 0;
#pragma unroll
        for (int l = 1; l <= LMAX; ++l) {
            float4 eq[2 * LMAX + 1];
            float r[2 * LMAX + 1];
            float4 pa = zero4();
            float rr = 0.f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                eq[mm] = ld4(qi + (size_t)(m0 + mm) * F);
                r[mm] = re[m0 + mm];
                pa = fma4(r[mm], eq[mm], pa);
                rr = fmaf(r[mm], r[mm], rr);
            }
            const float4 v = gw * pa * (rr - 2.0f);
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm)
                acc[m0 + mm] = fma4(r[mm], v, fma4(gw, eq[mm], acc[m0 + mm]));
            m0 += 2 * l + 1;
        }
    }
    reduce_rows<D>(acc, red, slot, c0, F, ns, [&](int row, float4 s) { st4(gEK + ((size_t)j * D + row) * F + c0, s); });
}

// =========================================================================== message backward
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR>
struct MsgShape {
    static constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    static constexpr int ND = SEP_DIR ? LMAX : 1;
    static constexpr int NT = SEP_TENSOR ? LMAX : 1;
    static constexpr int M = 1 + ND + NT;
    // degrees [lo, hi] served by gate block b (b >= 1)
    __host__ __device__ static constexpr bool is_dir(int b) { return b >= 1 && b < 1 + ND; }
    __host__ __device__ static constexpr int lo(int b) { return is_dir(b) ? (SEP_DIR ? b : 1) : (SEP_TENSOR ? b - ND : 1); }
    __host__ __device__ static constexpr int hi(int b) { return is_dir(b) ? (SEP_DIR ? b : LMAX) : (SEP_TENSOR ? b - ND : LMAX); }
    __host__ __device__ static constexpr int first_row(int l) { return l * l - 1; }     // first m of degree l
};


// by-target pass: g_tf, g_cut, g_rl, attention backward (g_a -> g_s), g_ta, g_q
// (body in a forceinline function with __restrict__ parameters: with the pointers read from the argument struct the
//  compiler had to assume that the g_eproj stores alias every later load and issued the row loads one at a time)
// GS_LDS: the head gradients g_a -> g_s of this target stay in LDS between the three phases (deg * H <= GS_CAP floats)
// and reach the global g_s rows (read by the by-source pass) in ONE coalesced copy; otherwise they go through those rows
// like in round 2 (three dependent global round trips per workgroup).  Same arithmetic, same order in both forms.
constexpr int GS_CAP = 2048;
// FIRST: X_in is identically zero (first interaction): every tensor-gate term vanishes -- those blocks of eproj / x / v
// are not read, their g_eproj columns not written (the W_e^T product that follows takes the K-prefix).
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, bool GS_LDS, bool FIRST, int FC>
__device__ __forceinline__ void msg_bwd_target_body(const MsgBwdArgs& p, float* gsl, float* red, float* hsum, const float* __restrict__ x_, const float* __restrict__ v_, const float* __restrict__ eproj_, const float* __restrict__ a_, const float* __restrict__ qk_, const float* __restrict__ X_in_, const float* __restrict__ rl_, const float* __restrict__ cut_, const int* __restrict__ outdeg_, const float* __restrict__ g_h1_, const float* __restrict__ g_X1_, const int* __restrict__ rowptr_, const int* __restrict__ src_, float* __restrict__ g_eproj_, float* __restrict__ g_s_, float* __restrict__ g_nproj_, float* __restrict__ g_rl_, float* __restrict__ g_cut_) {
    using S = MsgShape<LMAX, SEP_DIR, SEP_TENSOR>;
    constexpr int D = S::D, M = S::M;
    constexpr int KP = D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : 32));      // D rl sums, padded to a power of two
    const int N = p.N, F = FC ? FC : p.F, H = p.H;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = rowptr_[i], e1 = rowptr_[i + 1];
    const int per_head = (M * F) / H;
    int hb[M];
#pragma unroll
    for (int b = 0; b < M; ++b) hb[b] = (b * F + c0) / per_head;
    auto GS = [&](int e, int h) -> float& {
        if constexpr (GS_LDS) return gsl[(e - e0) * H + h];
        else return g_s_[(size_t)e * H + h];
    };

    const float4 gdh = ld4(g_h1_ + (size_t)i * F + c0);
    float4 gdX[D];
#pragma unroll
    for (int m = 0; m < D; ++m) gdX[m] = ld4(g_X1_ + ((size_t)i * D + m) * F + c0);

    // ---- phase 1: per-edge gate gradients
    for (int e = e0 + slot; e < e1; e += ns) {
        const int j = src_[e];
        const float ce = cut_[e];
        const float* xr = x_ + (size_t)j * p.ldxv + c0;
        const float* vr = v_ + (size_t)j * p.ldxv + c0;
        const float* tr = eproj_ + (size_t)e * p.lde + F + c0;
        float* gtr = g_eproj_ + (size_t)e * p.lde + F + c0;
        const float* ar = a_ + (size_t)e * H;
        const float* Xj = X_in_ + (size_t)j * D * F + c0;
        const float* re = rl_ + (size_t)e * D;
        float pa_h[M];
        float cutp = 0.f;
        float rlp[D];
#pragma unroll
        for (int b = 0; b < M; ++b) {
            if (FIRST && b > 0 && !S::is_dir(b)) { pa_h[b] = 0.f; continue; }
            float4 go;
            if (b == 0) {
                go = gdh;
            } else {
                go = zero4();
#pragma unroll
                for (int l = S::lo(b); l <= S::hi(b); ++l)
#pragma unroll
                    for (int m = S::first_row(l); m < S::first_row(l) + 2 * l + 1; ++m)
                        go = S::is_dir(b) ? fma4(re[m], gdX[m], go) : fma4(gdX[m], ld4(Xj + (size_t)m * F), go);
            }
            // (t_filter with an ORDINARY load here: the by-source pass re-reads these rows right after this launch, and with
            //  the non-temporal hint none of them survived in L2 / the Infinity Cache: 303.8 -> 296.3 us per layer, round 4)
            const float4 tfb = ld4(tr + b * F), xb = ld4(xr + b * F), vb = ld4(vr + b * F);
            st4_nt(gtr + b * F, (go * xb) * ce);
            cutp += hsum4(go * tfb * xb);
            pa_h[b] = hsum4(go * vb);
            if (S::is_dir(b)) {
                const float4 od = fma4(ar[hb[b]], vb, (tfb * xb) * ce);      // forward direction gate
#pragma unroll
                for (int l = S::lo(b); l <= S::hi(b); ++l)
#pragma unroll
                    for (int m = S::first_row(l); m < S::first_row(l) + 2 * l + 1; ++m) rlp[m] = hsum4(gdX[m] * od);
            }
        }
        cutp = group_sum(cutp, lps);
        if (lp == 0) g_cut_[e] = cutp;
        // head sums: head h owns the flattened float4 positions [h PH, (h+1) PH) of the (block, lane) grid,
        // PH = M lps / H.  Stage the M per-lane partials in LDS (a slot never spans waves: wave-ordered LDS
        // accesses, no barrier), lps / H reader lanes per head add M consecutive entries each, then a short
        // butterfly.  (Per-head selects cost 8 M compare/select pairs per edge and 77 spilled mask SGPRs.)
        {
            float* hrow = hsum + slot * (M * lps);
#pragma unroll
            for (int b = 0; b < M; ++b) hrow[b * lps + lp] = pa_h[b];
            const int rpl = lps / H, hh = lp / rpl, part = lp - hh * rpl;
            const float* hp = hrow + hh * (M * rpl) + part * M;
            float hv = hp[0];
#pragma unroll
            for (int k = 1; k < M; ++k) hv += hp[k];
            hv = group_sum(hv, rpl);
            if (part == 0) GS(e, hh) = hv;
        }
        if (lps >= KP) {                             // D rl sums in one butterfly
            float vals[KP];
#pragma unroll
            for (int m = 0; m < KP; ++m) vals[m] = m < D ? rlp[m] : 0.f;
            multi_group_sum<KP>(vals, lps, lp);
            const int stride = lps / KP;
            if ((lp & (stride - 1)) == 0 && lp / stride < D) g_rl_[(size_t)e * D + lp / stride] = vals[0];
        } else {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float s = group_sum(rlp[m], lps);
                if (lp == 0) g_rl_[(size_t)e * D + m] = s;
            }
        }
    }
    // phase 3's rows of this slot's first PF3 edges are requested HERE, ahead of the two barriers of phase 2: they do not
    // depend on it, and fetched inside phase 3 each trip exposes one HBM round trip (t_attn row) with two loads in flight
    constexpr int PF3 = GN_MSGB_PF3;
    float4 kj_pf[PF3 ? PF3 : 1], pta_pf[PF3 ? PF3 : 1];
    if (PF3 && e1 > e0) {
#pragma unroll
        for (int u = 0; u < PF3; ++u) {
            const int e = e0 + slot + u * ns, ec = e < e1 ? e : e1 - 1;
            kj_pf[u] = ld4(qk_ + (size_t)src_[ec] * p.ldqk + F + c0);
            pta_pf[u] = ld4_nt(eproj_ + (size_t)ec * p.lde + c0);
        }
    }
    __syncthreads();
    // ---- phase 2: softmax backward per head:  g_s = a g_a - (a / nrm) sum_e' a g_a
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int h = wave; h < H; h += 4) {
            float dot = 0.f;
            for (int e = e0 + lane; e < e1; e += 64) dot += a_[(size_t)e * H + h] * GS(e, h);
            dot = wave_sum(dot);
            for (int e = e0 + lane; e < e1; e += 64) {
                const float nrm = outdeg_ ? sqrtf((float)outdeg_[src_[e]]) * p.inv_sqrt_f : p.inv_sqrt_f;
                const float av = a_[(size_t)e * H + h];
                GS(e, h) = av * GS(e, h) - (av / nrm) * dot;
            }
        }
    }
    __syncthreads();
    // ---- phase 3: scores backward: g_ta (pre-SiLU' factor), g_q
    const int hq = c0 / (F / H);
    const float4 qi = ld4(qk_ + (size_t)i * p.ldqk + c0);
    float4 gq = zero4();
    if constexpr (GS_LDS) {
        const int n = (e1 - e0) * H;
        for (int idx = threadIdx.x; idx < n; idx += 256) g_s_[(size_t)e0 * H + idx] = gsl[idx];
    }
    auto score_bwd = [&](int e, float4 kj, float4 pta) {
        const float gs = GS(e, hq);
        float4 a_ta, d_ta;
        act_pair4(pta, GN_ACT_SILU, a_ta, d_ta);
        gq = fma4(gs, kj * a_ta, gq);
        st4_nt(g_eproj_ + (size_t)e * p.lde + c0, ((qi * kj) * gs) * d_ta);   // d/d(pre-activation of t_attn)
    };
#pragma unroll
    for (int u = 0; u < PF3; ++u) {
        const int e = e0 + slot + u * ns;
        if (e < e1) score_bwd(e, kj_pf[u], pta_pf[u]);
    }
    for (int e = e0 + slot + PF3 * ns; e < e1; e += ns)
        score_bwd(e, ld4(qk_ + (size_t)src_[e] * p.ldqk + F + c0), ld4_nt(eproj_ + (size_t)e * p.lde + c0));
    st4(&red[slot * F + c0], gq);
    __syncthreads();
    if (slot == 0) st4(g_nproj_ + (size_t)i * p.ldn + c0, red4(red, c0, F, ns));
}

template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, bool FIRST = false, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_TGT) void msg_bwd_target_kernel(const MsgBwdArgs p) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    __shared__ float hsum[256 * MsgShape<LMAX, SEP_DIR, SEP_TENSOR>::M];
    __shared__ float gsl[GS_CAP];
    const int i = xcd_item(blockIdx.x, p.N);
    if (i < 0) return;
    if ((p.rowptr[i + 1] - p.rowptr[i]) * p.H <= GS_CAP)      // workgroup-uniform, decided once
        msg_bwd_target_body<LMAX, SEP_DIR, SEP_TENSOR, true, FIRST, FC>(p, gsl, red, hsum, p.x, p.v, p.eproj, p.a, p.qk, p.X_in, p.rl, p.cut, p.outdeg, p.g_h1, p.g_X1, p.rowptr, p.src, p.g_eproj, p.g_s, p.g_nproj, p.g_rl, p.g_cut);
    else
        msg_bwd_target_body<LMAX, SEP_DIR, SEP_TENSOR, false, FIRST, FC>(p, gsl, red, hsum, p.x, p.v, p.eproj, p.a, p.qk, p.X_in, p.rl, p.cut, p.outdeg, p.g_h1, p.g_X1, p.rowptr, p.src, p.g_eproj, p.g_s, p.g_nproj, p.g_rl, p.g_cut);
}

// by-source pass: g_x, g_v, g_k, and g_X (tensor-gate path) of the gathered source rows
// FIRST (X_in == 0): no g_X rows (nothing consumes the gradient of a constant), tensor-gate blocks of g_x / g_v are zero
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, bool FIRST = false, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_SRC) void msg_bwd_source_kernel(const MsgBwdArgs p) {
    using S = MsgShape<LMAX, SEP_DIR, SEP_TENSOR>;
    constexpr int D = S::D, M = S::M;
    constexpr int XD = FIRST ? 0 : D;                 // g_X rows carried
    constexpr int ROWS = 2 * M + XD + 1;
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int N = p.N, F = FC ? FC : p.F, H = p.H;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int p0 = p.colptr[j], p1 = p.colptr[j + 1];
    const int per_head = (M * F) / H;
    const int hq = c0 / (F / H);
    int hb[M];
#pragma unroll
    for (int b = 0; b < M; ++b) hb[b] = (b * F + c0) / per_head;

    // rows: [0,M) g_x, [M,2M) g_v, [2M, 2M+XD) g_X, 2M+XD g_k
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();

    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = p.perm[pp];
        const int i = p.dst[pp];
        const float ce = p.cut[e];
        // own rows re-read per edge (L1-resident) instead of pinned in registers
        const float* xr = p.x + (size_t)j * p.ldxv + c0;
        const float* vr = p.v + (size_t)j * p.ldxv + c0;
        const float* Xj = p.X_in + (size_t)j * D * F + c0;
        asm volatile("" : "+v"(xr), "+v"(vr), "+v"(Xj));
        const float* tr = p.eproj + (size_t)e * p.lde + F + c0;
        const float* ar = p.a + (size_t)e * H;
        const float* re = p.rl + (size_t)e * D;
        const float* gXi = p.g_X1 + (size_t)i * D * F + c0;
#pragma unroll
        for (int b = 0; b < M; ++b) {
            if (FIRST && b > 0 && !S::is_dir(b)) continue;
            const float4 tfb = ld4_nt(tr + b * F);
            const float ab = ar[hb[b]];
            float4 go;
            if (b == 0) {
                go = ld4(p.g_h1 + (size_t)i * F + c0);
            } else {
                go = zero4();
                float4 ot = zero4();
                if (!S::is_dir(b)) ot = fma4(ab, ld4(vr + b * F), (tfb * ld4(xr + b * F)) * ce);   // forward tensor gate
#pragma unroll
                for (int l = S::lo(b); l <= S::hi(b); ++l)
#pragma unroll
                    for (int m = S::first_row(l); m < S::first_row(l) + 2 * l + 1; ++m) {
                        const float4 gx = ld4(gXi + (size_t)m * F);
                        if (S::is_dir(b)) {
                            go = fma4(re[m], gx, go);
                        } else if constexpr (!FIRST) {
                            go = fma4(gx, ld4(Xj + (size_t)m * F), go);
                            acc[2 * M + m] = fma4(gx, ot, acc[2 * M + m]);
                        }
                    }
            }
            acc[b] = fma4(go, tfb * ce, acc[b]);
            acc[M + b] = fma4(ab, go, acc[M + b]);
        }
        const float gs = p.g_s[(size_t)e * H + hq];
        const float4 qi = ld4(p.qk + (size_t)i * p.ldqk + c0);
        const float4 ta = act4(ld4_nt(p.eproj + (size_t)e * p.lde + c0), GN_ACT_SILU);
        acc[2 * M + XD] = fma4(gs, qi * ta, acc[2 * M + XD]);
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns, [&](int row, float4 s) {
        if (row < M) st4(p.g_x + (size_t)j * p.ldxv + row * F + c0, s);
        else if (row < 2 * M) st4(p.g_v + (size_t)j * p.ldxv + (row - M) * F + c0, s);
        else if (row < 2 * M + XD) {
            const size_t off = ((size_t)j * D + (row - 2 * M)) * F + c0;
            st4(p.g_X_out + off, ld4(p.g_X1 + off) + s);
        } else st4(p.g_nproj + (size_t)j * p.ldn + F + c0, s);
    });
}

// by-source pass that ALSO does the per-edge work of the by-target pass (round 5): the gate gradients `go` of an edge are
// computed once, here, and everything that needs them -- g_tf, g_cut, g_rl, the head sums of g_a (per edge) and g_x, g_v,
// g_X (per source) -- leaves this kernel; t_filter is read ONCE (the two-kernel form reads it in both passes: 1.50 x the
// algorithmic traffic).  The softmax / scores backward then runs by target (attn_bwd_kernel, as in the degree-group form)
// and g_k by source (msg_bwd_gk_kernel).  General launches (X_in != NULL), SiLU.
// Where the operands of an edge come from and when they are asked for (second form of the round; the first re-read the own
// rows through L1 per edge and fenced every block's loads behind the previous block's store: 252 -> 195 us per launch):
//  * the source's own rows x_j, v_j, X_j (2M + D rows, loop invariants that do not fit in registers next to the 2M + D
//    accumulators) are staged in LDS once per workgroup and read with ds_read_b128: the first form re-read them through the
//    vector-memory path on every edge -- 18 of the 33 16-byte row loads of an edge at lmax 2, i.e. more than half of the
//    L1 / texture-addresser time of the kernel (64 B per clock per CU: 38 KiB per edge = 61 us of the launch);
//  * the pointers are __restrict__ parameters of a forceinline body, and all rows of the edge (D rows of g_X1[i], M rows of
//    t_filter, g_h1[i]) are requested at the top of the trip: the first form had them in the argument struct, so every
//    g_eproj store fenced the next block's loads -- five dependent round trips per edge;
//  * the (edge, target) indices of the NEXT trip are requested while this one computes.
// FIRST: X_in is identically zero (first interaction): only the scalar and the direction-gate blocks exist -- the tensor-gate
// blocks of eproj / x / v are not read, their g_eproj columns not written, those blocks of g_x / g_v are written as zeros,
// no X rows, no g_X (as in the FIRST forms of the by-target / by-source pair, which this replaces when ga is given).
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int FC, bool FIRST>
__device__ __forceinline__ void msg_bwd_merged_body(
    const float* __restrict__ x, const float* __restrict__ v, int ldxv, const float* __restrict__ X_in,
    const float* __restrict__ eproj, int lde, const float* __restrict__ a, const float* __restrict__ rl,
    const float* __restrict__ cut, const float* __restrict__ g_h1, const float* __restrict__ g_X1,
    const int* __restrict__ dst, const int* __restrict__ colptr, const int* __restrict__ perm,
    float* __restrict__ g_eproj, float* __restrict__ g_x, float* __restrict__ g_v, float* __restrict__ g_X_out,
    float* __restrict__ g_rl, float* __restrict__ g_cut, float* __restrict__ ga, int N, int F_rt, int H) {
    using S = MsgShape<LMAX, SEP_DIR, SEP_TENSOR>;
    constexpr int D = S::D, M = S::M;
    constexpr int KP = D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : 32));
    constexpr int MB = FIRST ? 1 + S::ND : M;         // value blocks this launch handles
    constexpr int XR = FIRST ? 0 : D;                 // X rows / g_X rows
    constexpr int ROWS = 2 * MB + XR;                 // [0,MB) g_x, [MB,2MB) g_v, [2MB, 2MB+D) g_X
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    __shared__ __attribute__((aligned(16))) float own[ROWS * 256];     // x_j | v_j | X_j rows of this source
    __shared__ float hsum[256 * M];
    const int F = FC ? FC : F_rt;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int p0 = colptr[j], p1 = colptr[j + 1];
    for (int idx = threadIdx.x * 4; idx < ROWS * F; idx += 1024) {
        const int r = idx / F, c = idx - r * F;
        const float* g = r < MB ? x + (size_t)j * ldxv + r * F : (r < 2 * MB ? v + (size_t)j * ldxv + (r - MB) * F
                                                                               : X_in + ((size_t)j * D + (r - 2 * MB)) * F);
        st4(own + idx, ld4(g + c));
    }
    const int per_head = (M * F) / H;
    int hb[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) hb[b] = (b * F + c0) / per_head;
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();
    if (FIRST) {
#pragma unroll
        for (int b = MB; b < M; ++b) hsum[slot * (M * lps) + b * lps + lp] = 0.f;
    }
    __syncthreads();

    int pp = p0 + slot;
    int e_n = 0, i_n = 0;
    if (pp < p1) { e_n = perm[pp]; i_n = dst[pp]; }
    for (; pp < p1; pp += ns) {
        const int e = e_n, i = i_n;
        if (pp + ns < p1) { e_n = perm[pp + ns]; i_n = dst[pp + ns]; }
        const float* tr = eproj + (size_t)e * lde + F + c0;
        float* gtr = g_eproj + (size_t)e * lde + F + c0;
        const float* ar = a + (size_t)e * H;
        const float* re = rl + (size_t)e * D;
        const float* gXi = g_X1 + (size_t)i * D * F + c0;
        int oc = c0;
        asm volatile("" : "+v"(oc));                  // own rows: read from LDS per edge, not hoisted into 72 registers
        const float* ox = own + oc;
        const float* ov = own + MB * F + oc;
        const float* oX = own + 2 * MB * F + oc;
        // every global row of the edge, before the first store
        float4 tf[MB], gx[D];
#pragma unroll
        for (int b = 0; b < MB; ++b) tf[b] = ld4_nt(tr + b * F);
#pragma unroll
        for (int m = 0; m < D; ++m) gx[m] = ld4(gXi + (size_t)m * F);
        const float4 gh = ld4(g_h1 + (size_t)i * F + c0);
        const float ce = cut[e];
        float ab_[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) ab_[b] = ar[hb[b]];
        float pa_h[MB], rlp[D];
        float cutp = 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const float4 tfb = tf[b], xb = ld4(ox + b * F), vb = ld4(ov + b * F);
            const float ab = ab_[b];
            const float4 fw = fma4(ab, vb, (tfb * xb) * ce);       // the forward gate of this block
            float4 go;
            if (b == 0) {
                go = gh;
            } else {
                go = zero4();
#pragma unroll
                for (int l = S::lo(b); l <= S::hi(b); ++l)
#pragma unroll
                    for (int m = S::first_row(l); m < S::first_row(l) + 2 * l + 1; ++m) {
                        if (S::is_dir(b)) {
                            go = fma4(re[m], gx[m], go);
                            rlp[m] = hsum4(gx[m] * fw);
                        } else {
                            go = fma4(gx[m], ld4(oX + (size_t)m * F), go);
                            acc[2 * MB + m] = fma4(gx[m], fw, acc[2 * MB + m]);
                        }
                    }
            }
            st4_nt(gtr + b * F, (go * xb) * ce);
            cutp += hsum4(go * tfb * xb);
            pa_h[b] = hsum4(go * vb);
            acc[b] = fma4(go, tfb * ce, acc[b]);
            acc[MB + b] = fma4(ab, go, acc[MB + b]);
        }
        cutp = group_sum(cutp, lps);
        if (lp == 0) g_cut[e] = cutp;
        {   // head sums of g_a (same staging as msg_bwd_target_body: a slot never spans waves, wave-ordered LDS accesses)
            float* hrow = hsum + slot * (M * lps);  // (heads are cut from ALL M blocks of the value vector: FIRST leaves the rest zero)
#pragma unroll
            for (int b = 0; b < MB; ++b) hrow[b * lps + lp] = pa_h[b];
            const int rpl = lps / H, hh = lp / rpl, part = lp - hh * rpl;
            const float* hp = hrow + hh * (M * rpl) + part * M;
            float hv = hp[0];
#pragma unroll
            for (int k = 1; k < M; ++k) hv += hp[k];
            hv = group_sum(hv, rpl);
            if (part == 0) ga[(size_t)e * H + hh] = hv;
        }
        if (lps >= KP) {                             // D rl sums in one butterfly
            float vals[KP];
#pragma unroll
            for (int m = 0; m < KP; ++m) vals[m] = m < D ? rlp[m] : 0.f;
            multi_group_sum<KP>(vals, lps, lp);
            const int stride = lps / KP;
            if ((lp & (stride - 1)) == 0 && lp / stride < D) g_rl[(size_t)e * D + lp / stride] = vals[0];
        } else {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float sv = group_sum(rlp[m], lps);
                if (lp == 0) g_rl[(size_t)e * D + m] = sv;
            }
        }
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns, [&](int row, float4 sv) {
        if (row < MB) st4(g_x + (size_t)j * ldxv + row * F + c0, sv);
        else if (row < 2 * MB) st4(g_v + (size_t)j * ldxv + (row - MB) * F + c0, sv);
        else {
            const size_t off = ((size_t)j * D + (row - 2 * MB)) * F + c0;
            st4(g_X_out + off, ld4(g_X1 + off) + sv);
        }
    });
    if (FIRST)                                       // the tensor-gate blocks of g_x / g_v: zeros (their operands never reach a message)
        for (int b = MB + slot; b < M; b += ns) {
            st4(g_x + (size_t)j * ldxv + b * F + c0, zero4());
            st4(g_v + (size_t)j * ldxv + b * F + c0, zero4());
        }
}
// (occupancy hints per form, gn_tune.h: the general launch runs best without one, the first-interaction form at 2 waves per SIMD)
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_MRG) void msg_bwd_merged_kernel(const MsgBwdArgs p, float* __restrict__ ga) {

    msg_bwd_merged_body<LMAX, SEP_DIR, SEP_TENSOR, FC, false>(p.x, p.v, p.ldxv, p.X_in, p.eproj, p.lde, p.a, p.rl, p.cut, p.g_h1, p.g_X1,
                                                        p.dst, p.colptr, p.perm, p.g_eproj, p.g_x, p.g_v, p.g_X_out, p.g_rl,
                                                        p.g_cut, ga, p.N, p.F, p.H);
}
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_MRG_F) void msg_bwd_merged_first_kernel(const MsgBwdArgs p, float* __restrict__ ga) {

    msg_bwd_merged_body<LMAX, SEP_DIR, SEP_TENSOR, FC, true>(p.x, p.v, p.ldxv, p.X_in, p.eproj, p.lde, p.a, p.rl, p.cut, p.g_h1, p.g_X1,
                                                        p.dst, p.colptr, p.perm, p.g_eproj, p.g_x, p.g_v, p.g_X_out, p.g_rl,
                                                        p.g_cut, ga, p.N, p.F, p.H);
}

// g_k of the gathered source rows, after the softmax backward:  g_k_j = sum_e g_s[e, head] q_i SiLU(t_attn pre-activation)
template <int FC = 0>
__global__ __launch_bounds__(256) void msg_bwd_gk_kernel(const MsgBwdArgs p) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int N = p.N, F = FC ? FC : p.F, H = p.H;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int hq = c0 / (F / H);
    float4 gk[1] = {zero4()};
    for (int pp = p.colptr[j] + slot; pp < p.colptr[j + 1]; pp += ns) {
        const int e = p.perm[pp], i = p.dst[pp];
        const float gs = p.g_s[(size_t)e * H + hq];
        const float4 qi = ld4(p.qk + (size_t)i * p.ldqk + c0);
        const float4 ta = act4(ld4_nt(p.eproj + (size_t)e * p.lde + c0), GN_ACT_SILU);
        gk[0] = fma4(gs, qi * ta, gk[0]);
    }
    reduce_rows<1>(gk, red, slot, c0, F, ns, [&](int, float4 sv) { st4(p.g_nproj + (size_t)j * p.ldn + F + c0, sv); });
}

// =========================================================================== degree-grouped backward (lmax >= 3)
// For lmax >= 3 with sep_dir and sep_tensor the monolithic kernels above need 250+ VGPRs.  Gates are
// per degree, so the work is cut into degree groups {scalar,1,2}, {3}, {4}: one by-source launch per
// group (msg_bwd_merged_group_kernel below: the group's per-edge work rides in it), one attention-
// backward launch (softmax backward needs the head sums of ALL groups) and one for g_k.
// Value blocks: 0 scalar, l direction gate, LMAX + l tensor gate.
// softmax backward over the summed head gradients of all groups, then scores backward (g_ta, g_q)
template <bool GS_LDS, int FC>
__device__ __forceinline__ void attn_bwd_body(const MsgBwdArgs& p, const float* __restrict__ ga_parts, int G, size_t gstride,
                                              float* red, float* gsl, int i) {
    const int F = FC ? FC : p.F, H = p.H;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = p.rowptr[i], e1 = p.rowptr[i + 1];
    auto GS = [&](int e, int h) -> float& {
        if constexpr (GS_LDS) return gsl[(e - e0) * H + h];
        else return p.g_s[(size_t)e * H + h];
    };
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int h = wave; h < H; h += 4) {
            float dot = 0.f;
            for (int e = e0 + lane; e < e1; e += 64) {
                float ga = 0.f;
                for (int q = 0; q < G; ++q) ga += ga_parts[q * gstride + (size_t)e * H + h];
                GS(e, h) = ga;
                dot += p.a[(size_t)e * H + h] * ga;
            }
            dot = wave_sum(dot);
            for (int e = e0 + lane; e < e1; e += 64) {
                const float nrm = p.outdeg ? sqrtf((float)p.outdeg[p.src[e]]) * p.inv_sqrt_f : p.inv_sqrt_f;
                const float av = p.a[(size_t)e * H + h];
                GS(e, h) = av * GS(e, h) - (av / nrm) * dot;
            }
        }
    }
    __syncthreads();
    if constexpr (GS_LDS) {
        const int n = (e1 - e0) * H;
        for (int idx = threadIdx.x; idx < n; idx += 256) p.g_s[(size_t)e0 * H + idx] = gsl[idx];
    }
    const int hq = c0 / (F / H);
    const float4 qi = ld4(p.qk + (size_t)i * p.ldqk + c0);
    float4 gq = zero4();
    for (int e = e0 + slot; e < e1; e += ns) {
        const float gs = GS(e, hq);
        const float4 kj = ld4(p.qk + (size_t)p.src[e] * p.ldqk + F + c0);
        const float4 pta = ld4_nt(p.eproj + (size_t)e * p.lde + c0);
        float4 a_ta, d_ta;
        act_pair4(pta, GN_ACT_SILU, a_ta, d_ta);
        gq = fma4(gs, kj * a_ta, gq);
        st4_nt(p.g_eproj + (size_t)e * p.lde + c0, ((qi * kj) * gs) * d_ta);
    }
    st4(&red[slot * F + c0], gq);
    __syncthreads();
    if (slot == 0) st4(p.g_nproj + (size_t)i * p.ldn + c0, red4(red, c0, F, ns));
}
template <int FC = 0>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const MsgBwdArgs p, const float* __restrict__ ga_parts, int G, size_t gstride) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    __shared__ float gsl[GS_CAP];
    const int i = xcd_item(blockIdx.x, p.N);
    if (i < 0) return;
    if ((p.rowptr[i + 1] - p.rowptr[i]) * p.H <= GS_CAP) attn_bwd_body<true, FC>(p, ga_parts, G, gstride, red, gsl, i);
    else attn_bwd_body<false, FC>(p, ga_parts, G, gstride, red, gsl, i);
}

// degree-group form of msg_bwd_merged_kernel: the by-source pass of a group with the group's per-edge work merged in
// (g_tf of its blocks, its cut slice, its g_rl rows, its slice of the head sums); g_k is left to msg_bwd_gk_kernel.
// Like msg_bwd_merged_body: the group's own rows (its x / v blocks, its X rows) in LDS, every global row of the edge
// requested before the first store, next trip's indices prefetched (lmax 4 step 13.15 -> 12.14 ms, C5 14.9 -> 13.83).
template <int LMAX, int LLO, int LHI, bool SCALAR, int FC>
__device__ __forceinline__ void msg_bwd_merged_group_body(
    const float* __restrict__ x, const float* __restrict__ v, int ldxv, const float* __restrict__ X_in,
    const float* __restrict__ eproj, int lde, const float* __restrict__ a, const float* __restrict__ rl,
    const float* __restrict__ cut, const float* __restrict__ g_h1, const float* __restrict__ g_X1,
    const int* __restrict__ dst, const int* __restrict__ colptr, const int* __restrict__ perm,
    float* __restrict__ g_eproj, float* __restrict__ g_x, float* __restrict__ g_v, float* __restrict__ g_X_out,
    float* __restrict__ g_rl, float* __restrict__ ga_slice, float* __restrict__ cut_slice, int N, int F_rt, int H) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int M = 1 + 2 * LMAX;
    constexpr int NL = LHI - LLO + 1;
    constexpr int XR = (LHI + 1) * (LHI + 1) - LLO * LLO, M0 = LLO * LLO - 1;
    constexpr int NB = (SCALAR ? 1 : 0) + 2 * NL;            // value blocks of this group
    constexpr int ROWS = 2 * NB + XR;                        // g_x, g_v per block, g_X rows
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    constexpr int KP = (XR + 8) <= 16 ? 16 : 32;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    __shared__ __attribute__((aligned(16))) float own[ROWS * 256];     // x blocks | v blocks | X rows of this source (group order)
    const int F = FC ? FC : F_rt;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int p0 = colptr[j], p1 = colptr[j + 1];
    const int per_head = (M * F) / H;
    auto vblock = [&](int k) { return SCALAR ? (k == 0 ? 0 : (k <= NL ? LLO + k - 1 : LMAX + LLO + k - 1 - NL))
                                             : (k < NL ? LLO + k : LMAX + LLO + k - NL); };
    for (int idx = threadIdx.x * 4; idx < ROWS * F; idx += 1024) {
        const int r = idx / F, c = idx - r * F;
        const float* g = r < NB ? x + (size_t)j * ldxv + vblock(r) * F
                                : (r < 2 * NB ? v + (size_t)j * ldxv + vblock(r - NB) * F
                                              : X_in + ((size_t)j * D + M0 + (r - 2 * NB)) * F);
        st4(own + idx, ld4(g + c));
    }
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();
    __syncthreads();

    int pp = p0 + slot;
    int e_n = 0, i_n = 0;
    if (pp < p1) { e_n = perm[pp]; i_n = dst[pp]; }
    for (; pp < p1; pp += ns) {
        const int e = e_n, i = i_n;
        if (pp + ns < p1) { e_n = perm[pp + ns]; i_n = dst[pp + ns]; }
        const float* tr = eproj + (size_t)e * lde + F + c0;
        float* gtr = g_eproj + (size_t)e * lde + F + c0;
        const float* ar = a + (size_t)e * H;
        const float* re = rl + (size_t)e * D;
        const float* gXi = g_X1 + (size_t)i * D * F + c0;
        int oc = c0;
        asm volatile("" : "+v"(oc));                  // own rows: read from LDS per edge, not hoisted into registers
        const float* ox = own + oc;
        const float* ov = own + NB * F + oc;
        const float* oX = own + 2 * NB * F + oc;
        // every global row of the edge, before the first store
        float4 tf[NB], gxa[XR];
        float ab_[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) tf[k] = ld4_nt(tr + vblock(k) * F);
#pragma unroll
        for (int m = 0; m < XR; ++m) gxa[m] = ld4(gXi + (size_t)(M0 + m) * F);
        float4 gh = zero4();
        if (SCALAR) gh = ld4(g_h1 + (size_t)i * F + c0);
        const float ce = cut[e];
#pragma unroll
        for (int k = 0; k < NB; ++k) ab_[k] = ar[(vblock(k) * F + c0) / per_head];
        float cutp = 0.f;
        float vals[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) vals[k] = 0.f;
        // one value block: gradient `go` of its gate -> g_tf, cut partial, head partial, g_x / g_v rows; returns the forward gate
        auto block = [&](int b, int k, float4 go) {
            const float4 tfb = tf[k], xb = ld4(ox + k * F), vb = ld4(ov + k * F);
            const int hb = (b * F + c0) / per_head;
            const float ab = ab_[k];
            st4_nt(gtr + b * F, (go * xb) * ce);
            cutp += hsum4(go * tfb * xb);
            const float pa = hsum4(go * vb);
#pragma unroll
            for (int h = 0; h < 8; ++h) vals[XR + h] += (hb == h) ? pa : 0.f;
            acc[k] = fma4(go, tfb * ce, acc[k]);
            acc[NB + k] = fma4(ab, go, acc[NB + k]);
            return fma4(ab, vb, (tfb * xb) * ce);
        };
        if (SCALAR) block(0, 0, gh);
#pragma unroll
        for (int l = LLO; l <= LHI; ++l) {
            const int kd = (SCALAR ? 1 : 0) + (l - LLO), kt = kd + NL;
            float4 god = zero4(), got = zero4();
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                const int m = l * l - 1 + mm;
                god = fma4(re[m], gxa[m - M0], god);
                got = fma4(gxa[m - M0], ld4(oX + (size_t)(m - M0) * F), got);
            }
            const float4 od = block(l, kd, god);
            const float4 ot = block(LMAX + l, kt, got);
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                const int m = l * l - 1 + mm;
                vals[m - M0] = hsum4(gxa[m - M0] * od);
                acc[2 * NB + m - M0] = fma4(gxa[m - M0], ot, acc[2 * NB + m - M0]);
            }
        }
        cutp = group_sum(cutp, lps);
        if (lp == 0) cut_slice[e] = cutp;
        if (H <= 8 && lps >= KP) {
            multi_group_sum<KP>(vals, lps, lp);
            const int stride = lps / KP;
            if ((lp & (stride - 1)) == 0) {
                const int idx = lp / stride;
                if (idx < XR) g_rl[(size_t)e * D + M0 + idx] = vals[0];
                else if (idx - XR < H) ga_slice[(size_t)e * H + idx - XR] = vals[0];
            }
        } else {
#pragma unroll
            for (int k = 0; k < XR + 8; ++k) {
                const float sv = group_sum(vals[k], lps);
                if (lp == 0) {
                    if (k < XR) g_rl[(size_t)e * D + M0 + k] = sv;
                    else if (k - XR < H) ga_slice[(size_t)e * H + k - XR] = sv;
                }
            }
        }
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns, [&](int row, float4 sv) {
        if (row < NB) st4(g_x + (size_t)j * ldxv + vblock(row) * F + c0, sv);
        else if (row < 2 * NB) st4(g_v + (size_t)j * ldxv + vblock(row - NB) * F + c0, sv);
        else {
            const size_t off = ((size_t)j * D + M0 + (row - 2 * NB)) * F + c0;
            st4(g_X_out + off, ld4(g_X1 + off) + sv);
        }
    });
}
template <int LMAX, int LLO, int LHI, bool SCALAR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_MRG_G) void msg_bwd_merged_group_kernel(const MsgBwdArgs p, float* __restrict__ ga_slice,
                                                                                        float* __restrict__ cut_slice) {
    msg_bwd_merged_group_body<LMAX, LLO, LHI, SCALAR, FC>(p.x, p.v, p.ldxv, p.X_in, p.eproj, p.lde, p.a, p.rl, p.cut, p.g_h1,
                                                           p.g_X1, p.dst, p.colptr, p.perm, p.g_eproj, p.g_x, p.g_v, p.g_X_out,
                                                           p.g_rl, ga_slice, cut_slice, p.N, p.F, p.H);
}
// the {3} group (155 VGPRs) under its own occupancy hint
template <int LMAX, int LLO, int LHI, bool SCALAR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_MSG_MRG_G3) void msg_bwd_merged_group3_kernel(const MsgBwdArgs p, float* __restrict__ ga_slice,
                                                                                        float* __restrict__ cut_slice) {
    msg_bwd_merged_group_body<LMAX, LLO, LHI, SCALAR, FC>(p.x, p.v, p.ldxv, p.X_in, p.eproj, p.lde, p.a, p.rl, p.cut, p.g_h1,
                                                           p.g_X1, p.dst, p.colptr, p.perm, p.g_eproj, p.g_x, p.g_v, p.g_X_out,
                                                           p.g_rl, ga_slice, cut_slice, p.N, p.F, p.H);
}

// HTR backward per degree group (w = sum_l w_l: the degrees are independent)
template <int LMAX, int LLO, int LHI, bool FIRST, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_HTR_TGT_G) void htr_bwd_target_group_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w,
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F_rt,
    float* __restrict__ gEQ, float* __restrict__ g_rl, float* __restrict__ g_pre_t, int /* act: GN_ACT_SILU on this path, see gn_htr_backward */) {
    constexpr int act = GN_ACT_SILU;
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int XR = (LHI + 1) * (LHI + 1) - LLO * LLO, M0 = LLO * LLO - 1;
    constexpr int KP = XR <= 4 ? 4 : (XR <= 8 ? 8 : 16);
    constexpr int CH = XR < 9 ? XR : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int F = FC ? FC : F_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    float4 eq[XR], acc[XR];
#pragma unroll
    for (int m = 0; m < XR; ++m) { eq[m] = ld4(EQ + ((size_t)i * D + M0 + m) * F + c0); acc[m] = zero4(); }
    for (int e = rowptr[i] + slot; e < rowptr[i + 1]; e += ns) {
        const float4 gte = ld4(gtp + (size_t)e * F + c0), pte = ld4(pre_t + (size_t)e * F + c0);
        float4 a_pt, d_pt;
        act_pair4(pte, act, a_pt, d_pt);
        const float4 gw = gte * a_pt;
        if (FIRST) st4(g_pre_t + (size_t)e * F + c0, gte * ld4(w + (size_t)e * F + c0) * d_pt);
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float part[KP];
#pragma unroll
        for (int m = 0; m < KP; ++m) part[m] = 0.f;
#pragma unroll
        for (int l = LLO; l <= LHI; ++l) {
            const int b0 = l * l - 1 - M0;
            float4 ek[2 * LMAX + 1];
            float r[2 * LMAX + 1];
            float4 pa = zero4(), pb = zero4();
            float rr = 0.f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                ek[mm] = ld4(kj + (size_t)(M0 + b0 + mm) * F);
                r[mm] = re[M0 + b0 + mm];
                pa = fma4(r[mm], eq[b0 + mm], pa);
                pb = fma4(r[mm], ek[mm], pb);
                rr = fmaf(r[mm], r[mm], rr);
            }
            const float c = 2.0f - rr;
            const float4 u = gw * pb * (-c), v = gw * pa * (-c), s2 = gw * (pa * pb) * 2.0f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                acc[b0 + mm] = fma4(r[mm], u, fma4(gw, ek[mm], acc[b0 + mm]));
                part[b0 + mm] = hsum4(fma4(r[mm], s2, fma4(v, ek[mm], u * eq[b0 + mm])));
            }
        }
        if (lps >= KP) {
            multi_group_sum<KP>(part, lps, lp);
            const int stride = lps / KP;
            if ((lp & (stride - 1)) == 0 && lp / stride < XR) g_rl[(size_t)e * D + M0 + lp / stride] = part[0];
        } else {
#pragma unroll
            for (int m = 0; m < XR; ++m) {
                const float sv = group_sum(part[m], lps);
                if (lp == 0) g_rl[(size_t)e * D + M0 + m] = sv;
            }
        }
    }
    reduce_rows<XR>(acc, red, slot, c0, F, ns, [&](int row, float4 sv) { st4(gEQ + ((size_t)i * D + M0 + row) * F + c0, sv); });
}

template <int LMAX, int LLO, int LHI, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_HTR_SRC_G) void htr_bwd_source_group_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t,
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ colptr, const int* __restrict__ perm, const int* __restrict__ dst, int N, int F_rt,
    float* __restrict__ gEK, int /* act: GN_ACT_SILU on this path, see gn_htr_backward */) {
    constexpr int act = GN_ACT_SILU;
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int XR = (LHI + 1) * (LHI + 1) - LLO * LLO, M0 = LLO * LLO - 1;
    constexpr int CH = XR < 9 ? XR : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int F = FC ? FC : F_rt;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    GN_SLOT_GEOMETRY(FC);
    float4 acc[XR];
#pragma unroll
    for (int m = 0; m < XR; ++m) acc[m] = zero4();
    for (int pp = colptr[j] + slot; pp < colptr[j + 1]; pp += ns) {
        const int e = perm[pp];
        const float4 gw = ld4(gtp + (size_t)e * F + c0) * act4(ld4(pre_t + (size_t)e * F + c0), act);
        const float* qi = EQ + (size_t)dst[pp] * D * F + c0;
        const float* re = rl + (size_t)e * D;
#pragma unroll
        for (int l = LLO; l <= LHI; ++l) {
            const int b0 = l * l - 1 - M0;
            float4 eq[2 * LMAX + 1];
            float r[2 * LMAX + 1];
            float4 pa = zero4();
            float rr = 0.f;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                eq[mm] = ld4(qi + (size_t)(M0 + b0 + mm) * F);
                r[mm] = re[M0 + b0 + mm];
                pa = fma4(r[mm], eq[mm], pa);
                rr = fmaf(r[mm], r[mm], rr);
            }
            const float c = 2.0f - rr;
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) acc[b0 + mm] = fma4(gw, eq[mm] + pa * (-c * r[mm]), acc[b0 + mm]);
        }
    }
    reduce_rows<XR>(acc, red, slot, c0, F, ns, [&](int row, float4 sv) { st4(gEK + ((size_t)j * D + M0 + row) * F + c0, sv); });
}

// =========================================================================== EQFF backward
// part a: gm = [gh' | sum_m gX' Xp],  gXp = gX' * m2.  gX == NULL: dL/dX' is identically zero (the output layer of an
// energy head that reads h only): no zero-filled tensor is made or read
__global__ void eqff_bwd_a_kernel(const float* __restrict__ gh, const float* __restrict__ gX,
                                  const float* __restrict__ mm, const float* __restrict__ Xp,
                                  int N, int F, int D, float* __restrict__ gm, float* __restrict__ gXp) {
    const int f4 = F >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * f4) return;
    const int n = (int)(idx / f4), c0 = (int)(idx % f4) * 4;
    const float4 m2 = ld4(mm + (size_t)n * 2 * F + F + c0);
    float4 s = zero4();
    for (int m = 0; m < D; ++m) {
        const size_t off = ((size_t)n * D + m) * F + c0;
        const float4 g = gX ? ld4(gX + off) : zero4();
        s = gX ? fma4(g, ld4(Xp + off), s) : s;
        st4(gXp + off, g * m2);
    }
    st4(gm + (size_t)n * 2 * F + c0, ld4(gh + (size_t)n * F + c0));
    st4(gm + (size_t)n * 2 * F + F + c0, s);
}

// part b: gXp += g_n * Xp / n ;  gh1 = gh' + g_ctx[:, :F]     (n = ctx[:, F:2F])
__global__ void eqff_bwd_b_kernel(const float* __restrict__ gctx, const float* __restrict__ ctx,
                                  const float* __restrict__ Xp, const float* __restrict__ gh,
                                  int N, int F, int D, float* __restrict__ gXp, float* __restrict__ gh1) {
    const int f4 = F >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * f4) return;
    const int n = (int)(idx / f4), c0 = (int)(idx % f4) * 4;
    const float4 gn = ld4(gctx + (size_t)n * 2 * F + F + c0);
    const float4 nn = ld4(ctx + (size_t)n * 2 * F + F + c0);
    const float4 sc = make_float4(gn.x / nn.x, gn.y / nn.y, gn.z / nn.z, gn.w / nn.w);
    for (int m = 0; m < D; ++m) {
        const size_t off = ((size_t)n * D + m) * F + c0;
        st4(gXp + off, fma4(sc, ld4(Xp + off), ld4(gXp + off)));
    }
    st4(gh1 + (size_t)n * F + c0, ld4(gh + (size_t)n * F + c0) + ld4(gctx + (size_t)n * 2 * F + c0));
}

// =========================================================================== init backward
// EdgeInit: t0[e] = (h_i + h_j) fe[e].  g_fe[e] = gt0 (h_i + h_j);  gh[n] += sum_{in(n)} gt0 fe + sum_{out(n)} gt0 fe
__global__ __launch_bounds__(256) void edge_init_bwd_kernel(
    const float* __restrict__ gt0, const float* __restrict__ h, const float* __restrict__ feat, int ldf,
    const int* __restrict__ rowptr, const int* __restrict__ src,
    const int* __restrict__ colptr, const int* __restrict__ perm,
    int N, int F, float* __restrict__ g_feat, float* __restrict__ gh) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(0);
    const float4 hi = ld4(h + (size_t)i * F + c0);
    float4 acc = zero4();
    for (int e = rowptr[i] + slot; e < rowptr[i + 1]; e += ns) {
        const float4 g = ld4(gt0 + (size_t)e * F + c0);
        const float4 fe = ld4(feat + (size_t)e * ldf + F + c0);
        st4(g_feat + (size_t)e * ldf + F + c0, g * (hi + ld4(h + (size_t)src[e] * F + c0)));
        acc = fma4(g, fe, acc);
    }
    for (int pp = colptr[i] + slot; pp < colptr[i + 1]; pp += ns) {
        const int e = perm[pp];
        acc = fma4(ld4(gt0 + (size_t)e * F + c0), ld4(feat + (size_t)e * ldf + F + c0), acc);
    }
    st4(&red[slot * F + c0], acc);
    __syncthreads();
    if (slot == 0) {
        float* o = gh + (size_t)i * F + c0;
        st4(o, ld4(o) + red4(red, c0, F, ns));
    }
}

// NodeInit: m_i = sum_{j != i} A_nbr[z_j] (fn[e] cut_e).  g_fn[e] = g_m[i] A_nbr[z_j] cut_e;  g_cut[e] += sum_f g_m A fn
__global__ __launch_bounds__(256) void node_init_bwd_kernel(
    const float* __restrict__ g_ctx, const int* __restrict__ z, const float* __restrict__ feat, int ldf,
    const float* __restrict__ cut, const float* __restrict__ A_nbr,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F,
    float* __restrict__ g_feat, float* __restrict__ g_cut) {
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(0);
    const float4 gmi = ld4(g_ctx + (size_t)i * 2 * F + F + c0);
    for (int e = rowptr[i] + slot; e < rowptr[i + 1]; e += ns) {
        const int j = src[e];
        float4 gfn = zero4();
        float gc = 0.f;
        if (j != i) {
            const float4 ga = gmi * ld4(A_nbr + (size_t)z[j] * F + c0);
            gfn = ga * cut[e];
            gc = hsum4(ga * ld4(feat + (size_t)e * ldf + c0));
        }
        st4(g_feat + (size_t)e * ldf + c0, gfn);
        slot_sum_store(gc, lps, lp, g_cut + e, (size_t)rowptr[N]);       // (F > 256: one partial slice per 64-lane part)
    }
}

// y = SiLU(LN(x) gamma + beta): gx = rstd (g_xh - mean(g_xh) - xh mean(g_xh xh)), g_xh = g_out SiLU'(v) gamma
template <bool SILU>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ gout, int N, int F, float* __restrict__ gx, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (size_t)row * F;
    const float* gr = gout + (size_t)row * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += xr[f];
    const float mean = wave_sum(s) / (float)F;
    float q = 0.f;
    for (int f = lane; f < F; f += 64) { const float d = xr[f] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)F + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float xh = (xr[f] - mean) * rstd;
        const float g = gr[f] * (SILU ? dact1(xh * gamma[f] + beta[f], act) : 1.0f) * gamma[f];
        s1 += g; s2 += g * xh;
    }
    s1 = wave_sum(s1) / (float)F; s2 = wave_sum(s2) / (float)F;
    for (int f = lane; f < F; f += 64) {
        const float xh = (xr[f] - mean) * rstd;
        const float g = gr[f] * (SILU ? dact1(xh * gamma[f] + beta[f], act) : 1.0f) * gamma[f];
        gx[(size_t)row * F + f] = rstd * (g - s1 - xh * s2);
    }
}

// =========================================================================== geometry backward
struct Dual3 {
    float v, d[3];
};
__host__ __device__ inline Dual3 operator+(Dual3 a, Dual3 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__host__ __device__ inline Dual3 operator-(Dual3 a, Dual3 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__host__ __device__ inline Dual3 operator-(Dual3 a) { return {-a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
__host__ __device__ inline Dual3 operator*(Dual3 a, Dual3 b) {
    return {a.v * b.v, {a.v * b.d[0] + a.d[0] * b.v, a.v * b.d[1] + a.d[1] * b.v, a.v * b.d[2] + a.d[2] * b.v}};
}
__host__ __device__ inline Dual3 operator*(float s, Dual3 a) { return {s * a.v, {s * a.d[0], s * a.d[1], s * a.d[2]}}; }
__host__ __device__ inline Dual3 operator*(Dual3 a, float s) { return s * a; }

// per edge: g_phi [R], g_cut, g_rl [D]  ->  g_vec [3] (through the unit vector) and g_diff (through the distance)
template <int LMAX>
__global__ __launch_bounds__(128) void edge_geometry_bwd_kernel(
    const float* __restrict__ vec, const float* __restrict__ dist, const int* __restrict__ src, const int* __restrict__ dst,
    int E, int R, int basis, const float* __restrict__ means, const float* __restrict__ betas, float cutoff, float alpha,
    const float* __restrict__ g_rl, int n_rl, const float* __restrict__ g_cut, int n_cut,
    const float* __restrict__ g_phi, float* __restrict__ g_vec, float* __restrict__ g_diff) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (src[e] == dst[e]) {          // self-loop: pos_i - pos_i and d = 0 are constants
        g_vec[3 * e] = g_vec[3 * e + 1] = g_vec[3 * e + 2] = 0.f;
        g_diff[e] = 0.f;
        return;
    }
    const float pi = 3.14159265358979323846f;
    const float d = dist[e];
    float gd = 0.f;
    if (basis == 1) {                                // d/dd [ sin(a d) / d ]   (d > 0 here: self-loops returned above)
        for (int r = 0; r < R; ++r) {
            const float a = means[r], ad = a * d;
            gd += g_phi[(size_t)e * R + r] * (a * cosf(ad) / d - sinf(ad) / (d * d));
        }
    } else if (basis == 2) {                         // d/dd exp(c (d - o)^2) = 2 c (d - o) exp(.)
        for (int r = 0; r < R; ++r) {
            const float w = betas[r], q = d - means[r], c2 = -0.5f / (w * w);
            gd += g_phi[(size_t)e * R + r] * expf(c2 * (q * q)) * (2.0f * c2 * q);
        }
    }
    if (d < cutoff) {
        const float arg = d * pi / cutoff;
        const float c = 0.5f * (cosf(arg) + 1.0f);
        const float dc = -0.5f * (pi / cutoff) * sinf(arg);
        float s = 0.f;
        if (basis == 0) {
            const float u = expf(alpha * (-d));
            // d/dd [ c G ] = dc G + c G (-2 beta w) (-alpha u)
            auto term = [&](int r, float gp) {
                const float w = u - means[r];
                const float G = expf(-betas[r] * (w * w));
                s += gp * G * (dc + c * (2.0f * betas[r] * w * alpha * u));
            };
            if ((R & 3) == 0) {                      // a lane's own row in 16-byte pieces (a quarter of the divergent loads)
                for (int r = 0; r < R; r += 4) {
                    const float4 gp = ld4(g_phi + (size_t)e * R + r);
                    term(r, gp.x); term(r + 1, gp.y); term(r + 2, gp.z); term(r + 3, gp.w);
                }
            } else {
                for (int r = 0; r < R; ++r) term(r, g_phi[(size_t)e * R + r]);
            }
        }
        float gc = 0.f;                              // fixed-order sum of the per-kernel slices (four loads in flight)
        int q = 0;
        for (; q + 4 <= n_cut; q += 4) {
            const float c0 = g_cut[(size_t)q * E + e], c1 = g_cut[(size_t)(q + 1) * E + e];
            const float c2 = g_cut[(size_t)(q + 2) * E + e], c3 = g_cut[(size_t)(q + 3) * E + e];
            gc += c0; gc += c1; gc += c2; gc += c3;
        }
        for (; q < n_cut; ++q) gc += g_cut[(size_t)q * E + e];
        gd += s + gc * dc;
    }
    g_diff[e] = gd;
    const float x = vec[3 * e], y = vec[3 * e + 1], z = vec[3 * e + 2];
    const float n = sqrtf(x * x + y * y + z * z);
    const float ux = x / n, uy = y / n, uz = z / n;
    Dual3 o[D];
    real_harmonics<LMAX, Dual3>(Dual3{ux, {1.f, 0.f, 0.f}}, Dual3{uy, {0.f, 1.f, 0.f}}, Dual3{uz, {0.f, 0.f, 1.f}}, o);
    // fixed-order sum of the per-kernel slices, a slice's D values of this edge per trip (they are contiguous: D = 8 is two
    // 16-byte loads).  With the slice loop innermost every one of the n_rl * D loads was its own dependent round trip:
    // 30 us at C2 and 21 us for ONE molecule.
    float gsum[D];
#pragma unroll
    for (int m = 0; m < D; ++m) gsum[m] = 0.f;
#pragma unroll 2
    for (int q = 0; q < n_rl; ++q) {
        const float* row = g_rl + ((size_t)q * E + e) * D;
        if constexpr (D % 4 == 0) {
#pragma unroll
            for (int m = 0; m < D; m += 4) {
                const float4 v = ld4(row + m);
                gsum[m] += v.x; gsum[m + 1] += v.y; gsum[m + 2] += v.z; gsum[m + 3] += v.w;
            }
        } else {
#pragma unroll
            for (int m = 0; m < D; ++m) gsum[m] += row[m];
        }
    }
    float gu0 = 0.f, gu1 = 0.f, gu2 = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        gu0 += gsum[m] * o[m].d[0]; gu1 += gsum[m] * o[m].d[1]; gu2 += gsum[m] * o[m].d[2];
    }
    const float dotp = gu0 * ux + gu1 * uy + gu2 * uz;      // u = v / |v|:  g_v = (g_u - (g_u . u) u) / |v|
    g_vec[3 * e] = (gu0 - dotp * ux) / n;
    g_vec[3 * e + 1] = (gu1 - dotp * uy) / n;
    g_vec[3 * e + 2] = (gu2 - dotp * uz) / n;
}

// g_pos[n] = sum_{e: src = n} gv[e] - sum_{e: dst = n} gv[e],  gv = g_vec + g_diff * vec / |vec|
// One wave per atom: the lanes stride over its outgoing (CSC) and incoming (CSR) edges, then a fixed-order wave sum per
// component (a thread per (atom, component) walked ~40 edges one dependent round trip at a time: 28 us at C2, 20 us for ONE
// molecule).
__global__ __launch_bounds__(256) void pos_scatter_kernel(const float* __restrict__ g_vec, const float* __restrict__ g_diff,
                                                          const float* __restrict__ vec,
                                                          const int* __restrict__ rowptr, const int* __restrict__ colptr,
                                                          const int* __restrict__ perm, int N, float sign, float* __restrict__ g_pos) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    auto add = [&](int e, float w) {
        const float x = vec[3 * e], y = vec[3 * e + 1], z = vec[3 * e + 2];
        const float gd = g_diff[e];
        float rx = g_vec[3 * e], ry = g_vec[3 * e + 1], rz = g_vec[3 * e + 2];
        if (gd != 0.f) {
            const float q = gd / sqrtf(x * x + y * y + z * z);
            rx += q * x; ry += q * y; rz += q * z;
        }
        sx += w * rx; sy += w * ry; sz += w * rz;
    };
    const int p0 = colptr[n], p1 = colptr[n + 1], e0 = rowptr[n], e1 = rowptr[n + 1];
    for (int pp = p0 + lane; pp < p1; pp += 64) add(perm[pp], 1.0f);
    for (int e = e0 + lane; e < e1; e += 64) add(e, -1.0f);
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    if (lane == 0) {
        g_pos[3 * n] = sign * sx;
        g_pos[3 * n + 1] = sign * sy;
        g_pos[3 * n + 2] = sign * sz;
    }
}

// =========================================================================== energy head (Atomwise)
// y_n = scale * (sum_k SiLU(pre1[n,k]) W2[k] + b2) + shift (+ atomref[z_n]);  E_mol = agg_{n in mol} y_n + mol_shift
// (mol_shift: AtomwiseV3 adds its mean AFTER the aggregation, outputs.py:212; Atomwise standardises per atom: 0)
__global__ __launch_bounds__(1024) void head_energy_kernel(
    const float* __restrict__ pre1, const float* __restrict__ W2, float b2, float scale, float shift, float mol_shift,
    const float* __restrict__ atomref, const int* __restrict__ z, const int* __restrict__ mol_ptr,
    int Hd, float* __restrict__ y, float* __restrict__ energy, int mean, float* __restrict__ atom_scale, int act) {
    const int b = blockIdx.x;
    const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int n = n0 + wave; n < n1; n += 16) {       // 16 waves: a 21-atom molecule is two dependent trips instead of six
        float s = 0.f;
        for (int k = lane; k < Hd; k += 64) s += act1(pre1[(size_t)n * Hd + k], act) * W2[k];
        s = wave_sum(s);
        if (lane == 0) y[n] = (s + b2) * scale + shift + (atomref ? atomref[z[n]] : 0.f);
    }
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int n = n0 + lane; n < n1; n += 64) s += y[n];
        s = wave_sum(s);
        if (lane == 0) energy[b] = ((mean && n1 > n0) ? s / (float)(n1 - n0) : s) + mol_shift;   // aggregation_mode "mean" / "sum"
    }
    if (atom_scale)                                  // d(aggregate) / d(y_n): what gn_head_grad multiplies by
        for (int n = n0 + (int)threadIdx.x; n < n1; n += 1024) atom_scale[n] = mean ? 1.0f / (float)(n1 - n0) : 1.0f;
}

__global__ void head_grad_kernel(const float* __restrict__ pre1, const float* __restrict__ W2, float scale,
                                 const float* __restrict__ atom_scale, int N, int Hd, float* __restrict__ gpre1, int act) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * Hd) return;
    const float sc = atom_scale ? scale * atom_scale[idx / Hd] : scale;
    gpre1[idx] = sc * W2[idx % Hd] * dact1(pre1[idx], act);
}

}  // namespace gn

// ====================================================================================== C ABI
// F <= 256: every kernel family.  F = 512 / 1024 (a slot spans 2 / 4 waves): the degree-sliced kernels and the init kernels;
// the per-edge scalar gradients then come as F / 256 partial slices per call (gn_common.h slot_sum_store)
static bool bwd_dim_ok_wide(int F) { return F >= 16 && F <= 1024 && gn::is_pow2(F); }

#define GN_SWITCH_LMAX(KERNEL, grid, block, st, ...)                                                   \
    switch (lmax) {                                                                                    \
        case 1: hipLaunchKernelGGL(gn::KERNEL<1>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 2: hipLaunchKernelGGL(gn::KERNEL<2>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 3: hipLaunchKernelGGL(gn::KERNEL<3>, grid, block, 0, st, __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL(gn::KERNEL<4>, grid, block, 0, st, __VA_ARGS__); break;            \
    }

#define GN_SWITCH_LMAX8(KERNEL, grid, block, st, ...)                                                  \
    switch (lmax) {                                                                                    \
        case 1: hipLaunchKernelGGL(gn::KERNEL<1>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 2: hipLaunchKernelGGL(gn::KERNEL<2>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 3: hipLaunchKernelGGL(gn::KERNEL<3>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 4: hipLaunchKernelGGL(gn::KERNEL<4>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 5: hipLaunchKernelGGL(gn::KERNEL<5>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 6: hipLaunchKernelGGL(gn::KERNEL<6>, grid, block, 0, st, __VA_ARGS__); break;             \
        case 7: hipLaunchKernelGGL(gn::KERNEL<7>, grid, block, 0, st, __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL(gn::KERNEL<8>, grid, block, 0, st, __VA_ARGS__); break;            \
    }

extern "C" int gn_htr_backward(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                               const float* EQ,
                               const float* EK, const float* rl, const int* rowptr, const int* src, const int* dst,
                               const int* colptr, const int* perm, int N, int F, int lmax_arg, int mode,
                               float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act, void* stream) {
    const int lmax = lmax_arg & 0xff;               // GN_LMAX_SLICED may ride in the argument (gn_use_highl)
    if ((lmax_arg & ~(0xff | GN_LMAX_SLICED)) || !bwd_dim_ok_wide(F) || N < 0 || lmax < 1 || lmax > 8 || mode < 0 || mode > 31 ||
        act < 0 || act >= GN_ACT_COUNT)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (gn_use_highl(lmax_arg) || F > 256 || (!mode && act != GN_ACT_SILU))      // (F > 256: g_rl comes as F / 256 partial slices)
        return gn_highl_htr_backward(g_t_out, pre_t, w, w_raw, EQ, EK, rl, rowptr, src, dst, colptr, perm, N, F, lmax,
                                     mode, gEQ, gEK, g_rl, g_pre_t, act, st);
    if (mode)
        return gn_htr_backward_general(g_t_out, pre_t, w, w_raw, EQ, EK, rl, rowptr, src, dst, colptr, perm, N, F, lmax,
                                       mode, gEQ, gEK, g_rl, g_pre_t, act, st);
    const dim3 grid(gn::xcd_grid(N)), block(256);
#define GN_HTRB_T(L, LLO, LHI, FIRST, FC)                                                                        \
    hipLaunchKernelGGL((gn::htr_bwd_target_group_kernel<L, LLO, LHI, FIRST, FC>), grid, block, 0, st, g_t_out, pre_t, w, \
                       EQ, EK, rl, rowptr, src, N, F, gEQ, g_rl, g_pre_t, act)
#define GN_HTRB_S(L, LLO, LHI, FC)                                                                               \
    hipLaunchKernelGGL((gn::htr_bwd_source_group_kernel<L, LLO, LHI, FC>), grid, block, 0, st, g_t_out, pre_t, EQ, EK, \
                       rl, colptr, perm, dst, N, F, gEK, act)
#define GN_HTRB(L, LLO, LHI, FIRST, FC) GN_HTRB_T(L, LLO, LHI, FIRST, FC); GN_HTRB_S(L, LLO, LHI, FC)
    // F == 256: the instantiations with the width as a compile-time constant and wave-uniform slots (gn_common.h GN_SLOT_GEOMETRY)
#define GN_HTRB_ALL(FC)                                                                                          \
    if (lmax == 1) {                                                                                             \
        hipLaunchKernelGGL((gn::htr_bwd_target_kernel<1, FC>), grid, block, 0, st, g_t_out, pre_t, w, EQ, EK, rl, rowptr, src, N, F, gEQ, g_rl, g_pre_t, act); \
        hipLaunchKernelGGL((gn::htr_bwd_source_kernel<1, FC>), grid, block, 0, st, g_t_out, pre_t, EQ, EK, rl, colptr, perm, dst, N, F, gEK, act); \
    } else if (lmax == 2) {                                                                                      \
        hipLaunchKernelGGL((gn::htr_bwd_target_kernel<2, FC>), grid, block, 0, st, g_t_out, pre_t, w, EQ, EK, rl, rowptr, src, N, F, gEQ, g_rl, g_pre_t, act); \
        hipLaunchKernelGGL((gn::htr_bwd_source_kernel<2, FC>), grid, block, 0, st, g_t_out, pre_t, EQ, EK, rl, colptr, perm, dst, N, F, gEK, act); \
    } else if (lmax == 3) {                                                                                      \
        if (GN_HTRB_TGT_MODE == 1) { GN_HTRB_T(3, 1, 3, true, FC); }                                             \
        else { GN_HTRB_T(3, 1, 2, true, FC); GN_HTRB_T(3, 3, 3, false, FC); }                                    \
        if (GN_HTRB_SRC_ONE) { GN_HTRB_S(3, 1, 3, FC); }                                                         \
        else { GN_HTRB_S(3, 1, 2, FC); GN_HTRB_S(3, 3, 3, FC); }                                                 \
    } else {                                                                                                     \
        GN_HTRB_T(4, 1, 2, true, FC);                                                                            \
        if (GN_HTRB_TGT_MODE == 1) { GN_HTRB_T(4, 3, 4, false, FC); }                                            \
        else { GN_HTRB_T(4, 3, 3, false, FC); GN_HTRB_T(4, 4, 4, false, FC); }                                   \
        if (GN_HTRB_SRC_ONE) { GN_HTRB_S(4, 1, 4, FC); }                                                         \
        else { GN_HTRB_S(4, 1, 2, FC); GN_HTRB_S(4, 3, 3, FC); GN_HTRB_S(4, 4, 4, FC); }                         \
    }
    if (F == 256) { GN_HTRB_ALL(256) } else { GN_HTRB_ALL(0) }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// the {3} group runs under its own occupancy hint (gn_tune.h); only the forms that are launched are instantiated
template <int L, int LLO, int LHI, bool SC, int FC>
static inline void gn_launch_msg_bwd_group(dim3 grid, dim3 block, hipStream_t st, const gn::MsgBwdArgs& p, float* ga_slice, float* cut_slice) {
    if constexpr (LLO == 3 && LHI == 3)
        hipLaunchKernelGGL((gn::msg_bwd_merged_group3_kernel<L, LLO, LHI, SC, FC>), grid, block, 0, st, p, ga_slice, cut_slice);
    else
        hipLaunchKernelGGL((gn::msg_bwd_merged_group_kernel<L, LLO, LHI, SC, FC>), grid, block, 0, st, p, ga_slice, cut_slice);
}
// (lmax >= 3 with sep_dir and sep_tensor never comes here -- the degree groups take it -- so that form is not instantiated)
template <int L, bool SD, bool ST, int FC>
static inline void gn_launch_msg_bwd_merged(dim3 grid, dim3 block, hipStream_t st, const gn::MsgBwdArgs& p, float* ga_parts) {
    if constexpr (!(L >= 3 && SD && ST))
        hipLaunchKernelGGL((gn::msg_bwd_merged_kernel<L, SD, ST, FC>), grid, block, 0, st, p, ga_parts);
}
// X_in == NULL: the zero-X_in instantiations (one target + one source launch at every lmax <= 4: without the tensor-gate
// rows the register budget that forces the degree groups is gone; g_cut then uses ONE slice, the caller zeroes the rest)
#define GN_MSGB_LAUNCH_FC(L, SD, ST, FC)                                                                 \
    do {                                                                                                  \
        if (!X_in) {                                                                                      \
            if (GN_MSGB_MERGED_FIRST && ga_parts != nullptr) {   /* first interaction: merged form without the tensor-gate blocks */ \
                hipLaunchKernelGGL((gn::msg_bwd_merged_first_kernel<L, SD, ST, FC>), grid, block, 0, st, p, ga_parts); \
                hipLaunchKernelGGL(gn::attn_bwd_kernel<FC>, grid, block, 0, st, p, ga_parts, 1, (size_t)0); \
                hipLaunchKernelGGL(gn::msg_bwd_gk_kernel<FC>, grid, block, 0, st, p);                     \
                break;                                                                                    \
            }                                                                                             \
            hipLaunchKernelGGL((gn::msg_bwd_target_kernel<L, SD, ST, true, FC>), grid, block, 0, st, p);  \
            hipLaunchKernelGGL((gn::msg_bwd_source_kernel<L, SD, ST, true, FC>), grid, block, 0, st, p);  \
            break;                                                                                        \
        }                                                                                                 \
        if (GN_MSGB_MERGED && ga_parts != nullptr) {     /* general launches: t_filter read once (gn_tune.h) */ \
            gn_launch_msg_bwd_merged<L, SD, ST, FC>(grid, block, st, p, ga_parts);                        \
            hipLaunchKernelGGL(gn::attn_bwd_kernel<FC>, grid, block, 0, st, p, ga_parts, 1, (size_t)0);   \
            hipLaunchKernelGGL(gn::msg_bwd_gk_kernel<FC>, grid, block, 0, st, p);                         \
            break;                                                                                        \
        }                                                                                                 \
        hipLaunchKernelGGL((gn::msg_bwd_target_kernel<L, SD, ST>), grid, block, 0, st, p);                \
        hipLaunchKernelGGL((gn::msg_bwd_source_kernel<L, SD, ST>), grid, block, 0, st, p);                \
    } while (0)
#define GN_MSGB_LAUNCH(L, SD, ST) GN_MSGB_LAUNCH_FC(L, SD, ST, 0)
// the reference's defaults (sep_dir, sep_tensor) at F = 256: the compile-time-width instantiations
#define GN_MSGB_LAUNCH_DEFAULT(L)                                                                        \
    do {                                                                                                  \
        if (F == 256) GN_MSGB_LAUNCH_FC(L, true, true, 256); else GN_MSGB_LAUNCH_FC(L, true, true, 0);    \
    } while (0)

extern "C" int gn_message_backward_groups(int lmax_arg, int sep_dir, int sep_tensor, int act) {
    const int lmax = lmax_arg & 0xff;
    if (gn_use_highl(lmax_arg) || act != GN_ACT_SILU) return 1;      // degree-sliced kernels (gn_highl.hip): one slice
    return (lmax >= 3 && sep_dir && sep_tensor) ? lmax - 1 : 1;
}

extern "C" int gn_message_backward(
    const float* x, const float* v, int ldxv, const float* eproj, int lde, const float* a,
    const float* qk, int ldqk, const float* X_in, const float* rl, const float* cut, const int* outdeg,
    const float* g_h1, const float* g_X1,
    const int* rowptr, const int* src, const int* dst, const int* colptr, const int* perm,
    float* g_eproj, float* g_s, float* g_nproj, int ldn, float* g_x, float* g_v, float* g_X_out,
    float* g_rl, float* g_cut, float* ga_parts, long E,
    int N, int F, int H, int lmax_arg, int sep_dir, int sep_tensor, int act, void* stream) {
    const int lmax = lmax_arg & 0xff;               // GN_LMAX_SLICED / GN_LMAX_MEAN / GN_LMAX_MAX may ride in the argument
    const bool amax = (lmax_arg & GN_LMAX_MAX) != 0;
    if (amax && ((lmax_arg & GN_LMAX_MEAN) || !ga_parts || !X_in)) return GN_ERR_BAD_ARG;   // "max": ga_parts = the [E, 1 + D, F] workspace
    if ((lmax_arg & ~(0xff | GN_LMAX_SLICED | GN_LMAX_MEAN | GN_LMAX_MAX)) || !bwd_dim_ok_wide(F) || N < 0 || H <= 0 || !gn::is_pow2(H) || (F / 4) % H || (F / 4) / H > 64 || lmax < 1 || lmax > 8 ||
        (ldxv & 3) || (lde & 3) || (ldqk & 3) || (ldn & 3) || g_X_out == g_X1 || act < 0 || act >= GN_ACT_COUNT)
        return GN_ERR_BAD_ARG;
    if (!X_in && (act != GN_ACT_SILU || gn_use_highl(lmax_arg) || F > 256)) return GN_ERR_BAD_ARG;   // zero-X_in form: register-tiled SiLU kernels only
    if (N == 0) return GN_OK;
    gn::MsgBwdArgs p{x, v, ldxv, eproj, lde, a, qk, ldqk, X_in, rl, cut, outdeg, g_h1, g_X1,
                     rowptr, src, dst, colptr, perm, g_eproj, g_s, g_nproj, ldn, g_x, g_v, g_X_out, g_rl, g_cut,
                     N, F, H, (float)(1.0 / sqrt((double)F)), act, (lmax_arg & GN_LMAX_MEAN) ? 1 : 0, amax ? ga_parts : nullptr};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(gn::xcd_grid(N)), block(256);
    // (F > 256: the degree-sliced kernels; g_rl / g_cut come as F / 256 partial slices, the caller sized them so)
    if (gn_use_highl(lmax_arg) || F > 256 || act != GN_ACT_SILU) return gn_highl_message_backward(p, lmax, sep_dir, sep_tensor, st);
    if (X_in && gn_message_backward_groups(lmax_arg, sep_dir, sep_tensor, act) > 1) {
        if (ga_parts == nullptr || E <= 0) return GN_ERR_BAD_ARG;
        const size_t gs = (size_t)E * H;
#define GN_MSGB_M(L, LLO, LHI, SC, G, FC)                                                                    \
    gn_launch_msg_bwd_group<L, LLO, LHI, SC, FC>(grid, block, st, p, ga_parts + (size_t)(G) * gs, g_cut + (size_t)(G) * E)
        // by-source group kernels with the per-edge work merged in (t_filter read once; head sums and cut slices per group)
        // -> attention backward over the summed head gradients -> g_k
#define GN_MSGB_GROUPS(FC)                                                                                       \
        if (lmax == 3) { GN_MSGB_M(3, 1, 2, true, 0, FC); GN_MSGB_M(3, 3, 3, false, 1, FC); }                    \
        else { GN_MSGB_M(4, 1, 2, true, 0, FC); GN_MSGB_M(4, 3, 3, false, 1, FC); GN_MSGB_M(4, 4, 4, false, 2, FC); } \
        hipLaunchKernelGGL(gn::attn_bwd_kernel<FC>, grid, block, 0, st, p, ga_parts, lmax - 1, gs);              \
        hipLaunchKernelGGL(gn::msg_bwd_gk_kernel<FC>, grid, block, 0, st, p);
        if (F == 256) { GN_MSGB_GROUPS(256) } else { GN_MSGB_GROUPS(0) }
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    const int key = lmax * 4 + (sep_dir ? 2 : 0) + (sep_tensor ? 1 : 0);
    switch (key) {
        case 4: case 5: case 6: case 7:
            if (F == 256) GN_MSGB_LAUNCH_FC(1, false, false, 256); else GN_MSGB_LAUNCH(1, false, false);
            break;
        case 8: GN_MSGB_LAUNCH(2, false, false); break;
        case 9: GN_MSGB_LAUNCH(2, false, true); break;
        case 10: GN_MSGB_LAUNCH(2, true, false); break;
        case 11: GN_MSGB_LAUNCH_DEFAULT(2); break;
        case 12: GN_MSGB_LAUNCH(3, false, false); break;
        case 13: GN_MSGB_LAUNCH(3, false, true); break;
        case 14: GN_MSGB_LAUNCH(3, true, false); break;
        case 15: GN_MSGB_LAUNCH_DEFAULT(3); break;
        case 16: GN_MSGB_LAUNCH(4, false, false); break;
        case 17: GN_MSGB_LAUNCH(4, false, true); break;
        case 18: GN_MSGB_LAUNCH(4, true, false); break;
        default: GN_MSGB_LAUNCH_DEFAULT(4); break;
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_eqff_backward_a(const float* g_h, const float* g_X, const float* m, const float* Xp,
                                  int N, int F, int D, float* g_m, float* g_Xp, void* stream) {
    if (N < 0 || F <= 0 || (F & 3) || D <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * (F / 4);
    hipLaunchKernelGGL(gn::eqff_bwd_a_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_h, g_X, m, Xp, N, F, D, g_m, g_Xp);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_eqff_backward_b(const float* g_ctx, const float* ctx, const float* Xp, const float* g_h,
                                  int N, int F, int D, float* g_Xp, float* g_h1, void* stream) {
    if (N < 0 || F <= 0 || (F & 3) || D <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * (F / 4);
    hipLaunchKernelGGL(gn::eqff_bwd_b_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_ctx, ctx, Xp, g_h, N, F, D, g_Xp, g_h1);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_init_backward(const float* g_t0, const float* h, const float* feat, int ldf,
                                     const int* rowptr, const int* src, const int* colptr, const int* perm,
                                     int N, int F, float* g_feat, float* g_h, void* stream) {
    if (!bwd_dim_ok_wide(F) || N < 0 || (ldf & 3)) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::edge_init_bwd_kernel, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                       g_t0, h, feat, ldf, rowptr, src, colptr, perm, N, F, g_feat, g_h);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_node_init_backward(const float* g_ctx, const int* z, const float* feat, int ldf, const float* cut,
                                     const float* A_nbr, const int* rowptr, const int* src, int N, int F,
                                     float* g_feat, float* g_cut, void* stream) {
    if (!bwd_dim_ok_wide(F) || N < 0 || (ldf & 3)) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::node_init_bwd_kernel, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                       g_ctx, z, feat, ldf, cut, A_nbr, rowptr, src, N, F, g_feat, g_cut);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_layernorm_silu_backward(const float* x, const float* gamma, const float* beta, float eps,
                                          const float* g_out, int N, int F, float* g_x, int act, void* stream) {
    if (N < 0 || F <= 0 || act < 0 || act >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::layernorm_bwd_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, gamma, beta, eps, g_out, N, F, g_x, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_layernorm_backward(const float* x, const float* gamma, float eps,
                                     const float* g_out, int N, int F, float* g_x, void* stream) {
    if (N < 0 || F <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::layernorm_bwd_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, gamma, gamma, eps, g_out, N, F, g_x, 0);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_geometry_backward(const float* edge_vec, const float* edge_diff, const int* src, const int* dst,
                                         int E, int lmax, int R, int basis, const float* means, const float* betas,
                                         float cutoff, const float* g_rl, int n_rl, const float* g_cut, int n_cut,
                                         const float* g_phi, float* g_vec, float* g_diff, void* stream) {
    if (E < 0 || lmax < 1 || lmax > 8 || R <= 0 || n_rl < 0 || n_cut < 0 || basis < 0 || basis > 2) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((E + 127) / 128), block(128);
    GN_SWITCH_LMAX8(edge_geometry_bwd_kernel, grid, block, st, edge_vec, edge_diff, src, dst, E, R, basis, means, betas,
                   cutoff, 5.0f / cutoff, g_rl, n_rl, g_cut, n_cut, g_phi, g_vec, g_diff);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_pos_scatter(const float* g_vec, const float* g_diff, const float* edge_vec,
                              const int* rowptr, const int* colptr, const int* perm, int N, float sign,
                              float* out, void* stream) {
    if (N < 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::pos_scatter_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       g_vec, g_diff, edge_vec, rowptr, colptr, perm, N, sign, out);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_head_energy(const float* pre1, const float* W2, float b2, float scale, float shift, float mol_shift,
                              const float* atomref, const int* z, const int* mol_ptr, int n_mol, int Hd,
                              float* y, float* energy, int mean, float* atom_scale, int act, void* stream) {
    if (n_mol < 0 || Hd <= 0 || act < 0 || act >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (n_mol == 0) return GN_OK;
    hipLaunchKernelGGL(gn::head_energy_kernel, dim3(n_mol), dim3(1024), 0, (hipStream_t)stream,
                       pre1, W2, b2, scale, shift, mol_shift, atomref, z, mol_ptr, Hd, y, energy, mean, atom_scale, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_head_grad(const float* pre1, const float* W2, float scale, const float* atom_scale, int N, int Hd,
                            float* g_pre1, int act, void* stream) {
    if (N < 0 || Hd <= 0 || act < 0 || act >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * Hd;
    hipLaunchKernelGGL(gn::head_grad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pre1, W2, scale, atom_scale, N, Hd, g_pre1, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
