// gn_gata.hip -- the GATA interaction kernels: attention scores + segment softmax,
// the fused message / segmented-reduction / residual kernel (K6, the HBM-bound
// gather-scatter stage this library is measured on) and the HTR edge weights (K7).
//
// All three use the per-target "slot" layout described in gn_edge.hip: CSR rows by
// target, one workgroup per target atom, F/4 lanes per incoming edge with 16-byte
// loads, register accumulation, fixed-order LDS reduction across slots (no atomics).
// Workgroup -> target mapping is XCD-aware (gn::xcd_item) so the source rows of one
// molecule are re-read through a single XCD's L2.
#include "gn_common.h"
#include "gn_tune.h"
#include "gn_highl.h"

namespace gn {

// ------------------------------------------------------------------ attention weights
// reference gotennet.py:497-511 + PyG softmax.  One workgroup per target.  The raw scores, their exponentials and the
// normalised weights of a target live in LDS (deg * H floats) and a[] is written ONCE, coalesced, at the end: the
// round-2 form kept them in the target's global a[] rows, i.e. four dependent global round trips (store scores ->
// barrier -> load / store exp -> load / store weights) per workgroup of a kernel that moves 55 MB -- 29 us at C2, 1.9 TB/s.
// Targets whose scores do not fit (deg * H > ATTN_CAP: a 256-neighbour atom at 8 heads still fits) take the global
// form; the choice is workgroup-uniform and made ONCE (a per-access select cost 31 -> 38 us in round 2), the
// arithmetic and its order are the same in both, so the two forms agree bit for bit.
// ASILU: activation fixed to SiLU at compile time (the run-time switch over twelve kinds costs registers and branches).
constexpr int ATTN_CAP = 2048;
template <bool IN_LDS, bool ASILU>
__device__ __forceinline__ void attn_softmax_body(
    const float* __restrict__ q, const float* __restrict__ k, int ldqk, const float* __restrict__ ta, int ldt,
    const int* __restrict__ src, const int* __restrict__ outdeg, int i, int e0, int e1, int F, int H, float inv_sqrt_f,
    float* __restrict__ a, float* sc, int act) {
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, lp = threadIdx.x % lps, c0 = lp * 4;
    const int lph = lps / H;                       // lanes per head (power of two, >= 1)
    auto S = [&](int e, int h) -> float& {          // this target's score of edge e, head h
        if constexpr (IN_LDS) return sc[(e - e0) * H + h];
        else return a[(size_t)e * H + h];
    };
    const float4 qi = ld4(q + (size_t)i * ldqk + c0);
    // two edges per trip: both index -> row load chains in flight together (a slot sees ~5 edges of a 20-neighbour atom)
    for (int e = e0 + slot; e < e1; e += 2 * ns) {
        const int eb = e + ns < e1 ? e + ns : e;
        const int ja = src[e], jb = src[eb];
        const float4 ka = ld4(k + (size_t)ja * ldqk + c0), kb = ld4(k + (size_t)jb * ldqk + c0);
        const float4 ta_ = act4(ld4_nt(ta + (size_t)e * ldt + c0), act);     // stored pre-activation: t_attn = act(.)
        const float4 tb_ = act4(ld4_nt(ta + (size_t)eb * ldt + c0), act);
        float pa = qi.x * ka.x * ta_.x;
        pa += qi.y * ka.y * ta_.y;
        pa += qi.z * ka.z * ta_.z;
        pa += qi.w * ka.w * ta_.w;
        float pb = qi.x * kb.x * tb_.x;
        pb += qi.y * kb.y * tb_.y;
        pb += qi.z * kb.z * tb_.z;
        pb += qi.w * kb.w * tb_.w;
        pa = group_sum(pa, lph);
        pb = group_sum(pb, lph);
        if ((lp & (lph - 1)) == 0) {
            S(e, lp / lph) = pa;
            if (eb != e) S(eb, lp / lph) = pb;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < H; h += 4) {
        float mx = -INFINITY;
        for (int e = e0 + lane; e < e1; e += 64) mx = fmaxf(mx, S(e, h));
        mx = wave_max(mx);
        float sm = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) {
            const float ex = fast_exp(S(e, h) - mx);
            S(e, h) = ex;
            sm += ex;
        }
        const float rsm = __builtin_amdgcn_rcpf(wave_sum(sm) + 1e-16f);
        for (int e = e0 + lane; e < e1; e += 64) {
            const float nrm = outdeg ? sqrtf((float)outdeg[src[e]]) * inv_sqrt_f : inv_sqrt_f;
            S(e, h) = S(e, h) * rsm * nrm;
        }
    }
    if constexpr (IN_LDS) {
        __syncthreads();
        const int n = (e1 - e0) * H;
        for (int idx = threadIdx.x; idx < n; idx += 256) a[(size_t)e0 * H + idx] = sc[idx];
    }
}

template <bool ASILU>
__global__ __launch_bounds__(256) GN_WPE(GN_W_ATTN) void attn_softmax_kernel(
    const float* __restrict__ q, const float* __restrict__ k, int ldqk,
    const float* __restrict__ ta, int ldt,
    const int* __restrict__ rowptr, const int* __restrict__ src, const int* __restrict__ outdeg,
    int N, int F, int H, float inv_sqrt_f, float* __restrict__ a, int act_rt) {
    __shared__ float sc[ATTN_CAP];
    const int act = ASILU ? (int)GN_ACT_SILU : act_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    if ((e1 - e0) * H <= ATTN_CAP)
        attn_softmax_body<true, ASILU>(q, k, ldqk, ta, ldt, src, outdeg, i, e0, e1, F, H, inv_sqrt_f, a, sc, act);
    else
        attn_softmax_body<false, ASILU>(q, k, ldqk, ta, ldt, src, outdeg, i, e0, e1, F, H, inv_sqrt_f, a, sc, act);
}

// ---- one WAVE per target (F <= 256, H | 64): what ships for the default shapes.
// Round-3 counters on the workgroup-per-target kernel above (C2: 29 us for 55 MB): VALU-bound -- 1080 VALU instructions
// per wave, four waves per target, 20 % of the wave time issuing, the rest stalled on issue or parked at two barriers and
// 24 ds_bpermute round trips of the per-head reductions.  Here a target is ONE wave: its edges are walked 64 / (F / 4)
// at a time with the loads of eight steps in flight, the raw scores go to the wave's LDS strip, and the per-head max /
// sum run over the strip with (edge, head) = strip index per lane: the head of a lane is lane % H for every pass
// (H | 64), so the reductions are log2(64 / H) xor-shuffles, there is no barrier anywhere, and a[] is written once,
// coalesced.  Targets whose scores exceed the strip use their a[] rows as the strip (same code, same order).
constexpr int ATTN_W_STRIP = 512;                   // floats per wave: 64 incoming edges at 8 heads
template <bool IN_LDS, bool ASILU, int FC>
__device__ __forceinline__ void attn_softmax_wave_body(
    const float* __restrict__ q, const float* __restrict__ k, int ldqk, const float* __restrict__ ta, int ldt,
    const int* __restrict__ src, const int* __restrict__ outdeg, int i, int e0, int e1, int F_rt, int H, float inv_sqrt_f,
    float* __restrict__ a, float* sc, int act) {
    const int F = FC ? FC : F_rt;                   // FC = 256: one edge per step, every index of the walk is wave-uniform
    const int lane = threadIdx.x & 63;
    const int lps = F >> 2, ns = 64 / lps;          // lanes per edge, edges per step
    const int slot = FC == 256 ? 0 : lane / lps, lp = lane % lps, c0 = lp * 4;
    const int lph = lps / H;                        // lanes per head
    float* const S = IN_LDS ? sc : a + (size_t)e0 * H;            // strip: S[(e - e0) * H + h]
    const int n = (e1 - e0) * H;
    const float4 qi = ld4(q + (size_t)i * ldqk + c0);
    constexpr int U = 8;                            // steps per trip: U (index -> row) chains and U streamed rows in flight
    for (int eb = e0; eb < e1; eb += U * ns) {      // (a 20-neighbour target is three trips = three memory round trips)
        float4 kj[U], te[U];
        int ee[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = eb + u * ns + slot;
            ee[u] = e < e1 ? e : e1 - 1;            // clamped: a valid row, its score is not stored
            kj[u] = ld4(k + (size_t)src[ee[u]] * ldqk + c0);
            te[u] = ld4_nt(ta + (size_t)ee[u] * ldt + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 t4 = act4(te[u], act);     // stored pre-activation: t_attn = act(.)
            float p = qi.x * kj[u].x * t4.x;
            p += qi.y * kj[u].y * t4.y;
            p += qi.z * kj[u].z * t4.z;
            p += qi.w * kj[u].w * t4.w;
            p = group_sum(p, lph);
            if ((lp & (lph - 1)) == 0 && eb + u * ns + slot < e1) S[(ee[u] - e0) * H + lp / lph] = p;
        }
    }
    if constexpr (!IN_LDS) __builtin_amdgcn_s_waitcnt(0);         // the strip is this target's a[] rows: stores done before the loads
    __builtin_amdgcn_wave_barrier();
    // per-head max and sum over the strip; lane's head = lane % H in every pass
    float mx = -INFINITY;
    for (int idx = lane; idx < n; idx += 64) mx = fmaxf(mx, S[idx]);
    mx = stride_max(mx, H);
    float sm = 0.f;
    for (int idx = lane; idx < n; idx += 64) {
        const float ex = fast_exp(S[idx] - mx);
        S[idx] = ex;
        sm += ex;
    }
    sm = stride_sum(sm, H);
    const float rsm = __builtin_amdgcn_rcpf(sm + 1e-16f);
    if constexpr (!IN_LDS) __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < n; idx += 64) {
        const float nrm = outdeg ? sqrtf((float)outdeg[src[e0 + idx / H]]) * inv_sqrt_f : inv_sqrt_f;
        a[(size_t)e0 * H + idx] = S[idx] * rsm * nrm;
    }
}

template <bool ASILU, int FC = 0>
__global__ __launch_bounds__(256) void attn_softmax_wave_kernel(
    const float* __restrict__ q, const float* __restrict__ k, int ldqk,
    const float* __restrict__ ta, int ldt,
    const int* __restrict__ rowptr, const int* __restrict__ src, const int* __restrict__ outdeg,
    int N, int F, int H, float inv_sqrt_f, float* __restrict__ a, int act_rt) {
    __shared__ float strips[4 * ATTN_W_STRIP];
    const int act = ASILU ? (int)GN_ACT_SILU : act_rt;
    const int grp = xcd_item(blockIdx.x, (N + 3) >> 2);           // four consecutive targets per workgroup, one per wave
    if (grp < 0) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = 4 * grp + wave;
    if (i >= N) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    if (e1 == e0) return;
    float* sc = strips + wave * ATTN_W_STRIP;
    if ((e1 - e0) * H <= ATTN_W_STRIP)
        attn_softmax_wave_body<true, ASILU, FC>(q, k, ldqk, ta, ldt, src, outdeg, i, e0, e1, F, H, inv_sqrt_f, a, sc, act);
    else
        attn_softmax_wave_body<false, ASILU, FC>(q, k, ldqk, ta, ldt, src, outdeg, i, e0, e1, F, H, inv_sqrt_f, a, sc, act);
}

// ------------------------------------------------------------------ K6 message + aggregate (lmax <= 2: one launch)
// M F-wide blocks of the value vector: 0 = scalar; direction gate of degree l: block
// (SEP_DIR ? l : 1); tensor gate: block TB0 + (SEP_TENSOR ? l-1 : 0), TB0 = 1 + (SEP_DIR ? LMAX : 1).
// FIRST: X_in is identically zero (the first interaction of GotenNet.forward, gotennet.py:992): the tensor-gate blocks
// of t_filter / x / v and the X_in rows are not read (0 * gate contributes nothing) and X_out = the aggregated update.
// Same bits as the general kernel on a zero X_in.
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, bool FIRST = false, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_K6) void message_aggregate_kernel(
    const float* __restrict__ x, const float* __restrict__ v, int ldxv,
    const float* __restrict__ tf, int ldt, const float* __restrict__ a,
    const float* __restrict__ rl, const float* __restrict__ cut,
    const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ h_in, const float* __restrict__ X_in,
    float* __restrict__ h_out, float* __restrict__ X_out, int N, int F_rt, int H) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int ND = SEP_DIR ? LMAX : 1;
    constexpr int NT = SEP_TENSOR ? LMAX : 1;
    constexpr int M = 1 + ND + NT;
    constexpr int ROWS = 1 + D;
    constexpr int CH = ROWS < 9 ? ROWS : 9;         // rows reduced per LDS pass (<= 36 KiB)
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];

    const int F = FC ? FC : F_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int per_head = (M * F) / H;

    int hb[M];
#pragma unroll
    for (int b = 0; b < M; ++b) hb[b] = (b * F + c0) / per_head;

    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();

    for (int e = e0 + slot; e < e1; e += ns) {
        const int j = src[e];
        const float ce = cut[e];
        const float* xr = x + (size_t)j * ldxv + c0;
        const float* vr = v + (size_t)j * ldxv + c0;
        const float* tr = tf + (size_t)e * ldt + c0;
        const float* ar = a + (size_t)e * H;
        const float* Xj = X_in + (size_t)j * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 o[M];
#pragma unroll
        for (int b = 0; b < (FIRST ? 1 + ND : M); ++b) {
            // gotennet.py:516-529: (t_filter * x_j) * cutoff + attn * v_j
            const float4 sp = (ld4_nt(tr + b * F) * ld4(xr + b * F)) * ce;
            const float ab = ar[hb[b]];
            o[b] = fma4(ab, ld4(vr + b * F), sp);
        }
        acc[0] = acc[0] + o[0];
        int m = 0;
#pragma unroll
        for (int l = 1; l <= LMAX; ++l) {
            const float4 od = o[SEP_DIR ? l : 1];
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm, ++m) {
                // gotennet.py:538-558: rl * o_d + X_j * o_t
                if constexpr (FIRST)
                    acc[1 + m] = acc[1 + m] + od * re[m];
                else
                    acc[1 + m] = acc[1 + m] + fma4(ld4(Xj + (size_t)m * F), o[1 + ND + (SEP_TENSOR ? l - 1 : 0)], od * re[m]);
            }
        }
    }

    // fixed-order reduction over slots, CH rows per pass; slot s finishes rows s, s+ns, ...
#pragma unroll
    for (int base = 0; base < ROWS; base += CH) {
        if (base) __syncthreads();
#pragma unroll
        for (int r = 0; r < CH; ++r)
            if (base + r < ROWS) st4(&red[r * 1024 + slot * F + c0], acc[base + r]);
        __syncthreads();
        for (int r = slot; r < CH && base + r < ROWS; r += ns) {
            const float4 s = red4(red + r * 1024, c0, F, ns);
            const int row = base + r;
            if (row == 0) {
                st4(h_out + (size_t)i * F + c0, ld4(h_in + (size_t)i * F + c0) + s);
            } else {
                const size_t off = ((size_t)i * D + (row - 1)) * F + c0;
                if constexpr (FIRST) st4(X_out + off, s);
                else st4(X_out + off, ld4(X_in + off) + s);
            }
        }
    }
}

// ------------------------------------------------------------------ K6 message + aggregate (lmax >= 3: degree groups)
// M F-wide blocks of the value vector: 0 = scalar; direction gate of degree l: block
// (SEP_DIR ? l : 1); tensor gate: block 1 + ND + (SEP_TENSOR ? l-1 : 0), ND = SEP_DIR ? LMAX : 1.
//
// One launch covers the degrees LLO..LHI (and the scalar row when SCALAR): for lmax >= 3 the
// (1 + D) accumulator rows are cut into degree groups {scalar,1,2}, {3}, {4} so that every launch
// keeps <= 9 float4 accumulators per lane (3+ waves/SIMD instead of 2 at 246 VGPRs).  Gates are
// per degree, so the groups re-read nothing but the per-edge scalars.
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int LLO, int LHI, bool SCALAR, int FC>
__device__ __forceinline__ void message_aggregate_group_body(
    const float* __restrict__ x, const float* __restrict__ v, int ldxv,
    const float* __restrict__ tf, int ldt, const float* __restrict__ a,
    const float* __restrict__ rl, const float* __restrict__ cut,
    const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ h_in, const float* __restrict__ X_in,
    float* __restrict__ h_out, float* __restrict__ X_out, int N, int F_rt, int H) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int ND = SEP_DIR ? LMAX : 1;
    constexpr int NT = SEP_TENSOR ? LMAX : 1;
    constexpr int M = 1 + ND + NT;
    constexpr int XROWS = (LHI + 1) * (LHI + 1) - LLO * LLO;     // rows of degrees LLO..LHI
    constexpr int ROWS = (SCALAR ? 1 : 0) + XROWS;
    constexpr int M0 = LLO * LLO - 1;                             // first X row of the group
    constexpr int CH = ROWS < GN_K6G_CH ? ROWS : GN_K6G_CH;       // rows reduced per LDS pass (4 KiB each)
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];

    const int F = FC ? FC : F_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int per_head = (M * F) / H;

    int hb[M];                                      // attention head of this lane's channels in block b
#pragma unroll
    for (int b = 0; b < M; ++b) hb[b] = (b * F + c0) / per_head;

    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();

    for (int e = e0 + slot; e < e1; e += ns) {
        const int j = src[e];
        const float ce = cut[e];
        const float* xr = x + (size_t)j * ldxv + c0;
        const float* vr = v + (size_t)j * ldxv + c0;
        const float* tr = tf + (size_t)e * ldt + c0;
        const float* ar = a + (size_t)e * H;
        const float* Xj = X_in + (size_t)j * D * F + c0;
        const float* re = rl + (size_t)e * D;
        // gotennet.py:516-529: (t_filter * x_j) * cutoff + attn * v_j, block b of the value vector
        auto gate = [&](int b) {
            const float4 sp = (ld4_nt(tr + b * F) * ld4(xr + b * F)) * ce;
            const float ab = ar[hb[b]];
            return fma4(ab, ld4(vr + b * F), sp);
        };
        // all gate loads first (independent, issued back to back), then the X_j rows
        constexpr int NL = LHI - LLO + 1;
        float4 gd[NL], gt[NL];
#pragma unroll
        for (int l = LLO; l <= LHI; ++l) {
            gd[l - LLO] = (SEP_DIR || l == LLO) ? gate(SEP_DIR ? l : 1) : gd[0];
            gt[l - LLO] = (SEP_TENSOR || l == LLO) ? gate(1 + ND + (SEP_TENSOR ? l - 1 : 0)) : gt[0];
        }
        if (SCALAR) acc[0] = acc[0] + gate(0);
#pragma unroll
        for (int l = LLO; l <= LHI; ++l) {
#pragma unroll
            for (int mm = 0; mm < 2 * l + 1; ++mm) {
                const int m = l * l - 1 + mm;
                // gotennet.py:538-558: rl * o_d + X_j * o_t
                acc[(SCALAR ? 1 : 0) + m - M0] = acc[(SCALAR ? 1 : 0) + m - M0] +
                                                 fma4(ld4(Xj + (size_t)m * F), gt[l - LLO], gd[l - LLO] * re[m]);
            }
        }
    }

    // fixed-order reduction over slots, CH rows per pass; slot s finishes rows s, s+ns, ...
#pragma unroll
    for (int base = 0; base < ROWS; base += CH) {
        if (base) __syncthreads();
#pragma unroll
        for (int r = 0; r < CH; ++r)
            if (base + r < ROWS) st4(&red[r * 1024 + slot * F + c0], acc[base + r]);
        __syncthreads();
        for (int r = slot; r < CH && base + r < ROWS; r += ns) {
            const float4 s = red4(red + r * 1024, c0, F, ns);
            const int row = base + r;
            if (SCALAR && row == 0) {
                st4(h_out + (size_t)i * F + c0, ld4(h_in + (size_t)i * F + c0) + s);
            } else {
                const size_t off = ((size_t)i * D + (M0 + row - (SCALAR ? 1 : 0))) * F + c0;
                st4(X_out + off, ld4(X_in + off) + s);
            }
        }
    }
}

#define GN_MSG_GROUP_ARGS                                                                                   \
    const float *__restrict__ x, const float *__restrict__ v, int ldxv, const float *__restrict__ tf, int ldt,            \
        const float *__restrict__ a, const float *__restrict__ rl, const float *__restrict__ cut, const int *__restrict__ rowptr,    \
        const int *__restrict__ src, const float *__restrict__ h_in, const float *__restrict__ X_in,                     \
        float *__restrict__ h_out, float *__restrict__ X_out, int N, int F, int H
#define GN_MSG_GROUP_PASS x, v, ldxv, tf, ldt, a, rl, cut, rowptr, src, h_in, X_in, h_out, X_out, N, F, H
// one degree group per launch; the two-degree group {3,4} (16 accumulator rows) gets its own occupancy target
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int LLO, int LHI, bool SCALAR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_K6_G) void message_aggregate_group_kernel(GN_MSG_GROUP_ARGS) {
    message_aggregate_group_body<LMAX, SEP_DIR, SEP_TENSOR, LLO, LHI, SCALAR, FC>(GN_MSG_GROUP_PASS);
}
template <int LMAX, bool SEP_DIR, bool SEP_TENSOR, int LLO, int LHI, bool SCALAR, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_K6_G34) void message_aggregate_group34_kernel(GN_MSG_GROUP_ARGS) {
    message_aggregate_group_body<LMAX, SEP_DIR, SEP_TENSOR, LLO, LHI, SCALAR, FC>(GN_MSG_GROUP_PASS);
}

// ------------------------------------------------------------------ K7 HTR edge weights
// gotennet.py:351-364, 580-609 (sep_htr, rejection on).  LMAX <= 2: the literal two-rejection form
//   w_l = sum_m (EQ - (EQ.r) r)_m (EK - (EK.r) r)_m.
// LMAX == 3 (GN_HTR_CLOSED): the algebraically equal closed form  w_l = EQ.EK - (2 - r.r)(EQ.r)(EK.r)  -- 3 instead of 5
// float4 FMAs per row and no second pass over the rows (the form gn_options.hip uses for the non-default variants);
// r.r is not 1 for the reference's degree 3 (and 0 on self-loops), so it is carried.  Rounding differs from the literal
// form at the 1e-7 level.  Measured (round 4, in the step): lmax 3 (C5) 98.0 -> 90.5 us (126 VGPRs, 4 waves/SIMD instead
// of 132 / 3); at lmax 4 the same code LOSES, 105.9 -> 140.4 us at the same 2 waves/SIMD (the literal form keeps all 24 row
// loads of an edge in flight; the closed form's schedule does not).  LMAX >= GN_HTR_CLOSED_ALL (= 4): the closed form with
// every row of the edge requested before the first use (a sched_barrier keeps the loads together): 108.4 -> 90 us.
template <int LMAX, int FC = 0>
__global__ __launch_bounds__(256) GN_WPE(GN_W_HTR_EDGE) void htr_edge_kernel(
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F_rt, float* __restrict__ w) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr bool CLOSED = LMAX == 3 && GN_HTR_CLOSED;
    const int F = FC ? FC : F_rt;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    GN_SLOT_GEOMETRY(FC);
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 eq[D];
#pragma unroll
    for (int m = 0; m < D; ++m) eq[m] = ld4(EQ + ((size_t)i * D + m) * F + c0);
    for (int e = e0 + slot; e < e1; e += ns) {
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 wsum = zero4();
        int m0 = 0;
#if GN_HTR_CLOSED_ALL
        if constexpr (LMAX >= GN_HTR_CLOSED_ALL) {
            // every row of the edge requested before the first use (the literal form gets that from the compiler, the
            // closed form degree by degree does not), then  w_l = EQ.EK - (2 - r.r)(EQ.r)(EK.r)
            float4 eka[D];
            float ra[D];
#pragma unroll
            for (int m = 0; m < D; ++m) { eka[m] = ld4(kj + (size_t)m * F); ra[m] = re[m]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 1; l <= LMAX; ++l) {
                float4 pq = zero4(), pk = zero4(), ab = zero4();
                float rr = 0.f;
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    const int m = l * l - 1 + mm;
                    rr = fmaf(ra[m], ra[m], rr);
                    pq = fma4(ra[m], eq[m], pq);
                    pk = fma4(ra[m], eka[m], pk);
                    ab = fma4(eq[m], eka[m], ab);
                }
                wsum = wsum + (ab + (pq * pk) * (rr - 2.0f));
            }
            st4_nt(w + (size_t)e * F + c0, wsum);
            continue;
        }
#endif
#pragma unroll
        for (int l = 1; l <= LMAX; ++l) {
            float4 ek[2 * LMAX + 1];
            float r[2 * LMAX + 1];
            float4 pq = zero4(), pk = zero4();
            if constexpr (CLOSED) {
                float4 ab = zero4();
                float rr = 0.f;
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    ek[mm] = ld4(kj + (size_t)(m0 + mm) * F);
                    r[mm] = re[m0 + mm];
                }
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    rr = fmaf(r[mm], r[mm], rr);
                    pq = fma4(r[mm], eq[m0 + mm], pq);
                    pk = fma4(r[mm], ek[mm], pk);
                    ab = fma4(eq[m0 + mm], ek[mm], ab);
                }
                wsum = wsum + (ab + (pq * pk) * (rr - 2.0f));
            } else {
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    ek[mm] = ld4(kj + (size_t)(m0 + mm) * F);
                    r[mm] = re[m0 + mm];
                    pq = fma4(r[mm], eq[m0 + mm], pq);
                    pk = fma4(-r[mm], ek[mm], pk);
                }
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    const float4 a_ = eq[m0 + mm] + pq * (-r[mm]);       // EQ - proj * rl
                    const float4 b_ = ek[mm] + pk * r[mm];               // EK - proj' * (-rl)
                    wsum = fma4(a_, b_, wsum);
                }
            }
            m0 += 2 * l + 1;
        }
        st4_nt(w + (size_t)e * F + c0, wsum);
    }
}

}  // namespace gn

// ====================================================================================== C ABI
static bool feature_dim_ok(int F) { return F >= 16 && F <= 1024 && gn::is_pow2(F); }

extern "C" int gn_attn_softmax(const float* q, const float* k, int ldqk, const float* t_attn, int ldt,
                               const int* rowptr, const int* src, const int* outdeg,
                               int N, int F, int H, float* a, int act, void* stream) {
    if (!feature_dim_ok(F) || N < 0 || H <= 0 || !gn::is_pow2(H) || (F / 4) % H || (F / 4) / H > 64 ||
        (ldqk & 3) || (ldt & 3) || act < 0 || act >= GN_ACT_COUNT)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const float inv_sqrt_f = (float)(1.0 / sqrt((double)F));
    // one wave per target where a wave covers an edge row and the strip's head-per-lane map holds (-DGN_ATTN_WAVE=0
    // builds the workgroup-per-target kernel only, for an A/B: tools/variants.py)
    if (GN_ATTN_WAVE && F <= 256 && H <= 64) {
        const dim3 grid(gn::xcd_grid((N + 3) / 4)), block(256);
        if (act == GN_ACT_SILU && F == 256)            // width as a compile-time constant (gn_common.h GN_SLOT_GEOMETRY)
            hipLaunchKernelGGL((gn::attn_softmax_wave_kernel<true, 256>), grid, block, 0, (hipStream_t)stream,
                               q, k, ldqk, t_attn, ldt, rowptr, src, outdeg, N, F, H, inv_sqrt_f, a, act);
        else if (act == GN_ACT_SILU)
            hipLaunchKernelGGL(gn::attn_softmax_wave_kernel<true>, grid, block, 0, (hipStream_t)stream,
                               q, k, ldqk, t_attn, ldt, rowptr, src, outdeg, N, F, H, inv_sqrt_f, a, act);
        else
            hipLaunchKernelGGL(gn::attn_softmax_wave_kernel<false>, grid, block, 0, (hipStream_t)stream,
                               q, k, ldqk, t_attn, ldt, rowptr, src, outdeg, N, F, H, inv_sqrt_f, a, act);
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    if (act == GN_ACT_SILU)
        hipLaunchKernelGGL(gn::attn_softmax_kernel<true>, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                           q, k, ldqk, t_attn, ldt, rowptr, src, outdeg, N, F, H, inv_sqrt_f, a, act);
    else
        hipLaunchKernelGGL(gn::attn_softmax_kernel<false>, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                           q, k, ldqk, t_attn, ldt, rowptr, src, outdeg, N, F, H, inv_sqrt_f, a, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

#define GN_MSG_ONE(L, SD, ST, LLO, LHI, SC, FC)                                                                 \
    hipLaunchKernelGGL((gn::message_aggregate_group_kernel<L, SD, ST, LLO, LHI, SC, FC>), dim3(gn::xcd_grid(N)), dim3(256), \
                       0, (hipStream_t)stream, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in,      \
                       h_out, X_out, N, F, H)
// degree groups per lmax: {scalar,1..min(lmax,2)}, {3}, {4} (lmax = 4: {3,4} in one launch, GN_K6_MERGE34)
#define GN_MSG_MONO(L, SD, ST, FC)                                                                          \
    if (X_in)                                                                                                   \
        hipLaunchKernelGGL((gn::message_aggregate_kernel<L, SD, ST, false, FC>), dim3(gn::xcd_grid(N)), dim3(256), 0, \
                           (hipStream_t)stream, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, \
                           h_out, X_out, N, F, H);                                                              \
    else                                                                                                        \
        hipLaunchKernelGGL((gn::message_aggregate_kernel<L, SD, ST, true, FC>), dim3(gn::xcd_grid(N)), dim3(256), 0,  \
                           (hipStream_t)stream, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, \
                           h_out, X_out, N, F, H)
#define GN_MSG_LAUNCH_FC(L, SD, ST, FC)                                   \
    do {                                                                  \
        if (L <= 2 || !X_in) { GN_MSG_MONO(L, SD, ST, FC); }  /* zero X_in: 1 + D rows of rl * o_d only, one launch */ \
        else {                                                            \
            GN_MSG_ONE(L, SD, ST, 1, 2, true, FC);                        \
            if constexpr (L >= 4 && GN_K6_MERGE34) {                      \
                hipLaunchKernelGGL((gn::message_aggregate_group34_kernel<L, SD, ST, 3, 4, false, FC>), dim3(gn::xcd_grid(N)), \
                                   dim3(256), 0, (hipStream_t)stream, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, \
                                   h_in, X_in, h_out, X_out, N, F, H);    \
            }                                                             \
            else {                                                        \
                GN_MSG_ONE(L, SD, ST, 3, 3, false, FC);                   \
                if constexpr (L >= 4) { GN_MSG_ONE(L, SD, ST, 4, 4, false, FC); } \
            }                                                             \
        }                                                                 \
    } while (0)
#define GN_MSG_LAUNCH(L, SD, ST) GN_MSG_LAUNCH_FC(L, SD, ST, 0)
// the reference's defaults (sep_dir, sep_tensor) at F = 256: the compile-time-width instantiations
#define GN_MSG_LAUNCH_DEFAULT(L)                                                                 \
    do {                                                                                          \
        if (F == 256) GN_MSG_LAUNCH_FC(L, true, true, 256); else GN_MSG_LAUNCH_FC(L, true, true, 0); \
    } while (0)

extern "C" int gn_message_aggregate(const float* x, const float* v, int ldxv, const float* t_filter, int ldt,
                                    const float* a, const float* rl, const float* cut,
                                    const int* rowptr, const int* src,
                                    const float* h_in, const float* X_in, float* h_out, float* X_out,
                                    int N, int F, int H, int lmax_arg, int sep_dir, int sep_tensor, void* stream) {
    const int lmax = lmax_arg & 0xff;               // GN_LMAX_SLICED / _MEAN / _MAX may ride in the argument (gn_use_highl)
    const int aggr = (lmax_arg & GN_LMAX_MEAN) ? 1 : ((lmax_arg & GN_LMAX_MAX) ? 2 : 0);
    if ((lmax_arg & ~(0xff | GN_LMAX_SLICED | GN_LMAX_MEAN | GN_LMAX_MAX)) ||
        ((lmax_arg & GN_LMAX_MEAN) && (lmax_arg & GN_LMAX_MAX)) || !feature_dim_ok(F) || N < 0 || H <= 0 || lmax < 1 ||
        lmax > 8 || (ldxv & 3) || (ldt & 3) || X_in == X_out)
        return GN_ERR_BAD_ARG;
    if (!X_in && gn_use_highl(lmax_arg)) return GN_ERR_BAD_ARG;   // the zero-X_in form: register-tiled kernels only (lmax <= 4)
    const int M = 1 + (sep_dir ? lmax : 1) + (sep_tensor ? lmax : 1);
    if ((M * F) % H || ((M * F) / H) % 4) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    if (gn_use_highl(lmax_arg))                        // degrees 5..8: one launch per degree (gn_highl.hip)
        return gn_highl_message(x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, h_out, X_out, N, F, H,
                                lmax, sep_dir, sep_tensor, aggr, (hipStream_t)stream);
    const int key = lmax * 4 + (sep_dir ? 2 : 0) + (sep_tensor ? 1 : 0);
    switch (key) {
        case 4: case 5: case 6: case 7:                                          // lmax = 1: flags are no-ops
            if (F == 256) GN_MSG_LAUNCH_FC(1, false, false, 256); else GN_MSG_LAUNCH(1, false, false);
            break;
        case 8: GN_MSG_LAUNCH(2, false, false); break;
        case 9: GN_MSG_LAUNCH(2, false, true); break;
        case 10: GN_MSG_LAUNCH(2, true, false); break;
        case 11: GN_MSG_LAUNCH_DEFAULT(2); break;
        case 12: GN_MSG_LAUNCH(3, false, false); break;
        case 13: GN_MSG_LAUNCH(3, false, true); break;
        case 14: GN_MSG_LAUNCH(3, true, false); break;
        case 15: GN_MSG_LAUNCH_DEFAULT(3); break;
        case 16: GN_MSG_LAUNCH(4, false, false); break;
        case 17: GN_MSG_LAUNCH(4, false, true); break;
        case 18: GN_MSG_LAUNCH(4, true, false); break;
        default: GN_MSG_LAUNCH_DEFAULT(4); break;
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_htr_edge(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                           int N, int F, int lmax_arg, int mode, float* w_raw, float* w, void* stream) {
    const int lmax = lmax_arg & 0xff;
    if ((lmax_arg & ~(0xff | GN_LMAX_SLICED)) || !feature_dim_ok(F) || N < 0 || lmax < 1 || lmax > 8 || mode < 0 || mode > 15)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (gn_use_highl(lmax_arg)) return gn_highl_htr_edge(EQ, EK, rl, rowptr, src, N, F, lmax, mode, w_raw, w, st);
    if (mode) return gn_htr_edge_general(EQ, EK, rl, rowptr, src, N, F, lmax, mode, w_raw, w, st);
    const dim3 grid(gn::xcd_grid(N)), block(256);
#define GN_HTR_EDGE(FC)                                                                                                  \
    switch (lmax) {                                                                                                      \
        case 1: hipLaunchKernelGGL((gn::htr_edge_kernel<1, FC>), grid, block, 0, st, EQ, EK, rl, rowptr, src, N, F, w); break; \
        case 2: hipLaunchKernelGGL((gn::htr_edge_kernel<2, FC>), grid, block, 0, st, EQ, EK, rl, rowptr, src, N, F, w); break; \
        case 3: hipLaunchKernelGGL((gn::htr_edge_kernel<3, FC>), grid, block, 0, st, EQ, EK, rl, rowptr, src, N, F, w); break; \
        default: hipLaunchKernelGGL((gn::htr_edge_kernel<4, FC>), grid, block, 0, st, EQ, EK, rl, rowptr, src, N, F, w); break; \
    }
    if (F == 256) { GN_HTR_EDGE(256) } else { GN_HTR_EDGE(0) }
    GN_LAUNCH_CHECK();
    return GN_OK;
}
