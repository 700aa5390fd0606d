// gn_common.h -- shared device helpers for libgotennet_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gotennet_hip.h"

#define GN_WAVE 64

#define GN_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// non-default HTR variants (gn_options.hip), reached through gn_htr_edge / gn_htr_backward with mode != 0
int gn_htr_edge_general(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                        int N, int F, int lmax, int mode, float* w_raw, float* w, hipStream_t st);
int gn_htr_backward_general(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                            const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                            const int* dst, const int* colptr, const int* perm, int N, int F, int lmax, int mode,
                            float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act, hipStream_t st);

namespace gn {

// sigmoid on the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32, ~1 ulp each -- 4 VALU instead of the ~20 of
// expf + an IEEE division.  Round-3 counters: the segment softmax was VALU-bound (1080 VALU instructions per wave, 23 of
// its 29 us), SiLU / SiLU' are a quarter of the VALU work of the HTR and message backward passes.  Error: the argument
// x * log2(e) is rounded to fp32 BEFORE the exponential, so exp carries a relative error of ~6e-8 * |x| on top of the
// 1-ulp instruction error: 3e-7 for |x| <= 4 (every sigmoid argument that matters: beyond, s is within 2e-2 of 0 / 1 and
// the error of SiLU is relative to x), up to ~5e-6 at |x| = 80 (softmax arguments are <= 0 and that far down the weight
// itself is e^-80).  Inside the path's 1e-4 tolerance by more than a decade everywhere; forward and backward use the same
// functions.  A two-term split of x * log2(e) (hi / lo) would remove the |x| term if fp64-level agreement is ever wanted.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896341f * x); }
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float silu(float x) { return x * sigmoid_fast(x); }
// d/dx SiLU(x) = s (1 + x (1 - s)),  s = sigmoid(x)
__device__ __forceinline__ float dsilu(float x) {
    const float s = sigmoid_fast(x);
    return s * (1.0f + x * (1.0f - s));
}
// SiLU and SiLU' of the same argument from ONE sigmoid
__device__ __forceinline__ void silu_pair(float x, float& a, float& d) {
    const float s = sigmoid_fast(x);
    a = x * s;
    d = s * (1.0f + x * (1.0f - s));
}
// Activation kinds of the reference's `activation` argument (layers.py:596-700 str2act); GN_ACT_* in gotennet_hip.h.
// `k` is uniform over a launch; kind 0 (SiLU / swish, the reference default) takes the short path.
__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch: threshold 20
__device__ __forceinline__ float act_generic(float x, int k) {
    switch (k) {
        case GN_ACT_SSP: return softplus_t(x) - 0.69314718055994531f;         // shifted_softplus (layers.py:40-50)
        case GN_ACT_RELU: return fmaxf(x, 0.0f);
        case GN_ACT_TANH: return tanhf(x);
        case GN_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case GN_ACT_ELU: return x > 0.0f ? x : expm1f(x);
        case GN_ACT_SELU: return 1.0507009873554805f * (x > 0.0f ? x : 1.6732632423543772f * expm1f(x));
        case GN_ACT_MISH: return x * tanhf(softplus_t(x));
        case GN_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case GN_ACT_SOFTPLUS: return softplus_t(x);
        case GN_ACT_LEAKY: return x > 0.0f ? x : 0.01f * x;
        case GN_ACT_NONE: return x;
        default: return silu(x);
    }
}
__device__ __forceinline__ float dact_generic(float x, int k) {
    switch (k) {
        case GN_ACT_SSP: case GN_ACT_SOFTPLUS: return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
        case GN_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
        case GN_ACT_TANH: { const float t = tanhf(x); return 1.0f - t * t; }
        case GN_ACT_SIGMOID: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f - s); }
        case GN_ACT_ELU: return x > 0.0f ? 1.0f : expf(x);
        case GN_ACT_SELU: return 1.0507009873554805f * (x > 0.0f ? 1.0f : 1.6732632423543772f * expf(x));
        case GN_ACT_MISH: {
            const float t = tanhf(softplus_t(x)), s = 1.0f / (1.0f + expf(-x));
            return t + x * s * (1.0f - t * t);
        }
        case GN_ACT_GELU:
            return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
        case GN_ACT_LEAKY: return x > 0.0f ? 1.0f : 0.01f;
        case GN_ACT_NONE: return 1.0f;
        default: return dsilu(x);
    }
}
__device__ __forceinline__ float act1(float x, int k) { return k == GN_ACT_SILU ? silu(x) : act_generic(x, k); }
__device__ __forceinline__ float dact1(float x, int k) { return k == GN_ACT_SILU ? dsilu(x) : dact_generic(x, k); }
__device__ __forceinline__ float4 act4(float4 v, int k) {
    if (k == GN_ACT_SILU) return make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w));
    return make_float4(act_generic(v.x, k), act_generic(v.y, k), act_generic(v.z, k), act_generic(v.w, k));
}
__device__ __forceinline__ float4 dact4(float4 v, int k) {
    if (k == GN_ACT_SILU) return make_float4(dsilu(v.x), dsilu(v.y), dsilu(v.z), dsilu(v.w));
    return make_float4(dact_generic(v.x, k), dact_generic(v.y, k), dact_generic(v.z, k), dact_generic(v.w, k));
}
// act4 and dact4 of the same argument (SiLU: one sigmoid per element)
__device__ __forceinline__ void act_pair4(float4 v, int k, float4& a, float4& d) {
    if (k == GN_ACT_SILU) {
        silu_pair(v.x, a.x, d.x); silu_pair(v.y, a.y, d.y); silu_pair(v.z, a.z, d.z); silu_pair(v.w, a.w, d.w);
    } else {
        a = act4(v, k);
        d = dact4(v, k);
    }
}
__device__ __forceinline__ float hsum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// streamed-once rows (per-edge projections): non-temporal, so they do not evict the gathered node tables from L2
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
    f32x4_nt w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4_nt*>(p));
}

__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 fma4(float s, float4 b, float4 c) {
    return make_float4(fmaf(s, b.x, c.x), fmaf(s, b.y, c.y), fmaf(s, b.z, c.z), fmaf(s, b.w, c.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Fixed-order sum over the `ns` slot partials red[s * F + c0 .. c0+3] (deterministic).
__device__ __forceinline__ float4 red4(const float* red, int c0, int F, int ns) {
    float4 s = ld4(red + c0);
    for (int k = 1; k < ns; ++k) s = s + ld4(red + k * F + c0);
    return s;
}

// ---------------------------------------------------------------------------------------- cross-lane exchange
// xor_lane<OFF>(v): the value of lane (id ^ OFF) -- what __shfl_xor(v, OFF) returns, without the LDS crossbar.  hipcc
// lowers __shfl_xor to ds_bpermute_b32 + s_waitcnt lgkmcnt(0) (~100+ cycles of latency per step, and with a run-time width
// a loop of them): the per-edge chains of the backward kernels carried 15 such round trips per edge trip.  On gfx950 every
// power-of-two exchange has a VALU form: quad_perm (1, 2), a row_shl / row_shr pair under bank masks (4), row_ror:8 (8),
// v_permlane16_swap / v_permlane32_swap (16, 32).  Same values as the shuffle: sums built on it keep their bits.
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int OFF>
__device__ __forceinline__ float xor_lane(float v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "power of two below the wave size");
    const int x = __float_as_int(v);
    if constexpr (OFF == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));       // quad_perm:[1,0,3,2]
    else if constexpr (OFF == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));  // quad_perm:[2,3,0,1]
    else if constexpr (OFF == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);        // row_shl:4 -> lanes 0-3, 8-11 of a row read lane + 4
        return __int_as_float(__builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false));   // row_shr:4 -> lanes 4-7, 12-15 read lane - 4
    } else if constexpr (OFF == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true));   // row_ror:8
    else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);   // r[0] = even rows twice, r[1] = odd rows twice
        return __int_as_float((int)((lane_id() & 16) ? r[0] : r[1]));
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);   // r[0] = lower half twice, r[1] = upper half twice
        return __int_as_float((int)((lane_id() & 32) ? r[0] : r[1]));
    }
}
// v + xor_lane<OFF>(v) (commutative: the swap forms need no select)
template <int OFF>
__device__ __forceinline__ float xor_add(float v) {
    if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return (lane_id() & 16) ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (OFF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return (lane_id() & 32) ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else return v + xor_lane<OFF>(v);
}

// Sum over aligned groups of `width` lanes (power of two, <= 64, wave-uniform); every lane gets its group's sum.
__device__ __forceinline__ float group_sum(float v, int width) {
    if (width > 1) v = xor_add<1>(v);
    if (width > 2) v = xor_add<2>(v);
    if (width > 4) v = xor_add<4>(v);
    if (width > 8) v = xor_add<8>(v);
    if (width > 16) v = xor_add<16>(v);
    if (width > 32) v = xor_add<32>(v);
    return v;
}
// Sum over the lanes of a slot and store, for slots that may be WIDER than a wave (F = 512 / 1024: 128 / 256 lanes per
// edge): every 64-lane part of the slot stores its own partial into slice `lp / 64` of the output (slices are `stride`
// floats apart).  Used for the per-edge scalar gradients (g_rl, g_cut), which are written as slices and added in a fixed
// order by gn_edge_geometry_backward anyway -- no cross-wave reduction, no barrier inside the ragged edge loops.
__device__ __forceinline__ void slot_sum_store(float v, int lps, int lp, float* out, size_t stride, bool valid = true) {
    const int w = lps < GN_WAVE ? lps : GN_WAVE;
    const float s = group_sum(v, w);
    if (valid && (lp & (w - 1)) == 0) out[(size_t)(lp >> 6) * stride] = s;
}

// Sum K values (K a power of two, K <= width) over aligned groups of `width` lanes with a
// value-halving butterfly: ~K + log2(width) exchanges instead of K log2(width).  On return the lane
// whose in-group index lp satisfies lp % (width / K) == 0 holds the total of value lp / (width / K) in v[0].
// (every register index below is a compile-time constant: a formulation with a run-time trip count made the
//  compiler index v[] dynamically -- a 16-way v_cmp/v_cndmask chain per element, ~900 VALU instructions per edge.)
template <int LIVE, int OFF, int K>
__device__ __forceinline__ void mgs_halve(float (&v)[K], int lp) {   // LIVE live values -> LIVE / 2, partner = lane ^ OFF
#pragma unroll
    for (int i = 0; i < LIVE / 2; ++i) {
        const float a = v[i], b = v[i + LIVE / 2];
        if constexpr (OFF == 32 || OFF == 16) {
            // lower lanes keep a and need the partner's a, upper lanes keep b and need the partner's b: ONE two-register swap
            // puts each lane's kept value in one result and the received one in the other (keep + received, as below)
            const auto r = OFF == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false)
                                     : __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            v[i] = (lp & OFF) ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else {
            const bool up = (lp & OFF) != 0;
            const float keep = up ? b : a;
            const float send = up ? a : b;
            v[i] = keep + xor_lane<OFF>(send);
        }
    }
}
template <int K, int W>
__device__ __forceinline__ void multi_group_sum_w(float (&v)[K], int lp) {   // compile-time group width W >= K
    if constexpr (K >= 32) mgs_halve<32, W / 2>(v, lp);
    if constexpr (K >= 16) mgs_halve<16, (W >> 1) / (K >= 32 ? 2 : 1)>(v, lp);
    if constexpr (K >= 8) mgs_halve<8, (W >> 1) / (K >= 32 ? 4 : (K >= 16 ? 2 : 1))>(v, lp);
    if constexpr (K >= 4) mgs_halve<4, (W >> 1) / (K >= 32 ? 8 : (K >= 16 ? 4 : (K >= 8 ? 2 : 1)))>(v, lp);
    mgs_halve<2, W / K>(v, lp);
    // the remaining in-group sum over W / K lanes
    if constexpr (W / K > 32) v[0] = xor_add<32>(v[0]);
    if constexpr (W / K > 16) v[0] = xor_add<16>(v[0]);
    if constexpr (W / K > 8) v[0] = xor_add<8>(v[0]);
    if constexpr (W / K > 4) v[0] = xor_add<4>(v[0]);
    if constexpr (W / K > 2) v[0] = xor_add<2>(v[0]);
    if constexpr (W / K > 1) v[0] = xor_add<1>(v[0]);
}
template <int K>
__device__ __forceinline__ void multi_group_sum(float (&v)[K], int width, int lp) {   // requires width >= K (wave-uniform)
    static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "K must be a power of two <= 32");
    if (width == 64) { multi_group_sum_w<K, 64>(v, lp); return; }
    if constexpr (K <= 32) if (width == 32) { multi_group_sum_w<K, 32>(v, lp); return; }
    if constexpr (K <= 16) if (width == 16) { multi_group_sum_w<K, 16>(v, lp); return; }
    if constexpr (K <= 8) if (width == 8) { multi_group_sum_w<K, 8>(v, lp); return; }
    if constexpr (K <= 4) if (width == 4) { multi_group_sum_w<K, 4>(v, lp); return; }
    if constexpr (K <= 2) if (width == 2) { multi_group_sum_w<K, 2>(v, lp); return; }
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, xor_lane<1>(v)); v = fmaxf(v, xor_lane<2>(v)); v = fmaxf(v, xor_lane<4>(v));
    v = fmaxf(v, xor_lane<8>(v)); v = fmaxf(v, xor_lane<16>(v)); v = fmaxf(v, xor_lane<32>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, GN_WAVE); }
// reductions over the lanes that share (lane % stride): the exchanges at offsets stride, 2 stride, ..., 32 (stride a power of two)
__device__ __forceinline__ float stride_sum(float v, int stride) {
    if (stride <= 1) v = xor_add<1>(v);
    if (stride <= 2) v = xor_add<2>(v);
    if (stride <= 4) v = xor_add<4>(v);
    if (stride <= 8) v = xor_add<8>(v);
    if (stride <= 16) v = xor_add<16>(v);
    if (stride <= 32) v = xor_add<32>(v);
    return v;
}
__device__ __forceinline__ float stride_max(float v, int stride) {
    if (stride <= 1) v = fmaxf(v, xor_lane<1>(v));
    if (stride <= 2) v = fmaxf(v, xor_lane<2>(v));
    if (stride <= 4) v = fmaxf(v, xor_lane<4>(v));
    if (stride <= 8) v = fmaxf(v, xor_lane<8>(v));
    if (stride <= 16) v = fmaxf(v, xor_lane<16>(v));
    if (stride <= 32) v = fmaxf(v, xor_lane<32>(v));
    return v;
}

// Slot geometry of the per-node kernels (256 threads; an edge row of F floats is covered by lps = F / 4 lanes, ns = 256 / lps
// rows per trip).  FC = 0: F is a run-time value.  FC = 256 (the reference's width, every benchmarked model): the width is a
// compile-time constant and a slot IS a wave, so the slot index is wave-uniform (readfirstlane): edge indices, per-edge
// scalars and row bases become scalar loads and SGPR arithmetic, and every width-dependent branch folds.  Round 5, merged
// message backward alone: 262 -> 196 us.  Expects `F` in scope (const int F = FC ? FC : <run-time F>).
#define GN_SLOT_GEOMETRY(FC)                                                                                          \
    const int lps = F >> 2, ns = 256 / lps;                                                                           \
    const int slot = (FC) == 256 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)threadIdx.x / lps;  \
    const int lp = threadIdx.x % lps, c0 = lp * 4

// XCD-aware block -> work-item map.  Blocks are dealt round-robin to the 8 XCDs
// (block b -> XCD b % 8, observed, speed only); give every XCD a contiguous range of
// items so the neighbours of a molecule are gathered through ONE L2.
// grid must be 8 * ceil(n / 8); returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_item(int b, int n) {
    const int per = (n + 7) >> 3;
    const int i = (b & 7) * per + (b >> 3);
    return ((b >> 3) < per && i < n) ? i : -1;
}
static inline int xcd_grid(int n) { return 8 * ((n + 7) / 8); }

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace gn
