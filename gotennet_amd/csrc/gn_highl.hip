// gn_highl.hip -- degree-sliced message / HTR kernels for lmax = 5..8 (D = 35..80 rows per atom).
//
// The tuned kernels (gn_gata.hip, gn_backward.hip, gn_options.hip) pin the (1 + D) accumulator rows of a target --
// and in the backward the D gradient rows too -- in registers; at D = 80 that is 320+ VGPRs per array.  The
// reference instantiates TensorInit up to l = 8 (layers.py:805-1494), so this file serves those degrees with the same
// slot layout and the same fixed-order LDS reductions (no atomics, bit-reproducible), cut the other way:
//   * one launch (or one in-kernel pass) per DEGREE, with the 2l+1 <= 17 rows of that degree as accumulators;
//   * everything that is not an accumulator (the target's gradient rows, EQ_i, the source's own rows) is re-read
//     per edge through L1/L2 instead of being held in registers;
//   * lmax, sep_dir, sep_tensor and the HTR mode bits are runtime values: 9 + 8 + 8 + 8 + 4 kernels cover every
//     (lmax, flag) combination, where templates over (LMAX, flags, group) would need a few hundred.
// The per-row arithmetic is the tuned kernels' (same expressions, same order inside a row), so at lmax <= 4 the
// message outputs are bit-identical to theirs (tests/test_hip_highl.py runs both with GN_FORCE_HIGHL=1).
// Speed is not the point here -- nobody benchmarks l > 4 -- correctness and coverage are.
//
// Forward equations: gotennet.py:452-559 (message), 351-364 + 561-611 (HTR); backward as in gn_backward.hip.
#include <stdlib.h>
#include "gn_common.h"
#include "gn_highl.h"

namespace gn {

// value-vector blocks (gotennet.py:516-529): 0 scalar | ND direction gates | NT tensor gates
struct HlShape {
    int lmax, D, ND, NT, M, sd, st;
    __host__ __device__ HlShape(int lmax_, int sd_, int st_)
        : lmax(lmax_), D((lmax_ + 1) * (lmax_ + 1) - 1), ND(sd_ ? lmax_ : 1), NT(st_ ? lmax_ : 1),
          M(1 + (sd_ ? lmax_ : 1) + (st_ ? lmax_ : 1)), sd(sd_), st(st_) {}
    __device__ bool is_dir(int b) const { return b >= 1 && b < 1 + ND; }
    __device__ int lo(int b) const { return is_dir(b) ? (sd ? b : 1) : (st ? b - ND : 1); }     // degrees served by gate block b
    __device__ int hi(int b) const { return is_dir(b) ? (sd ? b : lmax) : (st ? b - ND : lmax); }
    __device__ int dir_block(int l) const { return sd ? l : 1; }
    __device__ int ten_block(int l) const { return 1 + ND + (st ? l - 1 : 0); }
};
constexpr int HL_MAX_M = 17;                       // 1 + 2 * 8

// ------------------------------------------------------------------ message forward, one degree per launch (L = 0: scalar row)
// AGGR: the reference's `aggr` (PyG scatter reduce of the per-edge messages, gotennet.py:638-639): 0 "add" (what every
// config uses), 1 "mean" = sum / in-degree, 2 "max" = element-wise maximum over the incoming edges; atoms without incoming
// edges get 0 in all three.  The fixed slot order makes the maximum as reproducible as the sums.
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
template <int L, int AGGR>
__global__ __launch_bounds__(256) void hl_msg_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ v, int ldxv, const float* __restrict__ tf, int ldt,
    const float* __restrict__ a, const float* __restrict__ rl, const float* __restrict__ cut,
    const int* __restrict__ rowptr, const int* __restrict__ src, const float* __restrict__ h_in,
    const float* __restrict__ X_in, float* __restrict__ h_out, float* __restrict__ X_out, int N, int F, int H,
    const HlShape S) {
    constexpr int ROWS = L == 0 ? 1 : 2 * L + 1;
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    constexpr int M0 = L == 0 ? 0 : L * L - 1;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int D = S.D, per_head = (S.M * F) / H;
    const int bd = L == 0 ? 0 : S.dir_block(L), bt = L == 0 ? 0 : S.ten_block(L);
    const int hd = (bd * F + c0) / per_head, ht = (bt * F + c0) / per_head;
    const float NEG = -INFINITY;

    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = AGGR == 2 ? make_float4(NEG, NEG, NEG, NEG) : zero4();
    for (int e = e0 + slot; e < e1; e += ns) {
        const int j = src[e];
        const float ce = cut[e];
        const float* xr = x + (size_t)j * ldxv + c0;
        const float* vr = v + (size_t)j * ldxv + c0;
        const float* tr = tf + (size_t)e * ldt + c0;
        const float* ar = a + (size_t)e * H;
        // gotennet.py:516-529: (t_filter * x_j) * cutoff + attn * v_j, block b of the value vector
        auto gate = [&](int b, int hb) {
            const float4 sp = (ld4_nt(tr + b * F) * ld4(xr + b * F)) * ce;
            return fma4(ar[hb], ld4(vr + b * F), sp);
        };
        if constexpr (L == 0) {
            const float4 c = gate(0, hd);
            acc[0] = AGGR == 2 ? max4(acc[0], c) : acc[0] + c;
        } else {
            const float* Xj = X_in + (size_t)j * D * F + c0;
            const float* re = rl + (size_t)e * D;
            const float4 gd = gate(bd, hd), gt = gate(bt, ht);
#pragma unroll
            for (int mm = 0; mm < ROWS; ++mm) {    // gotennet.py:538-558: rl * o_d + X_j * o_t
                const float4 c = fma4(ld4(Xj + (size_t)(M0 + mm) * F), gt, gd * re[M0 + mm]);
                acc[mm] = AGGR == 2 ? max4(acc[mm], c) : acc[mm] + c;
            }
        }
    }
    auto write = [&](int row, float4 s) {
        if constexpr (AGGR == 1) s = s * (1.0f / (float)(e1 > e0 ? e1 - e0 : 1));
        if constexpr (AGGR == 2) { if (e1 == e0) s = zero4(); }
        if constexpr (L == 0) {
            st4(h_out + (size_t)i * F + c0, ld4(h_in + (size_t)i * F + c0) + s);
        } else {
            const size_t off = ((size_t)i * D + (M0 + row)) * F + c0;
            st4(X_out + off, ld4(X_in + off) + s);
        }
    };
    if constexpr (AGGR != 2) {
        reduce_rows<ROWS>(acc, red, slot, c0, F, ns, write);
    } else {                                         // the same fixed-order pass over the slots, with max for +
#pragma unroll
        for (int base = 0; base < ROWS; base += CH) {
            if (base) __syncthreads();
#pragma unroll
            for (int r = 0; r < CH; ++r)
                if (base + r < ROWS) st4(&red[r * 1024 + slot * F + c0], acc[base + r]);
            __syncthreads();
            for (int r = slot; r < CH && base + r < ROWS; r += ns) {
                float4 s = ld4(red + r * 1024 + c0);
                for (int k = 1; k < ns; ++k) s = max4(s, ld4(red + r * 1024 + k * F + c0));
                write(base + r, s);
            }
        }
    }
}

// gradient of gate block b for edge (i <- j):  go_b = sum over the degrees the block serves of
//   direction gate:  sum_m rl[m] gX_i[m]        tensor gate:  sum_m gX_i[m] X_j[m]
__device__ __forceinline__ float4 hl_gate_grad(const HlShape& S, int b, const float* __restrict__ gXi,
                                               const float* __restrict__ Xj, const float* __restrict__ re, int F) {
    float4 go = zero4();
    const bool dir = S.is_dir(b);
    const int m_lo = S.lo(b) * S.lo(b) - 1, m_hi = (S.hi(b) + 1) * (S.hi(b) + 1) - 1;
    for (int m = m_lo; m < m_hi; ++m) {
        const float4 gx = ld4(gXi + (size_t)m * F);
        go = dir ? fma4(re[m], gx, go) : fma4(gx, ld4(Xj + (size_t)m * F), go);
    }
    return go;
}

// ------------------------------------------------------------------ aggr = "max": route the upstream gradient to the arg-max edges
// The reference reduces the per-edge messages with scatter(..., reduce="max") (gotennet.py:638-639; torch amax): the
// gradient of an output element goes to the message(s) that attain the maximum, split evenly among exact ties.  One
// workgroup per target, one output row at a time: the messages are recomputed (same expressions as hl_msg_fwd_kernel)
// three times -- maximum, tie count, routed write -- and the result is the [E, 1 + D, F] array of per-MESSAGE gradients
// that the three backward kernels then read in place of the per-target rows g_h1 / g_X1.
__global__ __launch_bounds__(256) void hl_max_route_kernel(const MsgBwdArgs p, const HlShape S, float* __restrict__ g_edge) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int N = p.N, F = p.F, H = p.H, D = S.D, M = S.M;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = p.rowptr[i], e1 = p.rowptr[i + 1];
    if (e1 == e0) return;                            // (uniform: no incoming edge, nothing to route)
    const int per_head = (M * F) / H;
    const size_t rowsz = (size_t)(1 + D) * F;
    const float NEG = -INFINITY;
    for (int r = 0; r <= D; ++r) {                   // r = 0: the scalar row; r >= 1: tensor row m = r - 1 of degree l
        int l = 0;
        if (r) { l = 1; while ((l + 1) * (l + 1) - 1 <= r - 1) ++l; }
        const int bd = r ? S.dir_block(l) : 0, bt = r ? S.ten_block(l) : 0;
        const int hd = (bd * F + c0) / per_head, ht = (bt * F + c0) / per_head;
        auto msg = [&](int e) -> float4 {
            const int j = p.src[e];
            const float ce = p.cut[e];
            const float* xr = p.x + (size_t)j * p.ldxv + c0;
            const float* vr = p.v + (size_t)j * p.ldxv + c0;
            const float* tr = p.eproj + (size_t)e * p.lde + F + c0;
            const float* ar = p.a + (size_t)e * H;
            auto gate = [&](int b, int hb) {
                const float4 sp = (ld4(tr + b * F) * ld4(xr + b * F)) * ce;
                return fma4(ar[hb], ld4(vr + b * F), sp);
            };
            if (r == 0) return gate(0, hd);
            const float4 gd = gate(bd, hd), gt = gate(bt, ht);
            return fma4(ld4(p.X_in + ((size_t)j * D + (r - 1)) * F + c0), gt, gd * p.rl[(size_t)e * D + (r - 1)]);
        };
        float4 mx = make_float4(NEG, NEG, NEG, NEG);
        for (int e = e0 + slot; e < e1; e += ns) mx = max4(mx, msg(e));
        st4(&red[slot * F + c0], mx);
        __syncthreads();
        mx = ld4(red + c0);
        for (int k = 1; k < ns; ++k) mx = max4(mx, ld4(red + k * F + c0));
        __syncthreads();
        float4 cnt = zero4();
        for (int e = e0 + slot; e < e1; e += ns) {
            const float4 c = msg(e);
            cnt.x += c.x == mx.x ? 1.f : 0.f; cnt.y += c.y == mx.y ? 1.f : 0.f;
            cnt.z += c.z == mx.z ? 1.f : 0.f; cnt.w += c.w == mx.w ? 1.f : 0.f;
        }
        st4(&red[slot * F + c0], cnt);
        __syncthreads();
        cnt = red4(red, c0, F, ns);
        __syncthreads();
        float4 g = r ? ld4(p.g_X1 + ((size_t)i * D + (r - 1)) * F + c0) : ld4(p.g_h1 + (size_t)i * F + c0);
        g.x = cnt.x > 0.f ? g.x / cnt.x : 0.f; g.y = cnt.y > 0.f ? g.y / cnt.y : 0.f;
        g.z = cnt.z > 0.f ? g.z / cnt.z : 0.f; g.w = cnt.w > 0.f ? g.w / cnt.w : 0.f;
        for (int e = e0 + slot; e < e1; e += ns) {
            const float4 c = msg(e);
            st4(g_edge + (size_t)e * rowsz + (size_t)r * F + c0,
                make_float4(c.x == mx.x ? g.x : 0.f, c.y == mx.y ? g.y : 0.f, c.z == mx.z ? g.z : 0.f, c.w == mx.w ? g.w : 0.f));
        }
    }
}

// ------------------------------------------------------------------ message backward, by target (all degrees, one launch)
// per edge: g_tf, g_cut, g_rl, head sums of g_a; then softmax backward and the score gradients (g_ta, g_q)
__global__ __launch_bounds__(256) void hl_msg_bwd_target_kernel(const MsgBwdArgs p, const HlShape S) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    __shared__ float hsum[256 * HL_MAX_M];
    const int N = p.N, F = p.F, H = p.H, D = S.D, M = S.M;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, lp = threadIdx.x % lps, c0 = lp * 4;
    const int e0 = p.rowptr[i], e1 = p.rowptr[i + 1];
    const int per_head = (M * F) / H;
    // aggr = "mean": every message of this target carries 1 / in-degree (the residual paths do not)
    const float inv = p.mean ? 1.0f / (float)(e1 > e0 ? e1 - e0 : 1) : 1.0f;
    const float4 gdh_t = ld4(p.g_h1 + (size_t)i * F + c0) * inv;
    const float* gXi_t = p.g_X1 + (size_t)i * D * F + c0;

    // Slots wider than a wave (F = 512 / 1024): the per-edge scalars go out as one partial slice per 64-lane part
    // (slot_sum_store), and the head-sum staging row is shared by the slot's waves, so every slot runs the same number of
    // trips (a slot past the end repeats the last edge without storing) and two barriers fence the row.
    const bool wide = lps > GN_WAVE;
    const size_t E_all = (size_t)p.rowptr[N];
    const int trips = (e1 - e0 + ns - 1) / ns;
    for (int it = 0; it < trips; ++it) {
        const int e_raw = e0 + it * ns + slot;
        const bool valid = e_raw < e1;
        const int e = valid ? e_raw : e1 - 1;
        const int j = p.src[e];
        const float ce = p.cut[e];
        // aggr = "max": this edge's own routed gradient rows instead of the target's
        const float* ge = p.g_edge ? p.g_edge + (size_t)e * (1 + D) * F + c0 : nullptr;
        const float4 gdh = ge ? ld4(ge) : gdh_t;
        const float* gXi = ge ? ge + F : gXi_t;
        const float* xr = p.x + (size_t)j * p.ldxv + c0;
        const float* vr = p.v + (size_t)j * p.ldxv + c0;
        const float* tr = p.eproj + (size_t)e * p.lde + F + c0;
        float* gtr = p.g_eproj + (size_t)e * p.lde + F + c0;
        const float* ar = p.a + (size_t)e * H;
        const float* Xj = p.X_in + (size_t)j * D * F + c0;
        const float* re = p.rl + (size_t)e * D;
        float* hrow = hsum + slot * (M * lps);
        float cutp = 0.f;
        for (int b = 0; b < M; ++b) {
            const float4 go = b == 0 ? gdh : hl_gate_grad(S, b, gXi, Xj, re, F) * inv;
            const float4 tfb = ld4_nt(tr + b * F), xb = ld4(xr + b * F), vb = ld4(vr + b * F);
            if (valid) st4_nt(gtr + b * F, (go * xb) * ce);
            cutp += hsum4(go * tfb * xb);
            hrow[b * lps + lp] = hsum4(go * vb);
            if (S.is_dir(b)) {
                const float4 od = fma4(ar[(b * F + c0) / per_head], vb, (tfb * xb) * ce);      // forward direction gate
                const int m_lo = S.lo(b) * S.lo(b) - 1, m_hi = (S.hi(b) + 1) * (S.hi(b) + 1) - 1;
                for (int m = m_lo; m < m_hi; ++m) {
                    slot_sum_store(hsum4(ld4(gXi + (size_t)m * F) * od) * inv, lps, lp, p.g_rl + (size_t)e * D + m, E_all * D, valid);
                }
            }
        }
        slot_sum_store(cutp, lps, lp, p.g_cut + e, E_all, valid);
        // head sums (same staging as msg_bwd_target_body; a slot within one wave: wave-ordered LDS accesses, no barrier)
        if (wide) __syncthreads();
        {
            const int rpl = lps / H, hh = lp / rpl, part = lp - hh * rpl;
            const float* hp = hrow + hh * (M * rpl) + part * M;
            float hv = hp[0];
            for (int k = 1; k < M; ++k) hv += hp[k];
            hv = group_sum(hv, rpl);
            if (valid && part == 0) p.g_s[(size_t)e * H + hh] = hv;
        }
        if (wide) __syncthreads();
    }
    __syncthreads();
    // softmax backward per head:  g_s = a g_a - (a / nrm) sum_e' a g_a
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int h = wave; h < H; h += 4) {
            float dot = 0.f;
            for (int e = e0 + lane; e < e1; e += 64) dot += p.a[(size_t)e * H + h] * p.g_s[(size_t)e * H + h];
            dot = wave_sum(dot);
            for (int e = e0 + lane; e < e1; e += 64) {
                const float nrm = p.outdeg ? sqrtf((float)p.outdeg[p.src[e]]) * p.inv_sqrt_f : p.inv_sqrt_f;
                const float av = p.a[(size_t)e * H + h];
                p.g_s[(size_t)e * H + h] = av * p.g_s[(size_t)e * H + h] - (av / nrm) * dot;
            }
        }
    }
    __syncthreads();
    // scores backward: g_ta (pre-activation), g_q
    const int hq = c0 / (F / H);
    const float4 qi = ld4(p.qk + (size_t)i * p.ldqk + c0);
    float4 gq = zero4();
    for (int e = e0 + slot; e < e1; e += ns) {
        const float gs = p.g_s[(size_t)e * H + hq];
        const float4 kj = ld4(p.qk + (size_t)p.src[e] * p.ldqk + F + c0);
        const float4 pta = ld4_nt(p.eproj + (size_t)e * p.lde + c0);
        gq = fma4(gs, kj * act4(pta, p.act), gq);
        st4_nt(p.g_eproj + (size_t)e * p.lde + c0, ((qi * kj) * gs) * dact4(pta, p.act));
    }
    st4(&red[slot * F + c0], gq);
    __syncthreads();
    if (slot == 0) st4(p.g_nproj + (size_t)i * p.ldn + c0, red4(red, c0, F, ns));
}

// ------------------------------------------------------------------ message backward, by source: g_x, g_v (one pass per
// value block) and g_k
__global__ __launch_bounds__(256) void hl_msg_bwd_source_gates_kernel(const MsgBwdArgs p, const HlShape S) {
    __shared__ __attribute__((aligned(16))) float red[2 * 1024];
    const int N = p.N, F = p.F, H = p.H, D = S.D, M = S.M;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int p0 = p.colptr[j], p1 = p.colptr[j + 1];
    const int per_head = (M * F) / H;
    const float* Xj = p.X_in + (size_t)j * D * F + c0;
    for (int b = 0; b < M; ++b) {
        float4 acc[2] = {zero4(), zero4()};
        const int hb = (b * F + c0) / per_head;
        for (int pp = p0 + slot; pp < p1; pp += ns) {
            const int e = p.perm[pp], i = p.dst[pp];
            const float ce = p.cut[e];
            const float4 tfb = ld4_nt(p.eproj + (size_t)e * p.lde + F + c0 + b * F);
            const float ab = p.a[(size_t)e * H + hb];
            const float* ge = p.g_edge ? p.g_edge + (size_t)e * (1 + D) * F + c0 : nullptr;     // aggr = "max": routed rows
            float4 go = b == 0 ? ld4(ge ? ge : p.g_h1 + (size_t)i * F + c0)
                               : hl_gate_grad(S, b, ge ? ge + F : p.g_X1 + (size_t)i * D * F + c0, Xj, p.rl + (size_t)e * D, F);
            if (p.mean) go = go * (1.0f / (float)(p.rowptr[i + 1] - p.rowptr[i]));      // (edge e exists: in-degree >= 1)
            acc[0] = fma4(go, tfb * ce, acc[0]);
            acc[1] = fma4(ab, go, acc[1]);
        }
        if (b) __syncthreads();                      // the previous pass's readers are done with `red`
        reduce_rows<2>(acc, red, slot, c0, F, ns, [&](int row, float4 s) {
            st4((row == 0 ? p.g_x : p.g_v) + (size_t)j * p.ldxv + b * F + c0, s);
        });
    }
    const int hq = c0 / (F / H);
    float4 gk[1] = {zero4()};
    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = p.perm[pp], i = p.dst[pp];
        const float gs = p.g_s[(size_t)e * H + hq];
        const float4 qi = ld4(p.qk + (size_t)i * p.ldqk + c0);
        const float4 ta = act4(ld4_nt(p.eproj + (size_t)e * p.lde + c0), p.act);
        gk[0] = fma4(gs, qi * ta, gk[0]);
    }
    __syncthreads();
    reduce_rows<1>(gk, red, slot, c0, F, ns, [&](int, float4 s) { st4(p.g_nproj + (size_t)j * p.ldn + F + c0, s); });
}

// g_X of the gathered source rows (tensor-gate path), one degree per launch:  g_X_j[m] = g_X1_j[m] + sum_e gX_i[m] o_t
template <int L>
__global__ __launch_bounds__(256) void hl_msg_bwd_source_X_kernel(const MsgBwdArgs p, const HlShape S) {
    constexpr int ROWS = 2 * L + 1, M0 = L * L - 1;
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int N = p.N, F = p.F, H = p.H, D = S.D;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int p0 = p.colptr[j], p1 = p.colptr[j + 1];
    const int b = S.ten_block(L), hb = (b * F + c0) / ((S.M * F) / H);
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();
    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = p.perm[pp], i = p.dst[pp];
        const float4 tfb = ld4_nt(p.eproj + (size_t)e * p.lde + F + c0 + b * F);
        const float4 ot = fma4(p.a[(size_t)e * H + hb], ld4(p.v + (size_t)j * p.ldxv + b * F + c0),
                               (tfb * ld4(p.x + (size_t)j * p.ldxv + b * F + c0)) * p.cut[e]);   // forward tensor gate
        const float* gXi = p.g_edge ? p.g_edge + ((size_t)e * (1 + D) + 1 + M0) * F + c0 : p.g_X1 + ((size_t)i * D + M0) * F + c0;
        const float4 otm = p.mean ? ot * (1.0f / (float)(p.rowptr[i + 1] - p.rowptr[i])) : ot;
#pragma unroll
        for (int mm = 0; mm < ROWS; ++mm) acc[mm] = fma4(ld4(gXi + (size_t)mm * F), otm, acc[mm]);
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns, [&](int row, float4 s) {
        const size_t off = ((size_t)j * D + (M0 + row)) * F + c0;
        st4(p.g_X_out + off, ld4(p.g_X1 + off) + s);
    });
}

// ------------------------------------------------------------------ HTR edge weights, all degrees and modes, one launch
// per block B (a degree, or all D rows when `joint`):  w += A.B - (2 - r.r)(A.r)(B.r)  [rejection on];  w += A.B  [off]
__global__ __launch_bounds__(256) void hl_htr_edge_kernel(
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F, int lmax, int joint, int rej, int gate,
    float* __restrict__ w_raw, float* __restrict__ w) {
    const int D = (lmax + 1) * (lmax + 1) - 1;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const float* qi = EQ + (size_t)i * D * F + c0;
    for (int e = e0 + slot; e < e1; e += ns) {
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 wsum = zero4(), ab = zero4(), pa = zero4(), pb = zero4();
        float rr = 0.f;
        int left = 3, l = 1;                         // rows left in the current degree
        for (int m = 0; m < D; ++m) {
            const float4 eq = ld4(qi + (size_t)m * F), ek = ld4(kj + (size_t)m * F);
            const float r = re[m];
            ab = fma4(eq, ek, ab);
            pa = fma4(r, eq, pa);
            pb = fma4(r, ek, pb);
            rr = fmaf(r, r, rr);
            if (--left == 0) {
                ++l;
                left = 2 * l + 1;
                if (!joint || m == D - 1) {
                    wsum = wsum + (rej ? ab - (pa * pb) * (2.0f - rr) : ab);
                    ab = pa = pb = zero4();
                    rr = 0.f;
                }
            }
        }
        if (w_raw) st4(w_raw + (size_t)e * F + c0, wsum);
        st4(w + (size_t)e * F + c0, gate4(wsum, gate));
    }
}

// projections (A.r), (B.r) and r.r of the block row m belongs to: degree L's rows, or all D rows when joint
__device__ __forceinline__ void hl_block_proj(const float* __restrict__ qi, const float* __restrict__ kj,
                                              const float* __restrict__ re, int lo, int hi, int F,
                                              float4& pa, float4& pb, float& rr) {
    pa = pb = zero4();
    rr = 0.f;
    for (int m = lo; m < hi; ++m) {
        const float r = re[m];
        pa = fma4(r, ld4(qi + (size_t)m * F), pa);
        if (kj) pb = fma4(r, ld4(kj + (size_t)m * F), pb);
        rr = fmaf(r, r, rr);
    }
}

// ------------------------------------------------------------------ HTR backward by target, one degree per launch
// dw/dA_m = B_m - c r_m (B.r);  dw/dr_m = -c [A_m (B.r) + (A.r) B_m] + 2 r_m (A.r)(B.r),  c = 2 - r.r
template <int L>
__global__ __launch_bounds__(256) void hl_htr_bwd_target_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w,
    const float* __restrict__ w_raw, const float* __restrict__ EQ, const float* __restrict__ EK,
    const float* __restrict__ rl, const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F, int lmax,
    int joint, int rej, int gate, int direct, float* __restrict__ gEQ, float* __restrict__ g_rl,
    float* __restrict__ g_pre_t, int act) {
    constexpr int ROWS = 2 * L + 1, M0 = L * L - 1;
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int D = (lmax + 1) * (lmax + 1) - 1;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, lp = threadIdx.x % lps, c0 = lp * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const float* qi = EQ + (size_t)i * D * F + c0;
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();
    for (int e = e0 + slot; e < e1; e += ns) {
        float4 gw = ld4(gtp + (size_t)e * F + c0);       // direct: this already is dL/dw
        if (!direct) {
            const float4 pte = ld4(pre_t + (size_t)e * F + c0);
            // t' = t + act(pre_t) * g(w):  d/d pre_t, written once (by the degree-1 launch)
            if (L == 1) st4(g_pre_t + (size_t)e * F + c0, gw * ld4(w + (size_t)e * F + c0) * dact4(pte, act));
            gw = gw * act4(pte, act);
            if (gate) gw = gw * dgate4(ld4(w_raw + (size_t)e * F + c0), gate);
        }
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 pa, pb;
        float rr;
        hl_block_proj(qi, kj, re, joint ? 0 : M0, joint ? D : M0 + ROWS, F, pa, pb, rr);
        const float c = rej ? 2.0f - rr : 0.0f;
        const float4 papb = pa * pb;
#pragma unroll
        for (int mm = 0; mm < ROWS; ++mm) {
            const int m = M0 + mm;
            const float4 eq = ld4(qi + (size_t)m * F), ek = ld4(kj + (size_t)m * F);
            const float r = re[m];
            acc[mm] = fma4(gw, ek + pb * (-c * r), acc[mm]);
            const float4 t4 = gw * ((eq * pb + pa * ek) * (-c) + papb * (rej ? 2.0f * r : 0.0f));
            slot_sum_store(hsum4(t4), lps, lp, g_rl + (size_t)e * D + m, (size_t)rowptr[N] * D);
        }
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns,
                      [&](int row, float4 s) { st4(gEQ + ((size_t)i * D + (M0 + row)) * F + c0, s); });
}

// ------------------------------------------------------------------ HTR backward by source:  dw/dB_m = A_m - c r_m (A.r)
template <int L>
__global__ __launch_bounds__(256) void hl_htr_bwd_source_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w_raw,
    const float* __restrict__ EQ, const float* __restrict__ rl, const int* __restrict__ colptr,
    const int* __restrict__ perm, const int* __restrict__ dst, int N, int F, int lmax, int joint, int rej, int gate,
    int direct, float* __restrict__ gEK, int act) {
    constexpr int ROWS = 2 * L + 1, M0 = L * L - 1;
    constexpr int CH = ROWS < 9 ? ROWS : 9;
    __shared__ __attribute__((aligned(16))) float red[CH * 1024];
    const int D = (lmax + 1) * (lmax + 1) - 1;
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int p0 = colptr[j], p1 = colptr[j + 1];
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = zero4();
    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = perm[pp];
        float4 gw = ld4(gtp + (size_t)e * F + c0);
        if (!direct) {
            gw = gw * act4(ld4(pre_t + (size_t)e * F + c0), act);
            if (gate) gw = gw * dgate4(ld4(w_raw + (size_t)e * F + c0), gate);
        }
        const float* qi = EQ + (size_t)dst[pp] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 pa, pb;
        float rr;
        hl_block_proj(qi, nullptr, re, joint ? 0 : M0, joint ? D : M0 + ROWS, F, pa, pb, rr);
        const float c = rej ? 2.0f - rr : 0.0f;
#pragma unroll
        for (int mm = 0; mm < ROWS; ++mm)
            acc[mm] = fma4(gw, ld4(qi + (size_t)(M0 + mm) * F) + pa * (-c * re[M0 + mm]), acc[mm]);
    }
    reduce_rows<ROWS>(acc, red, slot, c0, F, ns,
                      [&](int row, float4 s) { st4(gEK + ((size_t)j * D + (M0 + row)) * F + c0, s); });
}

}  // namespace gn

// ====================================================================================== launchers
bool gn_use_highl(int lmax_arg) { return (lmax_arg & 0xff) > 4 || (lmax_arg & (GN_LMAX_SLICED | GN_LMAX_MEAN | GN_LMAX_MAX)) != 0; }

// KERNEL<L> for L = 1..lmax (lmax <= 8), one launch per degree
#define GN_HL_FWD_DEGREES(AG, ...)                                                                    \
    do {                                                                                              \
        hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<0, AG>), grid, block, 0, st, __VA_ARGS__);           \
        hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<1, AG>), grid, block, 0, st, __VA_ARGS__);           \
        if (lmax >= 2) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<2, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 3) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<3, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 4) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<4, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 5) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<5, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 6) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<6, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 7) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<7, AG>), grid, block, 0, st, __VA_ARGS__); \
        if (lmax >= 8) hipLaunchKernelGGL((gn::hl_msg_fwd_kernel<8, AG>), grid, block, 0, st, __VA_ARGS__); \
    } while (0)

#define GN_HL_PER_DEGREE(KERNEL, ...)                                                                 \
    do {                                                                                              \
        hipLaunchKernelGGL(gn::KERNEL<1>, grid, block, 0, st, __VA_ARGS__);                            \
        if (lmax >= 2) hipLaunchKernelGGL(gn::KERNEL<2>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 3) hipLaunchKernelGGL(gn::KERNEL<3>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 4) hipLaunchKernelGGL(gn::KERNEL<4>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 5) hipLaunchKernelGGL(gn::KERNEL<5>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 6) hipLaunchKernelGGL(gn::KERNEL<6>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 7) hipLaunchKernelGGL(gn::KERNEL<7>, grid, block, 0, st, __VA_ARGS__);             \
        if (lmax >= 8) hipLaunchKernelGGL(gn::KERNEL<8>, grid, block, 0, st, __VA_ARGS__);             \
    } while (0)

int gn_highl_message(const float* x, const float* v, int ldxv, const float* t_filter, int ldt, const float* a,
                     const float* rl, const float* cut, const int* rowptr, const int* src, const float* h_in,
                     const float* X_in, float* h_out, float* X_out, int N, int F, int H, int lmax, int sep_dir,
                     int sep_tensor, int aggr, hipStream_t st) {
    if (lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    const gn::HlShape S(lmax, sep_dir, sep_tensor);
    const dim3 grid(gn::xcd_grid(N)), block(256);
    if (aggr == 1)
        GN_HL_FWD_DEGREES(1, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, h_out, X_out, N, F, H, S);
    else if (aggr == 2)
        GN_HL_FWD_DEGREES(2, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, h_out, X_out, N, F, H, S);
    else
        GN_HL_FWD_DEGREES(0, x, v, ldxv, t_filter, ldt, a, rl, cut, rowptr, src, h_in, X_in, h_out, X_out, N, F, H, S);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

int gn_highl_message_backward(const gn::MsgBwdArgs& p, int lmax, int sep_dir, int sep_tensor, hipStream_t st) {
    if (lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    const gn::HlShape S(lmax, sep_dir, sep_tensor);
    const dim3 grid(gn::xcd_grid(p.N)), block(256);
    if (p.g_edge)                                    // aggr = "max": per-message gradients first (p.g_edge is the caller's workspace)
        hipLaunchKernelGGL(gn::hl_max_route_kernel, grid, block, 0, st, p, S, const_cast<float*>(p.g_edge));
    hipLaunchKernelGGL(gn::hl_msg_bwd_target_kernel, grid, block, 0, st, p, S);      // g_s first: the source pass reads it
    hipLaunchKernelGGL(gn::hl_msg_bwd_source_gates_kernel, grid, block, 0, st, p, S);
    GN_HL_PER_DEGREE(hl_msg_bwd_source_X_kernel, p, S);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

int gn_highl_htr_edge(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                      int N, int F, int lmax, int mode, float* w_raw, float* w, hipStream_t st) {
    if (lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    const int joint = mode & GN_HTR_JOINT ? 1 : 0, rej = mode & GN_HTR_NOREJ ? 0 : 1, gate = (mode >> 2) & 3;
    hipLaunchKernelGGL(gn::hl_htr_edge_kernel, dim3(gn::xcd_grid(N)), dim3(256), 0, st, EQ, EK, rl, rowptr, src, N, F,
                       lmax, joint, rej, gate, w_raw, w);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

int gn_highl_htr_backward(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                          const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                          const int* dst, const int* colptr, const int* perm, int N, int F, int lmax, int mode,
                          float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act, hipStream_t st) {
    if (lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    const int joint = mode & GN_HTR_JOINT ? 1 : 0, rej = mode & GN_HTR_NOREJ ? 0 : 1, gate = (mode >> 2) & 3;
    const int direct = mode & GN_HTR_DIRECT ? 1 : 0;
    if (!direct && gate && !w_raw) return GN_ERR_BAD_ARG;
    const dim3 grid(gn::xcd_grid(N)), block(256);
    GN_HL_PER_DEGREE(hl_htr_bwd_target_kernel, g_t_out, pre_t, w, w_raw, EQ, EK, rl, rowptr, src, N, F, lmax, joint,
                     rej, gate, direct, gEQ, g_rl, g_pre_t, act);
    GN_HL_PER_DEGREE(hl_htr_bwd_source_kernel, g_t_out, pre_t, w_raw, EQ, rl, colptr, perm, dst, N, F, lmax, joint,
                     rej, gate, direct, gEK, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
