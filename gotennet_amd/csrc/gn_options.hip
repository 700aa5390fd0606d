// gn_options.hip -- kernels behind the reference's NON-default constructor flags:
//   * HTR edge weights with joint (sep_htr=False) blocks, rejection off ("norej") and the element-wise
//     gamma_w gate ("gated" sigmoid / "gatedt" tanh / "act" SiLU) -- gotennet.py:139-190, 285-291, 561-611;
//   * TensorLayerNorm (steerable_norm != "") -- layers.py:1497-1563.
// Same slot layout and fixed-order reductions as the default kernels (gn_gata.hip / gn_backward.hip); these
// variants trade a second pass over the [D, F] rows (L1/L2 hits) for lower register pressure, the default
// configuration never reaches them.
#include "gn_common.h"
#include "gn_highl.h"

namespace gn {



// row m (0-based, l = 0 omitted) closes degree l when m + 2 = (l + 1)^2
__host__ __device__ constexpr bool closes_degree(int m) {
    for (int l = 1; l <= 8; ++l)
        if (m + 2 == (l + 1) * (l + 1)) return true;
    return false;
}
__host__ __device__ constexpr int degree_of(int m) {
    int l = 1;
    while ((l + 1) * (l + 1) - 1 <= m) ++l;
    return l;
}

// ------------------------------------------------------------------ forward
// per block B (a degree, or all D rows when `joint`):  w += A.B - (2 - r.r)(A.r)(B.r)   [rejection on]
//                                                       w += A.B                        [rejection off]
// (closed form of the two vector rejections at gotennet.py:351-364, 586-599: P(a) = a - (a.r) r,
//  P'(b) = b - (b.(-r))(-r);  P(a).P'(b) = a.b - 2 (a.r)(b.r) + (a.r)(b.r)(r.r).)
template <int LMAX>
__global__ __launch_bounds__(256) void htr_edge_general_kernel(
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F, int joint, int rej, int gate,
    float* __restrict__ w_raw, float* __restrict__ w) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 eq[D];
#pragma unroll
    for (int m = 0; m < D; ++m) eq[m] = ld4(EQ + ((size_t)i * D + m) * F + c0);
    for (int e = e0 + slot; e < e1; e += ns) {
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 wsum = zero4(), ab = zero4(), pa = zero4(), pb = zero4();
        float rr = 0.f;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const float4 ek = ld4(kj + (size_t)m * F);
            const float r = re[m];
            ab = fma4(eq[m], ek, ab);
            pa = fma4(r, eq[m], pa);
            pb = fma4(r, ek, pb);
            rr = fmaf(r, r, rr);
            if (m == D - 1 || (closes_degree(m) && !joint)) {
                wsum = wsum + (rej ? ab - (pa * pb) * (2.0f - rr) : ab);
                ab = pa = pb = zero4();
                rr = 0.f;
            }
        }
        if (w_raw) st4(w_raw + (size_t)e * F + c0, wsum);
        st4(w + (size_t)e * F + c0, gate4(wsum, gate));
    }
}

// ------------------------------------------------------------------ backward, by target
// t' = t + SiLU(pre_t) * g(w):  g_pre_t = g_t' g(w) SiLU'(pre_t);  g_w = g_t' SiLU(pre_t) g'(w)
// dw/dA_m = B_m - c r_m (B.r);  dw/dr_m = -c [A_m (B.r) + (A.r) B_m] + 2 r_m (A.r)(B.r),  c = 2 - r.r
template <int LMAX>
__global__ __launch_bounds__(256) void htr_bwd_target_general_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w,
    const float* __restrict__ w_raw, const float* __restrict__ EQ, const float* __restrict__ EK,
    const float* __restrict__ rl, const int* __restrict__ rowptr, const int* __restrict__ src, int N, int F,
    int joint, int rej, int gate, int direct, float* __restrict__ gEQ, float* __restrict__ g_rl,
    float* __restrict__ g_pre_t, int act) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, lp = threadIdx.x % lps, c0 = lp * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 eq[D], acc[D];
#pragma unroll
    for (int m = 0; m < D; ++m) { eq[m] = ld4(EQ + ((size_t)i * D + m) * F + c0); acc[m] = zero4(); }
    for (int e = e0 + slot; e < e1; e += ns) {
        float4 gw = ld4(gtp + (size_t)e * F + c0);       // direct: this already is dL/dw
        if (!direct) {
            const float4 pte = ld4(pre_t + (size_t)e * F + c0);
            st4(g_pre_t + (size_t)e * F + c0, gw * ld4(w + (size_t)e * F + c0) * dact4(pte, act));
            gw = gw * act4(pte, act);
            if (gate) gw = gw * dgate4(ld4(w_raw + (size_t)e * F + c0), gate);
        }
        const float* kj = EK + (size_t)src[e] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 pa[LMAX], pb[LMAX];
        float rr[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { pa[l] = pb[l] = zero4(); rr[l] = 0.f; }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            constexpr int dummy = 0; (void)dummy;
            const int l = degree_of(m) - 1;
            const float4 ek = ld4(kj + (size_t)m * F);
            const float r = re[m];
            pa[l] = fma4(r, eq[m], pa[l]);
            pb[l] = fma4(r, ek, pb[l]);
            rr[l] = fmaf(r, r, rr[l]);
        }
        if (joint) {
            float4 ta = pa[0], tb = pb[0];
            float tr = rr[0];
#pragma unroll
            for (int l = 1; l < LMAX; ++l) { ta = ta + pa[l]; tb = tb + pb[l]; tr += rr[l]; }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { pa[l] = ta; pb[l] = tb; rr[l] = tr; }
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int l = degree_of(m) - 1;
            const float4 ek = ld4(kj + (size_t)m * F);
            const float r = re[m];
            const float c = rej ? 2.0f - rr[l] : 0.0f;
            acc[m] = fma4(gw, ek + pb[l] * (-c * r), acc[m]);
            const float4 t4 = gw * ((eq[m] * pb[l] + pa[l] * ek) * (-c) + (pa[l] * pb[l]) * (rej ? 2.0f * r : 0.0f));
            const float s = group_sum(hsum4(t4), lps);
            if (lp == 0) g_rl[(size_t)e * D + m] = s;
        }
    }
    // fixed-order cross-slot reduction, one row at a time
#pragma unroll
    for (int m = 0; m < D; ++m) {
        if (m) __syncthreads();
        st4(&red[slot * F + c0], acc[m]);
        __syncthreads();
        if (slot == 0) st4(gEQ + ((size_t)i * D + m) * F + c0, red4(red, c0, F, ns));
    }
}

// ------------------------------------------------------------------ backward, by source:  dw/dB_m = A_m - c r_m (A.r)
template <int LMAX>
__global__ __launch_bounds__(256) void htr_bwd_source_general_kernel(
    const float* __restrict__ gtp, const float* __restrict__ pre_t, const float* __restrict__ w_raw,
    const float* __restrict__ EQ, const float* __restrict__ rl,
    const int* __restrict__ colptr, const int* __restrict__ perm, const int* __restrict__ dst, int N, int F,
    int joint, int rej, int gate, int direct, float* __restrict__ gEK, int act) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int j = xcd_item(blockIdx.x, N);
    if (j < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int p0 = colptr[j], p1 = colptr[j + 1];
    float4 acc[D];
#pragma unroll
    for (int m = 0; m < D; ++m) acc[m] = zero4();
    for (int pp = p0 + slot; pp < p1; pp += ns) {
        const int e = perm[pp];
        float4 gw = ld4(gtp + (size_t)e * F + c0);
        if (!direct) {
            gw = gw * act4(ld4(pre_t + (size_t)e * F + c0), act);
            if (gate) gw = gw * dgate4(ld4(w_raw + (size_t)e * F + c0), gate);
        }
        const float* qi = EQ + (size_t)dst[pp] * D * F + c0;
        const float* re = rl + (size_t)e * D;
        float4 pa[LMAX];
        float rr[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { pa[l] = zero4(); rr[l] = 0.f; }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int l = degree_of(m) - 1;
            const float r = re[m];
            pa[l] = fma4(r, ld4(qi + (size_t)m * F), pa[l]);
            rr[l] = fmaf(r, r, rr[l]);
        }
        if (joint) {
            float4 ta = pa[0];
            float tr = rr[0];
#pragma unroll
            for (int l = 1; l < LMAX; ++l) { ta = ta + pa[l]; tr += rr[l]; }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { pa[l] = ta; rr[l] = tr; }
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int l = degree_of(m) - 1;
            const float c = rej ? 2.0f - rr[l] : 0.0f;
            acc[m] = fma4(gw, ld4(qi + (size_t)m * F) + pa[l] * (-c * re[m]), acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < D; ++m) {
        if (m) __syncthreads();
        st4(&red[slot * F + c0], acc[m]);
        __syncthreads();
        if (slot == 0) st4(gEK + ((size_t)j * D + m) * F + c0, red4(red, c0, F, ns));
    }
}

// ------------------------------------------------------------------ TensorLayerNorm (layers.py:1529-1563)
// one wave per (atom, degree): s_f = |X_l[:, f]|, c_f = max(s_f, eps), n_f = (c_f - min_f c) / (max_f c - min_f c)
// (denominator 1 when max = min), out = relu(n_f) * X / c_f * weight_f.
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

__global__ __launch_bounds__(256) void tensor_norm_kernel(
    const float* __restrict__ X, const float* __restrict__ weight, float eps, int N, int F, int LMAX,
    float* __restrict__ Y) {
    const int D = (LMAX + 1) * (LMAX + 1) - 1;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= N * LMAX) return;
    const int n = item / LMAX, l = item % LMAX + 1;
    const int off = l * l - 1, cnt = 2 * l + 1;
    const float* x = X + ((size_t)n * D + off) * F;
    float* y = Y + ((size_t)n * D + off) * F;
    float mx = -INFINITY, mn = INFINITY;
    for (int f = lane; f < F; f += 64) {
        float q = 0.f;
        for (int m = 0; m < cnt; ++m) { const float t = x[(size_t)m * F + f]; q = fmaf(t, t, q); }
        const float c = fmaxf(sqrtf(q), eps);
        mx = fmaxf(mx, c); mn = fminf(mn, c);
    }
    mx = wave_max(mx); mn = wave_min(mn);
    float delta = mx - mn;
    if (delta == 0.f) delta = 1.f;
    for (int f = lane; f < F; f += 64) {
        float q = 0.f;
        for (int m = 0; m < cnt; ++m) { const float t = x[(size_t)m * F + f]; q = fmaf(t, t, q); }
        const float c = fmaxf(sqrtf(q), eps);
        const float nf = fmaxf((c - mn) / delta, 0.f);
        const float wf = weight[f];
        for (int m = 0; m < cnt; ++m) y[(size_t)m * F + f] = (nf * (x[(size_t)m * F + f] / c)) * wf;
    }
}

// backward.  G_f = sum_m g_out[m,f] w_f x[m,f];  dn_f = [n_f > 0] G_f / c_f;
// d c_f = -relu(n_f) G_f / c_f^2 + dn_f / delta + [f = argmax] d_mx + [f = argmin] d_mn,
// d_delta = -(sum_f dn_f (c_f - mn)) / delta^2 (0 when max = min), d_mx = d_delta, d_mn = -(sum_f dn_f) / delta - d_delta;
// g_x[m,f] = g_out[m,f] w_f relu(n_f) / c_f + d c_f * [s_f >= eps] x[m,f] / s_f      (torch.max/min route the
// gradient to ONE index: the first extremal channel, as the CPU oracle does.)
__global__ __launch_bounds__(256) void tensor_norm_bwd_kernel(
    const float* __restrict__ X, const float* __restrict__ weight, const float* __restrict__ gY, float eps,
    int N, int F, int LMAX, float* __restrict__ gX) {
    const int D = (LMAX + 1) * (LMAX + 1) - 1;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= N * LMAX) return;
    const int n = item / LMAX, l = item % LMAX + 1;
    const int off = l * l - 1, cnt = 2 * l + 1;
    const float* x = X + ((size_t)n * D + off) * F;
    const float* gy = gY + ((size_t)n * D + off) * F;
    float* gx = gX + ((size_t)n * D + off) * F;
    float mx = -INFINITY, mn = INFINITY;
    int amx = 0x7fffffff, amn = 0x7fffffff;
    for (int f = lane; f < F; f += 64) {
        float q = 0.f;
        for (int m = 0; m < cnt; ++m) { const float t = x[(size_t)m * F + f]; q = fmaf(t, t, q); }
        const float c = fmaxf(sqrtf(q), eps);
        if (c > mx) { mx = c; amx = f; }
        if (c < mn) { mn = c; amn = f; }
    }
    for (int o = 1; o < GN_WAVE; o <<= 1) {          // (value, first index) butterflies
        const float ov = __shfl_xor(mx, o, GN_WAVE); const int oi = __shfl_xor(amx, o, GN_WAVE);
        if (ov > mx || (ov == mx && oi < amx)) { mx = ov; amx = oi; }
        const float uv = __shfl_xor(mn, o, GN_WAVE); const int ui = __shfl_xor(amn, o, GN_WAVE);
        if (uv < mn || (uv == mn && ui < amn)) { mn = uv; amn = ui; }
    }
    const bool flat = (mx - mn) == 0.f;
    const float delta = flat ? 1.f : mx - mn;
    float s1 = 0.f, s2 = 0.f;
    for (int f = lane; f < F; f += 64) {
        float q = 0.f, G = 0.f;
        const float wf = weight[f];
        for (int m = 0; m < cnt; ++m) {
            const float t = x[(size_t)m * F + f];
            q = fmaf(t, t, q);
            G = fmaf(gy[(size_t)m * F + f] * wf, t, G);
        }
        const float c = fmaxf(sqrtf(q), eps);
        const float nf = (c - mn) / delta;
        const float dn = nf > 0.f ? G / c : 0.f;
        s1 += dn; s2 += dn * (c - mn);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float d_delta = flat ? 0.f : -s2 / (delta * delta);
    const float d_mx = d_delta, d_mn = -s1 / delta - d_delta;
    for (int f = lane; f < F; f += 64) {
        float q = 0.f, G = 0.f;
        const float wf = weight[f];
        for (int m = 0; m < cnt; ++m) {
            const float t = x[(size_t)m * F + f];
            q = fmaf(t, t, q);
            G = fmaf(gy[(size_t)m * F + f] * wf, t, G);
        }
        const float s = sqrtf(q);
        const float c = fmaxf(s, eps);
        const float nf = (c - mn) / delta;
        const float rn = fmaxf(nf, 0.f);
        float dc = -rn * G / (c * c) + (nf > 0.f ? G / c : 0.f) / delta;
        if (f == amx) dc += d_mx;
        if (f == amn) dc += d_mn;
        const float k = (s >= eps && s > 0.f) ? dc / s : 0.f;
        for (int m = 0; m < cnt; ++m)
            gx[(size_t)m * F + f] = gy[(size_t)m * F + f] * wf * rn / c + k * x[(size_t)m * F + f];
    }
}

// ------------------------------------------------------------------ element-wise pieces of the composed edge update
// (gamma_t = 2-layer MLP "mlp"/"mlpa", gamma_w with W_edp "linw"/"linwa" and LayerNorms "ln"/"postln")
__global__ void gate_kernel(const float* __restrict__ x, int kind, size_t n4, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st4(y + 4 * i, gate4(ld4(x + 4 * i), kind));
}
__global__ void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, int kind, size_t n4,
                                float* __restrict__ gx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st4(gx + 4 * i, ld4(g + 4 * i) * dgate4(ld4(x + 4 * i), kind));
}
// t' = t + act(pre) * wg:  g_pre = g act'(pre) wg,  g_wg = g act(pre)      (act: -1 identity, else GN_ACT_*)
__global__ void edge_gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pre, int act,
                                     const float* __restrict__ wg, size_t n4, float* __restrict__ g_pre,
                                     float* __restrict__ g_wg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 gv = ld4(g + 4 * i), p = ld4(pre + 4 * i);
    st4(g_pre + 4 * i, (act < 0 ? gv : gv * dact4(p, act)) * ld4(wg + 4 * i));
    st4(g_wg + 4 * i, gv * (act < 0 ? p : act4(p, act)));
}

}  // namespace gn

// ====================================================================================== C ABI
#define GN_OPT_SWITCH(KERNEL, grid, block, st, ...)                                                         \
    switch (lmax) {                                                                                          \
        case 1: hipLaunchKernelGGL(gn::KERNEL<1>, grid, block, 0, st, __VA_ARGS__); break;                   \
        case 2: hipLaunchKernelGGL(gn::KERNEL<2>, grid, block, 0, st, __VA_ARGS__); break;                   \
        case 3: hipLaunchKernelGGL(gn::KERNEL<3>, grid, block, 0, st, __VA_ARGS__); break;                   \
        default: hipLaunchKernelGGL(gn::KERNEL<4>, grid, block, 0, st, __VA_ARGS__); break;                  \
    }

// called by gn_htr_edge / gn_htr_backward (gn_gata.hip, gn_backward.hip) when mode != 0
int gn_htr_edge_general(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                        int N, int F, int lmax, int mode, float* w_raw, float* w, hipStream_t st) {
    const int joint = mode & GN_HTR_JOINT ? 1 : 0, rej = mode & GN_HTR_NOREJ ? 0 : 1, gate = (mode >> 2) & 3;
    const dim3 grid(gn::xcd_grid(N)), block(256);
    GN_OPT_SWITCH(htr_edge_general_kernel, grid, block, st, EQ, EK, rl, rowptr, src, N, F, joint, rej, gate, w_raw, w);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

int gn_htr_backward_general(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                            const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                            const int* dst, const int* colptr, const int* perm, int N, int F, int lmax, int mode,
                            float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act, hipStream_t st) {
    const int joint = mode & GN_HTR_JOINT ? 1 : 0, rej = mode & GN_HTR_NOREJ ? 0 : 1, gate = (mode >> 2) & 3;
    const int direct = mode & GN_HTR_DIRECT ? 1 : 0;
    if (!direct && gate && !w_raw) return GN_ERR_BAD_ARG;
    const dim3 grid(gn::xcd_grid(N)), block(256);
    GN_OPT_SWITCH(htr_bwd_target_general_kernel, grid, block, st, g_t_out, pre_t, w, w_raw, EQ, EK, rl, rowptr, src,
                  N, F, joint, rej, gate, direct, gEQ, g_rl, g_pre_t, act);
    GN_LAUNCH_CHECK();
    GN_OPT_SWITCH(htr_bwd_source_general_kernel, grid, block, st, g_t_out, pre_t, w_raw, EQ, rl, colptr, perm, dst,
                  N, F, joint, rej, gate, direct, gEK, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_tensor_norm(const float* X, const float* weight, float eps, int N, int F, int lmax, float* Y,
                              void* stream) {
    if (N < 0 || F <= 0 || lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N * lmax + 3) / 4), block(256);
    hipLaunchKernelGGL(gn::tensor_norm_kernel, grid, block, 0, st, X, weight, eps, N, F, lmax, Y);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_tensor_norm_backward(const float* X, const float* weight, const float* g_Y, float eps, int N, int F,
                                       int lmax, float* g_X, void* stream) {
    if (N < 0 || F <= 0 || lmax < 1 || lmax > 8) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N * lmax + 3) / 4), block(256);
    hipLaunchKernelGGL(gn::tensor_norm_bwd_kernel, grid, block, 0, st, X, weight, g_Y, eps, N, F, lmax, g_X);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

static inline unsigned blocks_for(size_t n4) { return (unsigned)((n4 + 255) / 256); }

extern "C" int gn_gate(const float* x, int kind, long n, float* y, void* stream) {
    if (n < 0 || (n & 3) || kind < 0 || kind > 3) return GN_ERR_BAD_ARG;
    if (n == 0) return GN_OK;
    hipLaunchKernelGGL(gn::gate_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, kind, (size_t)n / 4, y);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_gate_backward(const float* g, const float* x, int kind, long n, float* g_x, void* stream) {
    if (n < 0 || (n & 3) || kind < 0 || kind > 3) return GN_ERR_BAD_ARG;
    if (n == 0) return GN_OK;
    hipLaunchKernelGGL(gn::gate_bwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, g, x, kind,
                       (size_t)n / 4, g_x);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_gate_backward(const float* g, const float* pre, int act, const float* wg, long n,
                                     float* g_pre, float* g_wg, void* stream) {
    if (n < 0 || (n & 3) || act < -1 || act >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (n == 0) return GN_OK;
    hipLaunchKernelGGL(gn::edge_gate_bwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, g, pre, act,
                       wg, (size_t)n / 4, g_pre, g_wg);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
