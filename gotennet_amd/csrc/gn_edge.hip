// gn_edge.hip -- graph plumbing, edge geometry (K1) and the two initialisation
// gathers (K2 NodeInit aggregate, K3 EdgeInit) of the GotenNet hot path.
//
// Lane layout shared by every per-target kernel in this library ("slot" layout):
// a 256-thread workgroup owns ONE target atom i; it is cut into NS = 256 / (F/4)
// slots of F/4 lanes; a slot walks the incoming edges e0+slot, e0+slot+NS, ... and
// each lane owns 4 consecutive feature channels, so every row access is one
// coalesced 16-byte-per-lane load (F = 256: one wave = one 1 KiB row).  Per-target
// sums are register accumulations followed by one fixed-order LDS reduction over the
// slots: no atomics, bit-reproducible.
#include "gn_common.h"
#include "gn_sh.h"

namespace gn {

// ---------------------------------------------------------------------------------- CSR
__global__ void build_csr_kernel(const int64_t* __restrict__ ei, int E, int N,
                                 int* __restrict__ src, int* __restrict__ dst, int* __restrict__ rowptr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (E == 0) {
        for (int n = e; n <= N; n += gridDim.x * blockDim.x) rowptr[n] = 0;
        return;
    }
    if (e >= E) return;
    const int s = (int)ei[e];
    const int d = (int)ei[(size_t)E + e];
    src[e] = s;
    dst[e] = d;
    // (targets outside [0, N) -- a caller error gn_check_edges reports -- must not make this kernel write out of bounds)
    const int dc = d < 0 ? -1 : (d >= N ? N - 1 : d);
    int prev = e > 0 ? (int)ei[(size_t)E + e - 1] : -1;
    prev = prev < -1 ? -1 : (prev >= N ? N - 1 : prev);
    for (int n = prev + 1; n <= dc; ++n) rowptr[n] = e;
    if (e == E - 1)
        for (int n = dc + 1; n <= N; ++n) rowptr[n] = E;
}

// CosineCutoff (layers.py:149-152) of a distance vector -- for callers that drive one GATA layer directly
__global__ void cosine_cutoff_kernel(const float* __restrict__ dist, int E, float cutoff, float* __restrict__ cut) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float d = dist[e];
    const float c = 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f);
    cut[e] = d < cutoff ? c : 0.0f;
}

// flag |= 1: edge_index[1] decreases somewhere (not target-major); |= 2: an index outside [0, N)
__global__ void check_edges_kernel(const int64_t* __restrict__ ei, int E, int N, int* __restrict__ flag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t s = ei[e], d = ei[(size_t)E + e];
    int bits = 0;
    if (s < 0 || s >= N || d < 0 || d >= N) bits |= 2;
    if (e > 0 && ei[(size_t)E + e - 1] > d) bits |= 1;
    if (bits) atomicOr(flag, bits);              // integer flag: order-independent
}

__global__ void out_degree_kernel(const int* __restrict__ src, int E, int* __restrict__ outdeg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&outdeg[src[e]], 1);   // integer count: order-independent
}

// ---------------------------------------------------------------------------------- K1
template <int LMAX>
__global__ void edge_sh_kernel(const float* __restrict__ vec, const float* __restrict__ dist,
                               const int* __restrict__ src, const int* __restrict__ dst, int E, float cutoff,
                               float* __restrict__ rl, float* __restrict__ cut) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float x = vec[3 * e], y = vec[3 * e + 1], z = vec[3 * e + 2];
    if (src[e] != dst[e]) {                       // gotennet.py:978-980
        const float n = sqrtf(x * x + y * y + z * z);
        x /= n; y /= n; z /= n;
    }
    float o[D];
    real_harmonics<LMAX, float>(x, y, z, o);
#pragma unroll
    for (int m = 0; m < D; ++m) rl[(size_t)e * D + m] = o[m];
    const float d = dist[e];                      // layers.py:149-152
    const float c = 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f);
    cut[e] = d < cutoff ? c : 0.0f;
}

// basis 0: ExpNormalSmearing (layers.py:703-746; p0 = means, p1 = betas, carries the cosine cutoff)
// basis 1: BesselBasis (layers.py:329-358; p0 = freqs):  sin(a d) / d, with d = 0 -> divide by 1
// basis 2: GaussianRBF (layers.py:276-326; p0 = offsets, p1 = widths):  exp(-0.5 (d - o)^2 / w^2)
__global__ void edge_rbf_kernel(const float* __restrict__ dist, int E, int R, int basis,
                                const float* __restrict__ p0, const float* __restrict__ p1,
                                float cutoff, float alpha, float* __restrict__ phi) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)E * R) return;
    const int e = (int)(idx / R), r = (int)(idx % R);
    const float d = dist[e];
    if (basis == 1) {
        phi[idx] = sinf(d * p0[r]) / (d == 0.0f ? 1.0f : d);
    } else if (basis == 2) {
        const float w = p1[r], q = d - p0[r];
        phi[idx] = expf((-0.5f / (w * w)) * (q * q));
    } else {                                          // layers.py:744-746
        const float c = d < cutoff ? 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f) : 0.0f;
        const float u = expf(alpha * (-d)) - p0[r];
        phi[idx] = c * expf(-p1[r] * (u * u));
    }
}

// ---------------------------------------------------------------------------------- K2
// ctx[i, 0:F] = A_na[z_i];  ctx[i, F:2F] = sum_{e: j->i, j != i} A_nbr[z_j] * (feat[e] * cut[e])
__global__ __launch_bounds__(256) void node_init_kernel(
    const int* __restrict__ z, const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ feat, int ldf, const float* __restrict__ cut,
    const float* __restrict__ A_na, const float* __restrict__ A_nbr, int N, int F, float* __restrict__ ctx) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 acc = zero4();
    for (int e = e0 + slot; e < e1; e += ns) {
        const int j = src[e];
        if (j == i) continue;                     // layers.py:1660-1664: self-loops dropped
        const float4 f = ld4(feat + (size_t)e * ldf + c0) * cut[e];
        acc = fma4(ld4(A_nbr + (size_t)z[j] * F + c0), f, acc);
    }
    st4(&red[slot * F + c0], acc);
    __syncthreads();
    if (slot == 0) {
        float4 s = red4(red, c0, F, ns);
        st4(ctx + (size_t)i * 2 * F + F + c0, s);
        st4(ctx + (size_t)i * 2 * F + c0, ld4(A_na + (size_t)z[i] * F + c0));
    }
}

// ---------------------------------------------------------------------------------- K3
// t[e] = (h[i] + h[j]) * feat[e]
__global__ __launch_bounds__(256) void edge_init_kernel(
    const float* __restrict__ h, const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ feat, int ldf, int N, int F, float* __restrict__ t) {
    const int i = xcd_item(blockIdx.x, N);
    if (i < 0) return;
    const int lps = F >> 2, ns = 256 / lps;
    const int slot = threadIdx.x / lps, c0 = (threadIdx.x % lps) * 4;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const float4 hi = ld4(h + (size_t)i * F + c0);
    for (int e = e0 + slot; e < e1; e += ns) {
        const float4 hj = ld4(h + (size_t)src[e] * F + c0);
        st4(t + (size_t)e * F + c0, (hi + hj) * ld4(feat + (size_t)e * ldf + c0));
    }
}

}  // namespace gn

// ====================================================================================== C ABI
static bool feature_dim_ok(int F) { return F >= 16 && F <= 1024 && gn::is_pow2(F); }

extern "C" int gn_abi_version(const char** arch_out) {
    if (arch_out) *arch_out = "gfx950";
    return GN_ABI_VERSION;
}

extern "C" int gn_build_csr(const int64_t* edge_index, int E, int N, int* src, int* dst, int* rowptr, void* stream) {
    if (E < 0 || N < 0) return GN_ERR_BAD_ARG;
    const int work = E > 0 ? E : N + 1;
    hipLaunchKernelGGL(gn::build_csr_kernel, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       edge_index, E, N, src, dst, rowptr);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_cosine_cutoff(const float* dist, int E, float cutoff, float* cut, void* stream) {
    if (E < 0) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipLaunchKernelGGL(gn::cosine_cutoff_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       dist, E, cutoff, cut);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_check_edges(const int64_t* edge_index, int E, int N, int* flag, void* stream) {
    if (E < 0 || N < 0 || !flag) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipLaunchKernelGGL(gn::check_edges_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       edge_index, E, N, flag);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_out_degree(const int* src, int E, int* outdeg, void* stream) {
    if (E < 0) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipLaunchKernelGGL(gn::out_degree_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, E, outdeg);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_geometry(const float* edge_vec, const float* edge_diff, const int* src, const int* dst, int E,
                                int lmax, int R, int basis, const float* means, const float* betas, float cutoff,
                                float* rl, float* phi, float* cut, void* stream) {
    if (E < 0 || lmax < 1 || lmax > 8 || R <= 0 || basis < 0 || basis > 2) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((E + 255) / 256), block(256);
    switch (lmax) {
        case 1: hipLaunchKernelGGL(gn::edge_sh_kernel<1>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 2: hipLaunchKernelGGL(gn::edge_sh_kernel<2>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 3: hipLaunchKernelGGL(gn::edge_sh_kernel<3>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 4: hipLaunchKernelGGL(gn::edge_sh_kernel<4>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 5: hipLaunchKernelGGL(gn::edge_sh_kernel<5>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 6: hipLaunchKernelGGL(gn::edge_sh_kernel<6>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        case 7: hipLaunchKernelGGL(gn::edge_sh_kernel<7>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
        default: hipLaunchKernelGGL(gn::edge_sh_kernel<8>, grid, block, 0, st, edge_vec, edge_diff, src, dst, E, cutoff, rl, cut); break;
    }
    GN_LAUNCH_CHECK();
    const size_t tot = (size_t)E * R;
    hipLaunchKernelGGL(gn::edge_rbf_kernel, dim3((unsigned)((tot + 255) / 256)), block, 0, st,
                       edge_diff, E, R, basis, means, betas, cutoff, 5.0f / cutoff, phi);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_node_init(const int* z, const int* rowptr, const int* src, const float* feat, int ldf,
                            const float* cut, const float* A_na, const float* A_nbr,
                            int N, int F, float* ctx, void* stream) {
    if (!feature_dim_ok(F) || N < 0 || (ldf & 3)) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::node_init_kernel, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                       z, rowptr, src, feat, ldf, cut, A_na, A_nbr, N, F, ctx);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_init(const float* h, const int* rowptr, const int* src, const float* feat, int ldf,
                            int N, int F, float* t, void* stream) {
    if (!feature_dim_ok(F) || N < 0 || (ldf & 3)) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::edge_init_kernel, dim3(gn::xcd_grid(N)), dim3(256), 0, (hipStream_t)stream,
                       h, rowptr, src, feat, ldf, N, F, t);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
