// gn_node.hip -- node-local pieces: LayerNorm+SiLU (NodeInit MLP) and EQFF (K8).
#include "gn_common.h"

namespace gn {

// one wave per row; torch.nn.LayerNorm semantics (biased variance, eps inside the sqrt)
template <bool SILU>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int N, int F, float* __restrict__ y, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (size_t)row * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += xr[f];
    const float mean = wave_sum(s) / (float)F;
    float q = 0.f;
    for (int f = lane; f < F; f += 64) { const float d = xr[f] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)F + eps);
    for (int f = lane; f < F; f += 64) {
        const float v = (xr[f] - mean) * rstd * gamma[f] + beta[f];
        y[(size_t)row * F + f] = SILU ? act1(v, act) : v;
    }
}

// gotennet.py:731-735
__global__ void eqff_context_kernel(const float* __restrict__ h, const float* __restrict__ Xp, float eps,
                                    int N, int F, int D, float* __restrict__ ctx) {
    const int f4 = F >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * f4) return;
    const int n = (int)(idx / f4), c0 = (int)(idx % f4) * 4;
    float4 s = zero4();
    for (int m = 0; m < D; ++m) {
        const float4 p = ld4(Xp + ((size_t)n * D + m) * F + c0);
        s = fma4(p, p, s);
    }
    st4(ctx + (size_t)n * 2 * F + c0, ld4(h + (size_t)n * F + c0));
    st4(ctx + (size_t)n * 2 * F + F + c0,
        make_float4(sqrtf(s.x + eps), sqrtf(s.y + eps), sqrtf(s.z + eps), sqrtf(s.w + eps)));
}

// gotennet.py:741-746
__global__ void eqff_update_kernel(const float* __restrict__ mm, const float* __restrict__ Xp,
                                   int N, int F, int D, float* __restrict__ h, float* __restrict__ X) {
    const int f4 = F >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * f4) return;
    const int n = (int)(idx / f4), c0 = (int)(idx % f4) * 4;
    const float4 m1 = ld4(mm + (size_t)n * 2 * F + c0);
    const float4 m2 = ld4(mm + (size_t)n * 2 * F + F + c0);
    float* hp = h + (size_t)n * F + c0;
    st4(hp, ld4(hp) + m1);
    for (int m = 0; m < D; ++m) {
        const size_t off = ((size_t)n * D + m) * F + c0;
        st4(X + off, fma4(m2, ld4(Xp + off), ld4(X + off)));
    }
}

}  // namespace gn

extern "C" int gn_layernorm_silu(const float* x, const float* gamma, const float* beta, float eps,
                                 int N, int F, float* y, int act, void* stream) {
    if (N < 0 || F <= 0 || act < 0 || act >= GN_ACT_COUNT) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::layernorm_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, gamma, beta, eps, N, F, y, act);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_layernorm(const float* x, const float* gamma, const float* beta, float eps,
                            int N, int F, float* y, void* stream) {
    if (N < 0 || F <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::layernorm_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, gamma, beta, eps, N, F, y, 0);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_eqff_context(const float* h, const float* Xp, float eps, int N, int F, int D, float* ctx, void* stream) {
    if (N < 0 || F <= 0 || (F & 3) || D <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * (F / 4);
    hipLaunchKernelGGL(gn::eqff_context_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       h, Xp, eps, N, F, D, ctx);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_eqff_update(const float* m, const float* Xp, int N, int F, int D, float* h, float* X, void* stream) {
    if (N < 0 || F <= 0 || (F & 3) || D <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * (F / 4);
    hipLaunchKernelGGL(gn::eqff_update_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       m, Xp, N, F, D, h, X);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
