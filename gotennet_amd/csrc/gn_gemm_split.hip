// gn_gemm_split.hip -- fp32-accurate projections on the bf16 matrix cores (3 x bf16 split): weight preparation.
//
// the exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR
// rate, 1/16 of the bf16 MFMA rate, and there is no TF32/xf32 on gfx950.  In split mode every fp32
// operand is cut into three bf16 planes,  x = hi + mid + lo  (8 + 8 + 8 significand bits, by
// truncation: the split is EXACT), and the product is accumulated in fp32 from the six plane pairs
// with i + j <= 2:
//     a b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_mid b_mid + a_lo b_hi
// The dropped pairs are below 2^-24 |a b|; bf16 x bf16 products are exact in fp32 and the MFMA
// accumulates in fp32, so the result carries fp32-class error (measured <= 1e-6 relative to an
// fp64 product, the same as the exact-fp32 kernel) at 6 bf16 MFMAs per fp32 MFMA-equivalent:
// 16 / 6 = 2.67x the fp32 MFMA rate.
//
// The kernel is the SPLIT instantiation of gn::gemm_f32_mfma (gn_gemm.hip): same grouping, persistent tile walk,
// prologues and epilogue as the exact-fp32 one.  A is split while it is staged into LDS; the weights are static, so
// they are split ONCE here and stored in the order the MFMA consumes them:
//
//   W3f[nt][g][plane][lane][e]  (bf16),  nt = column block of 32, g = k-step of 16 (padded to an even count),
//   element = W_plane[n = 32 nt + (lane & 31)][k = 16 g + 8 (lane >> 5) + e],  zero outside [N) x [K)
//
// i.e. one wave-wide 16-byte load (1 KiB, contiguous) is exactly one B operand of v_mfma_f32_32x32x16_bf16.
#include "gn_gemm.h"

namespace gn {

__global__ void split_bf16x3_frag_kernel(const float* __restrict__ w, int N, int K, int ks2, long total,
                                         __bf16* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (nt, g, lane, e) per thread, 3 planes
    if (idx >= total) return;
    const int e = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const long rest = idx >> 9;
    const int g = (int)(rest % ks2);
    const long nt = rest / ks2;
    const long n = nt * 32 + (lane & 31);
    const int k = 16 * g + 8 * (lane >> 5) + e;
    const float x = (n < N && k < K) ? w[n * K + k] : 0.f;
    const float hi = trunc_bf16(x);
    const float r1 = x - hi;
    const float mid = trunc_bf16(r1);
    const float lo = r1 - mid;
    unsigned short* o = reinterpret_cast<unsigned short*>(out) + ((nt * ks2 + g) * 3) * 512 + lane * 8 + e;
    o[0] = (unsigned short)(__float_as_uint(hi) >> 16);
    o[512] = (unsigned short)(__float_as_uint(mid) >> 16);
    o[1024] = (unsigned short)(__float_as_uint(lo) >> 16);
}

// ---- 2 x fp16 planes with a tensor exponent (MODE 2 of gemm_body) ---------------------------------------------------
// out = [256-byte header: int exponent ewt, uint amax bits][nt][g][plane 0..1][lane][8 fp16]; planes hold
// w' = w * 2^-ewt (|w'| < 2^15) as hi = fp16(w'), lo = fp16(w' - hi).  Two launches, no host read-back.
// (one atomic per wave of 4096 elements, not per 64: a [1536 x 256] weight took 118 us with 6144 contended atomics)
__global__ void weight_amax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ hdr) {
    const long base = (long)blockIdx.x * (blockDim.x * 64) + threadIdx.x;
    float m = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const long idx = base + (long)k * blockDim.x;
        m = fmaxf(m, idx < n ? fabsf(w[idx]) : 0.f);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(hdr + 1, __float_as_uint(m));      // |x| as uint: order-preserving, exact
}
__global__ void split_f16x2_frag_kernel(const float* __restrict__ w, int N, int K, int ks2, long total,
                                        unsigned* __restrict__ hdr) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ewt = (int)((hdr[1] >> 23) & 0xffu) - 126 - 15;                  // |w| < 2^(ewt + 15)
    if (idx == 0) reinterpret_cast<int*>(hdr)[0] = ewt;
    if (idx >= total) return;
    const int e = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const long rest = idx >> 9;
    const int g = (int)(rest % ks2);
    const long nt = rest / ks2;
    const long n = nt * 32 + (lane & 31);
    const int k = 16 * g + 8 * (lane >> 5) + e;
    const float x = ldexpf((n < N && k < K) ? w[n * K + k] : 0.f, -ewt);
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    _Float16* o = reinterpret_cast<_Float16*>(hdr) + 128 + ((nt * ks2 + g) * 2) * 512 + lane * 8 + e;
    o[0] = hi;
    o[512] = lo;
}

}  // namespace gn

extern "C" long gn_split_f16x2_size(int N, int K) {            // in 16-bit elements, header included
    if (N <= 0 || K <= 0) return 0;
    return 128 + (long)((N + 31) / 32) * (2L * ((K + gn::BK - 1) / gn::BK)) * 2 * 512;
}

extern "C" int gn_split_f16x2(const float* w, int N, int K, unsigned short* out, void* stream) {
    if (N <= 0 || K <= 0 || !w || !out) return GN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ks2 = 2 * ((K + gn::BK - 1) / gn::BK);
    const long total = (long)((N + 31) / 32) * ks2 * 512, n = (long)N * K;
    if (hipMemsetAsync(out, 0, 256, st) != hipSuccess) return GN_ERR_BAD_ARG;
    hipLaunchKernelGGL(gn::weight_amax_kernel, dim3((unsigned)((n + 256 * 64 - 1) / (256 * 64))), dim3(256), 0, st, w, n,
                       reinterpret_cast<unsigned*>(out));
    hipLaunchKernelGGL(gn::split_f16x2_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, N, K, ks2,
                       total, reinterpret_cast<unsigned*>(out));
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" long gn_split_bf16x3_size(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return (long)((N + 31) / 32) * (2L * ((K + gn::BK - 1) / gn::BK)) * 3 * 512;
}

extern "C" int gn_split_bf16x3(const float* w, int N, int K, unsigned short* out, void* stream) {
    if (N <= 0 || K <= 0 || !w || !out) return GN_ERR_BAD_ARG;
    const int ks2 = 2 * ((K + gn::BK - 1) / gn::BK);
    const long total = (long)((N + 31) / 32) * ks2 * 512;
    hipLaunchKernelGGL(gn::split_bf16x3_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, N, K, ks2, total, reinterpret_cast<__bf16*>(out));
    GN_LAUNCH_CHECK();
    return GN_OK;
}
