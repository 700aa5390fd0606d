// gn_gemm_split.hip -- fp32-accurate projections on the bf16 matrix cores (3 x bf16 split).
//
// the exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR
// rate, 1/16 of the bf16 MFMA rate, and there is no TF32/xf32 on gfx950.  Here every fp32
// operand is split into three bf16 planes,  x = hi + mid + lo  (8 + 8 + 8 significand bits:
// the split is exact to 2^-25 |x|), and the product is accumulated in fp32 from the six
// plane pairs with i + j <= 2:
//     a b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_mid b_mid + a_lo b_hi
// The dropped pairs are below 2^-25 |a b|; bf16 x bf16 products are exact in fp32 and the MFMA
// accumulates in fp32, so the result carries fp32-class error (measured <= 1e-6 relative to an
// fp64 product, the same as the exact-fp32 kernel) at 6 bf16 MFMAs per fp32 MFMA-equivalent:
// 16 / 6 = 2.67x the fp32 MFMA rate.
//
// Same interface, tiling (128x128 / 64x64 workgroup tiles, 2x2 waves, K slabs of 32, XCD-aware
// tile order) and fused prologue / epilogue as gn_gemm.hip.  A is split while it is staged into
// LDS (v_cvt_pk_bf16_f32, after the SiLU / SiLU' prologue); the weights arrive pre-split
// ([3][N][K] bf16 planes, made once per weight by gn_split_bf16x3).  LDS planes are [rows][40] bf16
// (80-byte pitch: the 16-lane ds_read_b128 groups hit 16 distinct 4-bank slots).
#include "gn_gemm.h"

namespace gn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int PB = 40;                               // bf16 elements per LDS row (32 + 8 pad)

__global__ void split_bf16x3_kernel(const float* __restrict__ w, size_t n, __bf16* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    out[i] = hi; out[n + i] = mid; out[2 * n + i] = (__bf16)r2;
}

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
    }
}

template <int TM, int TN, bool PRO>
__global__ __launch_bounds__(256) void gemm_bf16x3_mfma(const GemmArgs p, const __bf16* __restrict__ W3) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM / 32;                     // A: float4 rows per thread per slab
    constexpr int RBW = BN / 64;                    // W: 16-byte (8 x bf16) rows per thread per plane per slab
    constexpr int APLANE = BM * PB, BPLANE = BN * PB;           // bf16 elements per plane
    constexpr int STAGE_BYTES = 3 * (APLANE + BPLANE) * 2;
    constexpr int CP = BN + 4;
    constexpr int LDS_BYTES = (STAGE_BYTES > BM * CP * 4) ? STAGE_BYTES : BM * CP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[LDS_BYTES];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);           // [3][BM][PB]
    __bf16* Bs = As + 3 * APLANE;                               // [3][BN][PB]
    float* Cs = reinterpret_cast<float*>(smem_raw);

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int xq = tiles_m >> 3, xr = tiles_m & 7, xcd = blockIdx.x & 7;
    const int idx = blockIdx.x >> 3;
    const int rows_here = xq + (xcd < xr ? 1 : 0);
    if (idx / tiles_n >= rows_here) return;
    const int tm = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx / tiles_n;
    const int m0 = tm * BM;
    const int n0 = (idx % tiles_n) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 7, sr = tid >> 3;          // A staging: (row sr + 32 i, float4 column c4)
    const int c8 = tid & 3, wr = tid >> 2;          // W staging: (row wr + 64 i, 8-element column c8)
    const size_t plane = (size_t)p.N * p.K;

    int prow[RA];
    bool aok[RA], bok[RBW];
    const __bf16* wrow[RBW];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int gm = m0 + sr + 32 * i;
        aok[i] = gm < p.M;
        prow[i] = phys_row(p, aok[i] ? gm : 0);
    }
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int gn = n0 + wr + 64 * i;
        bok[i] = gn < p.N;
        wrow[i] = W3 + (size_t)(bok[i] ? gn : 0) * p.K + 8 * c8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 pa[RA];
    uint4 pw[3][RBW];
    auto fetch = [&](int k0) {
        const int kc = k0 + 4 * c4;
        const bool kok = kc < p.K;
        const bool pro = PRO && p.pro_mode && kc >= p.pro_lo && kc < p.pro_hi;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float4 v = zero4();
            if (aok[i] && kok) {
                v = ld4(p.A + (size_t)prow[i] * p.lda + kc);
                if constexpr (PRO) {
                    if (pro) v = (p.pro_mode == 1) ? silu4(v) : v * dsilu4(ld4(p.a_pre + (size_t)prow[i] * p.ldp + kc));
                    if (p.a_gate) v = v * ld4(p.a_gate + (size_t)prow[i] * p.ldg + kc);
                }
            }
            pa[i] = v;
        }
        const bool wok = (k0 + 8 * c8) < p.K;       // K is a multiple of 8 on this path
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < RBW; ++i)
                pw[s][i] = (bok[i] && wok) ? *reinterpret_cast<const uint4*>(wrow[i] + s * plane + k0) : make_uint4(0, 0, 0, 0);
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bf16x4 h, m, l;
            split4(pa[i], h, m, l);
            __bf16* d = As + (sr + 32 * i) * PB + 4 * c4;
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + APLANE) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * APLANE) = l;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < RBW; ++i)
                *reinterpret_cast<uint4*>(Bs + s * BPLANE + (wr + 64 * i) * PB + 8 * c8) = pw[s][i];
    };

    const int nk = (p.K + BK - 1) / BK;
    fetch(0);
    stash();
    __syncthreads();

    const int frow = lane & 31;
    const int kq = (lane >> 5) * 8;                 // lanes 0-31: k 0..7, lanes 32-63: k 8..15 of a 16-deep step
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fetch((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {            // two 16-deep MFMA steps per slab
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    a[i][s] = *reinterpret_cast<const bf16x8*>(As + s * APLANE + (wm * 32 * TM + i * 32 + frow) * PB + ks * 16 + kq);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    b[j][s] = *reinterpret_cast<const bf16x8*>(Bs + s * BPLANE + (wn * 32 * TN + j * 32 + frow) * PB + ks * 16 + kq);
            // smallest terms first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi); consecutive MFMAs
            // rotate over the TM*TN accumulators so no instruction waits on the one before it
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            stash();
            __syncthreads();
        }
    }

    // epilogue through LDS (identical to gn_gemm.hip)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * CP + wn * 32 * TN + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    constexpr int C4 = BN / 4;
#pragma unroll 4
    for (int it = 0; it < (BM * C4) / 256; ++it) {
        const int e = it * 256 + tid;
        const int row = e / C4, cc = (e % C4) * 4;
        const int gm = m0 + row, gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float4 v = ld4(&Cs[row * CP + cc]);
        if (p.bias) v = v + ld4(p.bias + gn);
        const size_t off = (size_t)phys_row(p, gm) * p.ldc + gn;
        if (p.pre_out) st4(p.pre_out + off, v);
        if (gn >= p.act_lo && gn < p.act_hi) v = silu4(v);
        if (p.gate) v = v * (p.gate_mode ? dsilu4(ld4(p.gate + off)) : ld4(p.gate + off));
        if (p.res) v = ld4(p.res + off) + v;
        st4(p.C + off, v);
    }
}

}  // namespace gn

int gn_gemm_split_launch(gn::GemmArgs p, const unsigned short* W3, void* stream) {
    const __bf16* w3 = reinterpret_cast<const __bf16*>(W3);
    const long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const long grid_big = 8L * (((p.M + 127) / 128 + 7) / 8) * ((p.N + 127) / 128);
    const long grid_small = 8L * (((p.M + 63) / 64 + 7) / 8) * ((p.N + 63) / 64);
    const bool pro = p.pro_mode != 0 || p.a_gate != nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (big >= 384) {
        if (pro) hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<2, 2, true>), dim3((unsigned)grid_big), dim3(256), 0, st, p, w3);
        else hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<2, 2, false>), dim3((unsigned)grid_big), dim3(256), 0, st, p, w3);
    } else {
        if (pro) hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<1, 1, true>), dim3((unsigned)grid_small), dim3(256), 0, st, p, w3);
        else hipLaunchKernelGGL((gn::gemm_bf16x3_mfma<1, 1, false>), dim3((unsigned)grid_small), dim3(256), 0, st, p, w3);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_split_bf16x3(const float* w, long n, unsigned short* out, void* stream) {
    if (n < 0) return GN_ERR_BAD_ARG;
    if (n == 0) return GN_OK;
    hipLaunchKernelGGL(gn::split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (size_t)n, reinterpret_cast<__bf16*>(out));
    GN_LAUNCH_CHECK();
    return GN_OK;
}
