// gn_gemm.hip -- fp32 dense projections on the CDNA4 matrix cores.
//
//   C[r, n] = epi( sum_k A[r, k] * W[n, k] + bias[n] )        W is nn.Linear's [out, in]
//
// Replaces every Dense/MLP on the path (reference layers.py:457-581; call sites
// gotennet.py:400-407, 432-441, 611, 728, 738).  Exact fp32: v_mfma_f32_32x32x2_f32
// is bitwise an fmaf chain, 157.3 TFLOP/s peak on MI355X (no TF32/xf32 on gfx950).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 patch = 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs), K in slabs of 32.
// A and W slabs are staged through LDS as [128][36] floats: the 36-float row pitch
// keeps ds_read_b128 conflict-free for the 16-lane service groups (rows r*36 mod 64
// are 16 distinct 4-bank slots) and keeps 16-byte alignment.  Within a slab the K
// order is permuted so that lanes 0-31 own k in [0,16) and lanes 32-63 own k in
// [16,32): each lane then fetches its MFMA operands as contiguous float4s.  The next
// slab is prefetched into registers while the current one is multiplied.
#include "gn_common.h"

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* W; const float* bias; float* C;
    const float* res; const float* gate;
    int lda, ldc, M, N, K;
    int act_lo, act_hi;
    int row_cnt, row_gstride, row_goff;
};

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 36;

__device__ __forceinline__ int phys_row(const GemmArgs& p, int r) {
    return (r / p.row_cnt) * p.row_gstride + p.row_goff + (r % p.row_cnt);
}

__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];

    // XCD-aware tile order: consecutive row tiles of one column strip share W in L2,
    // consecutive column strips of one row tile share A.  Column tile is the fast index.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: thread -> (row r + 32 i, float4 column c4)
    const int c4 = tid & 7;
    const int sr = tid >> 3;

    const float* arow[4];
    const float* brow[4];
    bool aok[4], bok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + sr + 32 * i;
        aok[i] = gm < p.M;
        arow[i] = p.A + (size_t)phys_row(p, aok[i] ? gm : 0) * p.lda + 4 * c4;
        const int gn = n0 + sr + 32 * i;
        bok[i] = gn < p.N;
        brow[i] = p.W + (size_t)(bok[i] ? gn : 0) * p.K + 4 * c4;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 pa[4], pb[4];
    auto fetch = [&](int k0) {
        const bool kok = (k0 + 4 * c4) < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] = (aok[i] && kok) ? ld4(arow[i] + k0) : zero4();
            pb[i] = (bok[i] && kok) ? ld4(brow[i] + k0) : zero4();
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st4(&As[(sr + 32 * i) * PITCH + 4 * c4], pa[i]);
            st4(&Bs[(sr + 32 * i) * PITCH + 4 * c4], pb[i]);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    fetch(0);
    stash();
    __syncthreads();

    const int khalf = (lane >> 5) * 16;
    const int frow = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fetch((kt + 1) * BK);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float a[2][8], b[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* ap = &As[(wm * 64 + i * 32 + frow) * PITCH + khalf + hh * 8];
                const float* bp = &Bs[(wn * 64 + i * 32 + frow) * PITCH + khalf + hh * 8];
                const float4 a0 = ld4(ap), a1 = ld4(ap + 4), b0 = ld4(bp), b1 = ld4(bp + 4);
                a[i][0] = a0.x; a[i][1] = a0.y; a[i][2] = a0.z; a[i][3] = a0.w;
                a[i][4] = a1.x; a[i][5] = a1.y; a[i][6] = a1.z; a[i][7] = a1.w;
                b[i][0] = b0.x; b[i][1] = b0.y; b[i][2] = b0.z; b[i][3] = b0.w;
                b[i][4] = b1.x; b[i][5] = b1.y; b[i][6] = b1.z; b[i][7] = b1.w;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            stash();
            __syncthreads();
        }
    }

    // epilogue: lane holds column (lane & 31), rows (r&3) + 8 (r>>2) + 4 (lane>>5) of each 32x32 tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
        if (gn >= p.N) continue;
        const float bv = p.bias ? p.bias[gn] : 0.f;
        const bool act = gn >= p.act_lo && gn < p.act_hi;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (act) v = silu(v);
                const size_t off = (size_t)phys_row(p, gm) * p.ldc + gn;
                if (p.gate) v = fmaf(v, p.gate[off], p.res[off]);
                p.C[off] = v;
            }
        }
    }
}

}  // namespace gn

extern "C" int gn_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                       int Mrows, int Nout, int K, int act_lo, int act_hi,
                       int row_cnt, int row_gstride, int row_goff,
                       const float* res, const float* gate, void* stream) {
    if (Mrows < 0 || Nout <= 0 || K <= 0 || (K & 3) || (lda & 3) || row_cnt <= 0) return GN_ERR_BAD_ARG;
    if ((gate == nullptr) != (res == nullptr)) return GN_ERR_BAD_ARG;
    if (Mrows == 0) return GN_OK;
    gn::GemmArgs p{A, W, bias, C, res, gate, lda, ldc, Mrows, Nout, K, act_lo, act_hi, row_cnt, row_gstride, row_goff};
    const int tiles = ((Mrows + gn::BM - 1) / gn::BM) * ((Nout + gn::BN - 1) / gn::BN);
    hipLaunchKernelGGL(gn::gemm_f32_mfma, dim3(tiles), dim3(256), 0, (hipStream_t)stream, p);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
