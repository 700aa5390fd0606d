// gn_gemm.h -- declarations shared by the exact-fp32 (gn_gemm.hip) and the 3xbf16-split
// (gn_gemm_split.hip) projection kernels.
#pragma once
#include "gn_common.h"

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* W; const float* bias; float* C;
    const float* res; const float* gate; float* pre_out;
    const float* a_pre; const float* a_gate;
    int lda, ldc, ldp, ldg, M, N, K;
    int act_lo, act_hi;
    int pro_mode, pro_lo, pro_hi;
    int row_cnt, row_gstride, row_goff;
    int gate_mode;                  // 0: value * gate; 1: value * SiLU'(gate)
    // K-segmented A operand: columns [s * a_seg, (s+1) * a_seg) of the logical A come from A / A2 / A3 (same lda,
    // same row addressing): sums up to three products that share their output rows.  a_seg = 0: plain A.
    const float* A2; const float* A3; int a_seg;
    int nt_store;                   // outputs far larger than the L2s are written with non-temporal stores
};

constexpr int GN_MAX_GROUP = 4;
struct GroupArgs {
    GemmArgs g[GN_MAX_GROUP];       // entries >= n repeat the last problem (never selected)
    int tile_end[GN_MAX_GROUP];     // running tile count: problem i owns the global tile ids [tile_end[i-1], tile_end[i])
    int n;
    int spread;                     // 1: every problem's tiles are cut into 8 XCD ranges; 0: one cut of the whole list
};

constexpr int BK = 32, PITCH = 36;

__device__ __forceinline__ int phys_row(const GemmArgs& p, int r) {
    return (r / p.row_cnt) * p.row_gstride + p.row_goff + (r % p.row_cnt);
}

__device__ __forceinline__ float4 silu4(float4 v) { return make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w)); }
__device__ __forceinline__ float4 dsilu4(float4 v) { return make_float4(dsilu(v.x), dsilu(v.y), dsilu(v.z), dsilu(v.w)); }


}  // namespace gn

// exact-fp32 launcher for a group of n <= GN_MAX_GROUP problems (gn_gemm.hip)
int gn_gemm_launch(const gn::GemmArgs* g, int n, hipStream_t st);

// split kernel launcher (gn_gemm_split.hip); W3 = [3][N][K] bf16 (hi, mid, lo planes)
int gn_gemm_split_launch(gn::GemmArgs p, const unsigned short* W3, void* stream);
