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
    int nt_store;                   // first output column written with non-temporal stores (outputs far larger than the L2s; INT_MAX: none)
    int act_kind;                   // GN_ACT_*: the activation of the epilogue columns, of gate_mode 1 and of the prologues
};

constexpr int GN_MAX_GROUP = 4;
struct GroupArgs {
    GemmArgs g[GN_MAX_GROUP];       // entries >= n repeat the last problem (never selected)
    int tile_end[GN_MAX_GROUP];     // running tile count: problem i owns the global tile ids [tile_end[i-1], tile_end[i])
    int n;
    int spread;                     // 1: every problem's tiles are cut into 8 XCD ranges; 0: one cut of the whole list
};

constexpr int BK = 32, PITCH = 36;

// ---- 3 x bf16 split (gn_gemm_split.hip) ----------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int SPLIT_PB = 40;    // bf16 per LDS row of an A plane (32 + 8 pad: 80-byte pitch, the 16-lane ds_read_b128
                                // groups hit 16 distinct 4-bank slots)

// x = hi + mid + lo EXACTLY, each a bf16 (8 significand bits): hi = x truncated to its top 16 bits, r1 = x - hi is
// exact (<= 16 significant bits), mid = r1 truncated, lo = r1 - mid has <= 8 significant bits left, so its own
// truncation is exact too.  (Truncation instead of round-to-nearest: two bit operations instead of a convert and a
// shift per plane, and no rounding term.)  Results below the fp32 normal range flush like any fp32 subtraction.
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {      // (bf16(a), bf16(b)) by truncation
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ void split4_trunc(float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    const float r0 = v.x - trunc_bf16(v.x), r1 = v.y - trunc_bf16(v.y);
    const float r2 = v.z - trunc_bf16(v.z), r3 = v.w - trunc_bf16(v.w);
    const float s0 = r0 - trunc_bf16(r0), s1 = r1 - trunc_bf16(r1);
    const float s2 = r2 - trunc_bf16(r2), s3 = r3 - trunc_bf16(r3);
    uint2 h, m, l;
    h.x = pack_hi16(v.x, v.y); h.y = pack_hi16(v.z, v.w);
    m.x = pack_hi16(r0, r1);   m.y = pack_hi16(r2, r3);
    l.x = pack_hi16(s0, s1);   l.y = pack_hi16(s2, s3);
    hi = __builtin_bit_cast(bf16x4, h);
    mid = __builtin_bit_cast(bf16x4, m);
    lo = __builtin_bit_cast(bf16x4, l);
}

// ---- 2 x fp16 split with block exponents (MODE 2) --------------------------------------------------------------
// x' = x * 2^-e with |x'| < 2^15 (e: a per-block power of two, see gemm_body), x' = hi + lo with hi = fp16(x') and
// lo = fp16(x' - hi): 22 significand bits in two planes, so an fp32 product needs THREE fp16 MFMAs (hi*hi, hi*lo, lo*hi;
// lo*lo is below 2^-22 of the product) instead of the six of the bf16 split.  fp16 has five exponent bits: elements more
// than 2^17 below their block's maximum keep fewer bits (absolute error <= 2^-39 of the block maximum) -- harmless in a
// dot product, which is dominated by the block's large elements.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// vector forms so that the compiler picks the packed instructions of gfx950: v_pk_mul_f32, v_cvt_pk_f16_f32 (round to
// nearest even), v_pk_add_f32 -- 3 VALU per element instead of 6 with scalar ldexp / convert / subtract
__device__ __forceinline__ void split4_f16(float4 v, float scale /* 2^-e */, f16x4& hi, f16x4& lo) {
    const f32x2 a0 = f32x2{v.x, v.y} * scale, a1 = f32x2{v.z, v.w} * scale;
    const f16x2 h0 = __builtin_convertvector(a0, f16x2), h1 = __builtin_convertvector(a1, f16x2);
    const f16x2 l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x2), f16x2);
    const f16x2 l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x2), f16x2);
    hi = f16x4{h0.x, h0.y, h1.x, h1.y};
    lo = f16x4{l0.x, l0.y, l1.x, l1.y};
}

// wave-wide maximum of an unsigned value without touching LDS: row_shr 1/2/4/8 (zero-filled), then row_bcast 15 and 31;
// the result is read from lane 63 and is wave-uniform (an SGPR)
__device__ __forceinline__ unsigned wave_umax_sgpr(unsigned v) {
    auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int phys_row(const GemmArgs& p, int r) {
    // identity row map (row_cnt = 1: every product except the per-degree W_vk ones): no integer division -- the
    // epilogue calls this once per stored row, and a runtime division is ~40 VALU instructions
    if (p.row_cnt == 1) return r * p.row_gstride + p.row_goff;
    return (r / p.row_cnt) * p.row_gstride + p.row_goff + (r % p.row_cnt);
}



}  // namespace gn

namespace gn {
// ---- column-loop kernel (gn_gemm_colpipe.hip) -----------------------------------------------------------------------
// Work decomposition: a UNIT is (row panel of 64, column pass of 128) of a problem, units are numbered panel-major.
// Workgroup w of G takes the units [w T / G, (w + 1) T / G) (T = all units of the group): every workgroup gets the same MFMA
// work to within one unit, and consecutive units of a workgroup are consecutive column passes of the same row panel.
struct ClArgs {
    GemmArgs g[GN_MAX_GROUP];
    long wend[GN_MAX_GROUP];        // running unit count: problem i owns the units [wend[i-1], wend[i])
    int n;
};
constexpr int CL_KC = 256;          // panel depth (one K chunk)
// workgroup barrier for LDS hand-over that does NOT drain the vector-memory counter (a __syncthreads() waits for every
// outstanding global store and load of the wave)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace gn

// launcher for a group of n <= GN_MAX_GROUP problems (gn_gemm.hip).  split = 0: exact fp32 MFMA, W = fp32 [N][K];
// split = 1: 3 x bf16-split MFMA, W = the fragment-major bf16 planes written by gn_split_bf16x3;
// split = 2: 2 x fp16-split MFMA with block exponents, W = the planes (+ header) written by gn_split_f16x2
int gn_gemm_launch(const gn::GemmArgs* g, int n, hipStream_t st, int split);
// the K-resident panel kernel for small f16x2 groups (gn_gemm_panel.hip): 1 = launched, 0 = does not apply, < 0 = -hipError_t
int gn_gemm_panel_launch(const gn::GemmArgs* g, int n, hipStream_t st);
// the column-loop kernel for large K = 256 f16x2 groups (gn_gemm_colpipe.hip; g[i].nt_store set by the caller): same return convention
int gn_gemm_colpipe_launch(const gn::GemmArgs* g, int n, hipStream_t st);
