// gn_segment.hip -- segment-resident HTR edge weights (K7): the node rows of one closed atom segment live in LDS.
//
// A "segment" is a contiguous atom range [lo, hi) that no edge leaves: every edge whose target lies in the range has
// its source in the range too (one molecule of a batch, or several; gn_graph.hip derives the ranges from the edge list).
// The per-target kernel (gn_gata.hip: htr_edge_kernel) gathers the D rows of EK[src] for every edge through L1/L2 --
// at lmax = 4 that is 24 KB per edge, 1.3 GB per launch through the L2->CU path, which bounds it at 0.24 of the HBM
// roofline.  Here one workgroup owns (segment, channel slice of CS): it stages the segment's EQ and EK rows for its
// channels once ([atoms][D][CS] each, coalesced 128-byte pieces), then walks the segment's edges with 8 lanes per edge
// reading both operands from LDS.  HBM traffic is the algorithmic minimum (each node row once per launch); the
// per-edge gather runs at LDS bandwidth.
// Arithmetic: the literal two-rejection form of htr_edge_kernel, same operation order -> bit-identical weights.
#include <cstdlib>
#include "gn_common.h"

namespace gn {

constexpr int SEG_CS = 32;                   // channels per slice (8 lanes x float4 per edge)
constexpr int SEG_LPE = SEG_CS / 4;          // lanes per edge
// atom stride = 32 banks mod 64: two atoms of different parity read disjoint bank halves (a row is 32 banks wide)
__host__ __device__ constexpr int seg_atom_stride(int D) { return ((D * SEG_CS + 63) / 64) * 64 + 32; }
__host__ __device__ constexpr int seg_threads(int lmax) { return lmax <= 2 ? 1024 : 512; }   // VGPR budget per lane

// persistent workgroups walk the (segment, slice) items: no host read-back of the segment count, no empty workgroups
// holding LDS
template <int LMAX>
__global__ __launch_bounds__(seg_threads(LMAX)) void htr_edge_seg_kernel(
    const float* __restrict__ EQ, const float* __restrict__ EK, const float* __restrict__ rl,
    const int* __restrict__ rowptr, const int* __restrict__ src, const int* __restrict__ dst,
    const int* __restrict__ seg_first, const int* __restrict__ seg_hi, const int* __restrict__ nseg, int F, int cap,
    float* __restrict__ w, int dbg) {
    constexpr int D = (LMAX + 1) * (LMAX + 1) - 1;
    constexpr int AST = seg_atom_stride(D);
    constexpr int NT = seg_threads(LMAX), SLOTS = NT / SEG_LPE;
    extern __shared__ float seg_lds[];
    const int nslices = F / SEG_CS, nitems = nseg[0] * nslices;
    const int slot = threadIdx.x / SEG_LPE, c0 = (threadIdx.x % SEG_LPE) * 4;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int a0 = seg_first[item / nslices], a1 = seg_hi[a0], A = a1 - a0;
        if (A > cap) continue;                             // too large for LDS: the per-target kernel takes it
        const int cbase = (item % nslices) * SEG_CS;
        float* qs = seg_lds;
        float* ks = seg_lds + (size_t)A * AST;
        for (int p = threadIdx.x; p < A * D * SEG_LPE; p += NT) {
            const int row = p / SEG_LPE, c = (p % SEG_LPE) * 4;
            const int at = row / D, m = row - at * D;
            const size_t g = ((size_t)(a0 + at) * D + m) * F + cbase + c;
            const float4 vq = ld4(EQ + g), vk = ld4(EK + g);
            *reinterpret_cast<float4*>(qs + at * AST + m * SEG_CS + c) = vq;
            *reinterpret_cast<float4*>(ks + at * AST + m * SEG_CS + c) = vk;
        }
        __syncthreads();
        const int e0 = rowptr[a0], e1 = rowptr[a1];
        int e = e0 + slot;
        int di = e < e1 ? dst[e] : a0, si = e < e1 ? src[e] : a0;
        for (; e < e1; e += SLOTS) {
            const float* qa = qs + (di - a0) * AST + c0;
            const float* kb = ks + (si - a0) * AST + c0;
            if (e + SLOTS < e1) { di = dst[e + SLOTS]; si = src[e + SLOTS]; }
            const float* re = rl + (size_t)e * D;
            float4 wsum = zero4();
            int m0 = 0;
#pragma unroll
            for (int l = 1; l <= LMAX; ++l) {
                float4 eq[2 * LMAX + 1], ek[2 * LMAX + 1];
                float r[2 * LMAX + 1];
                float4 pq = zero4(), pk = zero4();
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    eq[mm] = *reinterpret_cast<const float4*>(qa + (m0 + mm) * SEG_CS);
                    ek[mm] = *reinterpret_cast<const float4*>(kb + (m0 + mm) * SEG_CS);
                    r[mm] = re[m0 + mm];
                    pq = fma4(r[mm], eq[mm], pq);
                    pk = fma4(-r[mm], ek[mm], pk);
                }
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; ++mm) {
                    const float4 a_ = eq[mm] + pq * (-r[mm]);
                    const float4 b_ = ek[mm] + pk * r[mm];
                    wsum = fma4(a_, b_, wsum);
                }
                m0 += 2 * l + 1;
            }
            st4_nt(w + (size_t)e * F + cbase + c0, wsum);
        }
        __syncthreads();                                   // the next item restages the rows
    }
}

}  // namespace gn

// LDS bytes for a segment of `atoms` atoms
static inline size_t seg_lds_bytes(int lmax, int atoms) {
    const int D = (lmax + 1) * (lmax + 1) - 1;
    return (size_t)2 * atoms * gn::seg_atom_stride(D) * sizeof(float);
}

extern "C" int gn_htr_edge_seg(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                               const int* dst, const int* seg_first, const int* seg_hi, const int* nseg, int N, int F,
                               int lmax, int cap, float* w, void* stream) {
    if (N < 0 || F < gn::SEG_CS || (F % gn::SEG_CS) || lmax < 1 || lmax > 4 || cap < 1) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t lds = seg_lds_bytes(lmax, cap);
    if (lds > 160 * 1024) return GN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nt = gn::seg_threads(lmax);
    const int dbg = getenv("GN_SEG_DBG") ? atoi(getenv("GN_SEG_DBG")) : 0;
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 2048 / nt) per_cu = 2048 / nt;
    long items = (long)N * (F / gn::SEG_CS);               // upper bound (every atom its own segment)
    const dim3 grid((unsigned)(items < 256L * per_cu ? items : 256L * per_cu)), block(nt);
#define GN_SEG_LAUNCH(L)                                                                                              \
    {                                                                                                                 \
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(gn::htr_edge_seg_kernel<L>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        if (once != hipSuccess) return GN_ERR_BAD_ARG;                                                                \
        hipLaunchKernelGGL(gn::htr_edge_seg_kernel<L>, grid, block, lds, st, EQ, EK, rl, rowptr, src, dst, seg_first, \
                           seg_hi, nseg, F, cap, w, dbg);                                                                \
    }
    switch (lmax) {
        case 1: GN_SEG_LAUNCH(1); break;
        case 2: GN_SEG_LAUNCH(2); break;
        case 3: GN_SEG_LAUNCH(3); break;
        default: GN_SEG_LAUNCH(4); break;
    }
#undef GN_SEG_LAUNCH
    GN_LAUNCH_CHECK();
    return GN_OK;
}
