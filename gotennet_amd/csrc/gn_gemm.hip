// gn_gemm.hip -- fp32 dense projections on the CDNA4 matrix cores.
//
//   C[r, n] = epi( sum_k pro(A)[r, k] * W[n, k] + bias[n] )        W is nn.Linear's [out, in]
//
// Replaces every Dense/MLP on the path (reference layers.py:457-581; call sites
// gotennet.py:400-407, 432-441, 611, 728, 738) and, with transposed weights, their
// input-gradients in the force backward.  Exact fp32: v_mfma_f32_32x32x2_f32 is
// bitwise an fmaf chain, 157.3 TFLOP/s peak on MI355X (no TF32/xf32 on gfx950).
//
// Tiling: a 256-thread workgroup (4 waves as 2x2) owns a (64 TM) x (64 TN) output
// tile; each wave a (32 TM) x (32 TN) patch of 32x32 MFMA tiles.  TM = TN = 2 for
// the edge-sized products, TM = TN = 1 when the problem has too few 128x128 tiles to
// fill 256 CUs (atom-sized products: a wave's chain of 64-cycle MFMAs is the latency).
// K in slabs of 32.  A and W slabs are staged through LDS as [rows][36] floats: the
// 36-float pitch keeps ds_read_b128 conflict-free for its 16-lane service groups
// (r*36 mod 64 hits 16 distinct 4-bank slots) and keeps 16-byte alignment.  Within a
// slab the K order is permuted so lanes 0-31 own k in [0,16) and lanes 32-63 own
// k in [16,32): each lane fetches its MFMA operands as contiguous float4s.  The next
// slab is prefetched into registers while the current one is multiplied.
//
// Prologue on A (applied while staging; columns [pro_lo, pro_hi) only):
//   1: A <- SiLU(A)                 (activations are stored pre-activation, consumers apply SiLU)
//   2: A <- A * SiLU'(P)            (backward through an activation; P = stored pre-activation)
//   and, for every column, A <- A * G when a_gate != NULL.
#include "gn_gemm.h"

namespace gn {

template <int TM, int TN, bool PRO, int PF>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM / 32, RB = BN / 32;       // staged float4 rows per thread
    constexpr int STAGE = (BM + BN) * PITCH;        // floats per K-slab buffer (A rows then W rows)
    constexpr int CP = BN + 4;                      // epilogue tile pitch
    constexpr int LDS_FLOATS = (2 * STAGE > BM * CP) ? 2 * STAGE : BM * CP;
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];

    // XCD-aware tile order (block b runs on XCD b % 8, speed only): every XCD owns a contiguous range
    // of row tiles and walks all column tiles of a row tile back to back, so an A row tile is pulled
    // through ONE L2 instead of all eight.  grid = 8 * ceil(tiles_m / 8) * tiles_n; the excess exits.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int xq = tiles_m >> 3, xr = tiles_m & 7, xcd = blockIdx.x & 7;
    const int rows_here = xq + (xcd < xr ? 1 : 0);
    const int row_base = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 7;                         // staging map: thread -> (row sr + 32 i, float4 column c4)
    const int sr = tid >> 3;
    const int stride = gridDim.x >> 3;

    // fetch-side context of a tile (set one tile AHEAD at the end of the K loop: the first slab of the next
    // tile is already in flight while the current tile's epilogue runs)
    int prow[RA];
    const float* brow[RB];
    bool aok[RA], bok[RB];
    auto set_tile = [&](int t_idx) {
        const int m0f = (row_base + t_idx / tiles_n) * BM, n0f = (t_idx % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int gm = m0f + sr + 32 * i;
            aok[i] = gm < p.M;
            prow[i] = phys_row(p, aok[i] ? gm : 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int gn = n0f + sr + 32 * i;
            bok[i] = gn < p.N;
            brow[i] = p.W + (size_t)(bok[i] ? gn : 0) * p.K + 4 * c4;
        }
    };

    f32x16 acc[TM][TN];

    // PF register sets of prefetched slabs: a fetched slab stays in flight for PF compute phases (PF = 1 is
    // what ships: PF = 2/3 measured no gain on MI355X and costs a wave of occupancy)
    float4 pa[PF][RA], pb[PF][RB];
    auto fetch = [&](int k0, float4 (&qa)[RA], float4 (&qb)[RB]) {
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
            qa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) qb[i] = (bok[i] && kok) ? ld4(brow[i] + k0) : zero4();
    };
    auto stash = [&](float* buf, const float4 (&qa)[RA], const float4 (&qb)[RB]) {
#pragma unroll
        for (int i = 0; i < RA; ++i) st4(&buf[(sr + 32 * i) * PITCH + 4 * c4], qa[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) st4(&buf[(BM + sr + 32 * i) * PITCH + 4 * c4], qb[i]);
    };

    // double-buffered LDS K loop, one barrier per slab; register set s holds slab kt+1 when slab kt is
    // multiplied and is refilled with slab kt+1+PF right after it has been written to LDS
    const int nk = (p.K + BK - 1) / BK;
    int idx = blockIdx.x >> 3;
    if (idx / tiles_n >= rows_here) return;
    set_tile(idx);
    fetch(0, pa[0], pb[0]);
  // persistent over tiles: workgroup b walks the tiles b/8, b/8 + gridDim/8, ... of ITS XCD's range
  for (;;) {
    const int m0 = (row_base + idx / tiles_n) * BM;
    const int n0 = (idx % tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    stash(smem, pa[0], pb[0]);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (1 + s < nk) fetch((1 + s) * BK, pa[s], pb[s]);

    const int khalf = (lane >> 5) * 16;
    const int frow = lane & 31;
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int kt = kt0 + s;
        if (kt >= nk) break;
        const float* As = smem + (kt & 1) * STAGE;
        const float* Bs = As + BM * PITCH;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float* ap = &As[(wm * 32 * TM + i * 32 + frow) * PITCH + khalf + hh * 8];
                const float4 a0 = ld4(ap), a1 = ld4(ap + 4);
                a[i][0] = a0.x; a[i][1] = a0.y; a[i][2] = a0.z; a[i][3] = a0.w;
                a[i][4] = a1.x; a[i][5] = a1.y; a[i][6] = a1.z; a[i][7] = a1.w;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* bp = &Bs[(wn * 32 * TN + j * 32 + frow) * PITCH + khalf + hh * 8];
                const float4 b0 = ld4(bp), b1 = ld4(bp + 4);
                b[j][0] = b0.x; b[j][1] = b0.y; b[j][2] = b0.z; b[j][3] = b0.w;
                b[j][4] = b1.x; b[j][5] = b1.y; b[j][6] = b1.z; b[j][7] = b1.w;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stash(smem + ((kt + 1) & 1) * STAGE, pa[s], pb[s]);   // other buffer: last read in iteration kt-1
        __syncthreads();
        if (kt + 1 + PF < nk) fetch((kt + 1 + PF) * BK, pa[s], pb[s]);
      }
    }

    // next tile's first slab goes in flight now and lands during the epilogue
    const int next = idx + stride;
    const bool has_next = next / tiles_n < rows_here;
    if (has_next) {
        set_tile(next);
        fetch(0, pa[0], pb[0]);
    }

    // epilogue through LDS: accumulators -> [BM][BN+4] tile -> coalesced float4 rows
    // (lane holds column (lane & 31), rows (r&3) + 8 (r>>2) + 4 (lane>>5) of each 32x32 tile)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                smem[row * CP + wn * 32 * TN + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    {
        constexpr int C4 = BN / 4;                  // 256 % C4 == 0: a thread keeps ONE column group
        const int cc = (tid % C4) * 4, gn = n0 + cc;
        if (gn < p.N) {
            const float4 bias4 = p.bias ? ld4(p.bias + gn) : zero4();
            const bool act = gn >= p.act_lo && gn < p.act_hi;          // act ranges are multiples of 4
            if (p.res || p.gate) {
                // four rows per pass, all residual / gate loads of the pass issued BEFORE its first store: the
                // stores may alias them as far as the compiler knows (res == C is allowed), and a load behind
                // a store was one exposed round trip per row
                constexpr int ITS = (BM * C4) / 256, UN = ITS < 4 ? ITS : 4;
                for (int it0 = 0; it0 < ITS; it0 += UN) {
                    size_t off[UN];
                    bool ok[UN];
                    float4 rv[UN], gv[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int gm = m0 + (it0 + u) * (256 / C4) + tid / C4;
                        ok[u] = gm < p.M;
                        off[u] = (size_t)phys_row(p, ok[u] ? gm : 0) * p.ldc + gn;
                        rv[u] = (ok[u] && p.res) ? ld4(p.res + off[u]) : zero4();
                        gv[u] = (ok[u] && p.gate) ? ld4(p.gate + off[u]) : zero4();
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        if (!ok[u]) continue;
                        const int row = (it0 + u) * (256 / C4) + tid / C4;
                        float4 v = ld4(&smem[row * CP + cc]) + bias4;
                        if (p.pre_out) st4(p.pre_out + off[u], v);
                        if (act) v = silu4(v);
                        if (p.gate) v = v * (p.gate_mode ? dsilu4(gv[u]) : gv[u]);
                        if (p.res) v = rv[u] + v;
                        st4(p.C + off[u], v);
                    }
                }
            } else {
#pragma unroll 4
                for (int it = 0; it < (BM * C4) / 256; ++it) {
                    const int row = it * (256 / C4) + tid / C4;
                    const int gm = m0 + row;
                    if (gm >= p.M) continue;
                    float4 v = ld4(&smem[row * CP + cc]) + bias4;
                    const size_t off = (size_t)phys_row(p, gm) * p.ldc + gn;
                    if (p.pre_out) st4(p.pre_out + off, v);
                    if (act) v = silu4(v);
                    st4(p.C + off, v);
                }
            }
        }
    }
    __syncthreads();                                // LDS is reused by the next tile's first slab
    if (!has_next) break;
    idx = next;
  }
}

}  // namespace gn

extern "C" int gn_gemm_ex(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                          int Mrows, int Nout, int K, int act_lo, int act_hi,
                          int row_cnt, int row_gstride, int row_goff,
                          const float* res, const float* gate, int gate_mode, float* pre_out,
                          int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                          const float* a_gate, int ldg, void* stream) {
    if (Mrows < 0 || Nout <= 0 || K <= 0 || (K & 3) || (lda & 3) || (Nout & 3) || (ldc & 3) || (act_lo & 3) ||
        (act_hi & 3) || row_cnt <= 0)
        return GN_ERR_BAD_ARG;
    if (gate != nullptr && res == nullptr && gate_mode == 0) return GN_ERR_BAD_ARG;
    if (pro_mode < 0 || pro_mode > 2 || (pro_mode == 2 && (!a_pre || (ldp & 3))) || (a_gate && (ldg & 3)) ||
        (pro_mode && ((pro_lo & 3) || (pro_hi & 3))))
        return GN_ERR_BAD_ARG;
    if (Mrows == 0) return GN_OK;
    gn::GemmArgs p{A, W, bias, C, res, gate, pre_out, a_pre, a_gate, lda, ldc, ldp, ldg, Mrows, Nout, K,
                   act_lo, act_hi, pro_mode, pro_lo, pro_hi, row_cnt, row_gstride, row_goff, gate_mode};
    const long big = (long)((Mrows + 127) / 128) * ((Nout + 127) / 128);
    long grid_big = 8L * (((Mrows + 127) / 128 + 7) / 8) * ((Nout + 127) / 128);
    long grid_small = 8L * (((Mrows + 63) / 64 + 7) / 8) * ((Nout + 63) / 64);
    const bool pro = pro_mode != 0 || a_gate != nullptr;
    hipStream_t st = (hipStream_t)stream;
    // persistent launch: at most 2 (big tiles) / 4 (small tiles) workgroups per CU walk the tile list (+2 %)
    if (grid_big > 512) grid_big = 512;
    if (grid_small > 1024) grid_small = 1024;
    if (big >= 384) {                          // measured best switch-over (tools/gemm_bench.py sweep)
        if (pro) hipLaunchKernelGGL((gn::gemm_f32_mfma<2, 2, true, 1>), dim3((unsigned)grid_big), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gn::gemm_f32_mfma<2, 2, false, 1>), dim3((unsigned)grid_big), dim3(256), 0, st, p);
    } else {
        if (pro) hipLaunchKernelGGL((gn::gemm_f32_mfma<1, 1, true, 1>), dim3((unsigned)grid_small), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gn::gemm_f32_mfma<1, 1, false, 1>), dim3((unsigned)grid_small), dim3(256), 0, st, p);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_gemm_split(const float* A, int lda, const unsigned short* W3, const float* bias, float* C, int ldc,
                             int Mrows, int Nout, int K, int act_lo, int act_hi,
                             int row_cnt, int row_gstride, int row_goff,
                             const float* res, const float* gate, int gate_mode, float* pre_out,
                             int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                             const float* a_gate, int ldg, void* stream) {
    if (Mrows < 0 || Nout <= 0 || K <= 0 || (K & 7) || (lda & 3) || (Nout & 3) || (ldc & 3) || (act_lo & 3) ||
        (act_hi & 3) || row_cnt <= 0)
        return GN_ERR_BAD_ARG;
    if (gate != nullptr && res == nullptr && gate_mode == 0) return GN_ERR_BAD_ARG;
    if (pro_mode < 0 || pro_mode > 2 || (pro_mode == 2 && (!a_pre || (ldp & 3))) || (a_gate && (ldg & 3)) ||
        (pro_mode && ((pro_lo & 3) || (pro_hi & 3))))
        return GN_ERR_BAD_ARG;
    if (Mrows == 0) return GN_OK;
    gn::GemmArgs p{A, nullptr, bias, C, res, gate, pre_out, a_pre, a_gate, lda, ldc, ldp, ldg, Mrows, Nout, K,
                   act_lo, act_hi, pro_mode, pro_lo, pro_hi, row_cnt, row_gstride, row_goff, gate_mode};
    return gn_gemm_split_launch(p, W3, stream);
}

extern "C" int gn_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                       int Mrows, int Nout, int K, int act_lo, int act_hi,
                       int row_cnt, int row_gstride, int row_goff,
                       const float* res, const float* gate, void* stream) {
    return gn_gemm_ex(A, lda, W, bias, C, ldc, Mrows, Nout, K, act_lo, act_hi, row_cnt, row_gstride, row_goff,
                      res, gate, 0, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, stream);
}
