/*
 * gotennet_hip.h -- C ABI of libgotennet_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for ONE path of sarpaykent/GotenNet: the equivariant
 * interaction stack GotenNet.forward(atomic_numbers, edge_index, edge_diff,
 * edge_vec) -> (h, X)   (reference gotennet/models/representation/gotennet.py:956-1010).
 * The reference has no FFI of its own (it is pure Python on ATen); each entry
 * point below replaces the run of ATen/PyG ops cited next to it, and is what a
 * ctypes binding on the reference side would bind.
 *
 * Contract:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch
 *     `tensor.data_ptr()`), fp32 row-major with the feature axis F fastest;
 *     indices are int32 unless stated; the library never allocates, frees or
 *     synchronises (safe inside hipGraph capture);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - return value: 0 on success, otherwise a hipError_t (launch errors) or
 *     GN_ERR_* (argument errors); no C++ exceptions cross the ABI;
 *   - re-entrant, no global mutable state: no entry point reads the process environment or a mutable static; every
 *     choice of kernel is a function of the arguments (GN_LMAX_SLICED below) or a build-time constant.
 *
 * Symbols: N atoms, E directed edges in CSR-by-target order (edge e runs
 * src[e] = j  ->  dst = i, rowptr[i] <= e < rowptr[i+1]), F = n_atom_basis,
 * H = num_heads, lmax in [1,8] (the range of the reference's TensorInit, layers.py:805-1494), D = (lmax+1)^2 - 1,
 * M = multiplier (number of F-wide blocks in the value vector, gotennet.py:197-203), R = n_rbf.
 * lmax <= 4 runs the register-tiled kernels (the benchmarked configurations); lmax 5..8 the degree-sliced ones
 * (gn_highl.hip: one launch per degree, same arithmetic per row, same fixed-order reductions) behind the SAME
 * entry points.
 */
#ifndef GOTENNET_HIP_H
#define GOTENNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_OK 0
#define GN_ERR_BAD_ARG 10001     /* shape/flag combination the kernels do not implement */
#define GN_ABI_VERSION 8
/* OR-ed into the `lmax` ARGUMENT of gn_message_aggregate, gn_message_backward(_groups), gn_htr_edge and
 * gn_htr_backward: run this call on the degree-sliced kernel family at lmax <= 4 as well (the family that serves
 * lmax 5..8).  An explicit per-call request -- the library reads no environment variable and keeps no switch; every
 * tuning value of the launchers is a build-time constant (csrc/gn_tune.h). */
#define GN_LMAX_SLICED 0x100
/* The reference's `aggr` constructor argument (gotennet.py:84,129,638: PyG scatter reduce of the messages), also carried in
 * the `lmax` argument: default "add"; GN_LMAX_MEAN = "mean" (sum / in-degree, 0 for atoms without incoming edges) in
 * gn_message_aggregate AND gn_message_backward; GN_LMAX_MAX = "max" (element-wise maximum over the incoming edges, 0 without)
 * in gn_message_aggregate only (no backward: GN_ERR_BAD_ARG).  Both run on the degree-sliced kernel family at every lmax. */
#define GN_LMAX_MEAN 0x200
#define GN_LMAX_MAX 0x400

/* `act`: the element-wise activation of the reference's `activation` constructor argument (str2act, layers.py:596-700;
 * shifted_softplus layers.py:40-50).  Wherever an entry point below says SiLU, it means this kind. */
enum {
    GN_ACT_SILU = 0, GN_ACT_SSP = 1, GN_ACT_RELU = 2, GN_ACT_TANH = 3, GN_ACT_SIGMOID = 4, GN_ACT_ELU = 5,
    GN_ACT_SELU = 6, GN_ACT_MISH = 7, GN_ACT_GELU = 8, GN_ACT_SOFTPLUS = 9, GN_ACT_LEAKY = 10,
    GN_ACT_NONE = 11 /* identity: a head without hidden layer */, GN_ACT_COUNT = 12
};

/* Library identity: returns GN_ABI_VERSION; *arch_out (if non-NULL) receives "gfx950". */
int gn_abi_version(const char** arch_out);

/* ---- graph plumbing ------------------------------------------------------------------ */

/* CSR row pointer from the target column of a target-sorted edge list, plus int32
 * copies of both columns.  edge_index is the reference's int64 [2,E] tensor
 * (row 0 = source j, row 1 = target i; gotennet.py:412-424 gathers _j/_i this way).
 * Replaces PyG's implicit index handling in MessagePassing.propagate. */
int gn_build_csr(const int64_t* edge_index, int E, int N, int* src, int* dst, int* rowptr, void* stream);

/* Validation of a caller-supplied edge list (the reference gets this for free from PyG's scatter, which accepts any
 * order; the CSR kernels here need target-major order): *flag (zeroed by the caller) |= 1 when edge_index[1] is not
 * non-decreasing, |= 2 when any index lies outside [0, N).  GotenNet.forward / EnergyForces sort or raise on it. */
int gn_check_edges(const int64_t* edge_index, int E, int N, int* flag, void* stream);

/* By-source (CSC) view of a target-major edge list for the force backward's by-source passes: colptr [N + 1],
 * perm [E] = the CSR edge ids of every source in increasing order (what a stable sort of `src` gives), tgt_by_src [E] =
 * dst[perm].  Built without a sort (count, one-workgroup scan, scatter, per-bucket ranking): four small launches and a
 * memset instead of the ~18 of torch.sort / index_add_ / cumsum.  work: N + E ints of scratch.  Integer arithmetic only. */
int gn_build_csc(const int* src, const int* dst, int E, int N, int* colptr, int* perm, int* tgt_by_src, int* work,
                 void* stream);

/* Offsets of the molecules in a SORTED int64 batch vector (the heads' segment boundaries, outputs.py:349-357 scatter over
 * `batch`): mol_ptr [n_mol + 1], mol_ptr[m] = first atom whose batch index is >= m.  One launch, no host read. */
int gn_molecule_ptr(const int64_t* batch, int N, int n_mol, int* mol_ptr, void* stream);

/* Out-degree of every node counted over ALL edges incl. self-loops
 * (gotennet.py:986-989: scatter(ones, edge_index[0])).  outdeg must be zeroed by the caller. */
int gn_out_degree(const int* src, int E, int* outdeg, void* stream);

/* CosineCutoff alone (layers.py:149-152): cut[e] = 0.5 (cos(pi d / cutoff) + 1) for d < cutoff, else 0.  Used when a
 * caller drives ONE GATA layer (GATA.forward, gotennet.py:366-450, takes r_ij and applies the cutoff inside message). */
int gn_cosine_cutoff(const float* dist, int E, float cutoff, float* cut, void* stream);

/* ---- K1 edge geometry ---------------------------------------------------------------- */
/* unit vector on non-self edges (gotennet.py:978-980), TensorInit real harmonics
 * (layers.py:805-902), ExpNormalSmearing (layers.py:744-746), CosineCutoff (layers.py:149-152).
 * rl [E,D], phi [E,R], cut [E]. edge_vec is NOT modified. */
int gn_edge_geometry(const float* edge_vec, const float* edge_diff, const int* src, const int* dst, int E,
                     int lmax, int R, int basis, const float* means, const float* betas, float cutoff,
                     float* rl, float* phi, float* cut, void* stream);

/* ---- K2/K3 initialisation ------------------------------------------------------------ */
/* The two radial projections W_ndp phi + b (layers.py:1668) and W_erp phi + b (layers.py:1710) are
 * plain gn_gemm calls on phi [E,R]; `feat` below is a row of that product (leading dim ldf). */

/* NodeInit message + aggregate (layers.py:1658-1675) fused with the A_na embedding
 * lookup (gotennet.py:973): ctx[n, 0:F] = A_na[z[n]],
 * ctx[n, F:2F] = sum_{j->n, j!=n} A_nbr[z_j] * (feat_e * cut_e).  ctx is [N, 2F]. */
int gn_node_init(const int* z, const int* rowptr, const int* src, const float* feat, int ldf,
                 const float* cut, const float* A_na, const float* A_nbr,
                 int N, int F, float* ctx, void* stream);

/* EdgeInit.message (layers.py:1709-1710): t[e] = (h[i] + h[j]) * feat_e for all edges. */
int gn_edge_init(const float* h, const int* rowptr, const int* src, const float* feat, int ldf,
                 int N, int F, float* t, void* stream);

/* y = SiLU(LayerNorm(x) * gamma + beta) row-wise over F (Dense with norm='layer',
 * layers.py:518-529, inside NodeInit's W_nrd_nru).  In place allowed (y == x). */
int gn_layernorm_silu(const float* x, const float* gamma, const float* beta, float eps,
                      int N, int F, float* y, int act /* GN_ACT_* */, void* stream);

/* ---- optional GATA input norms (gotennet.py:305-315, 397-398; off by default) ----------- */
/* y = LayerNorm(x) * gamma + beta (nn.LayerNorm(F), `layernorm != ""`). */
int gn_layernorm(const float* x, const float* gamma, const float* beta, float eps,
                 int N, int F, float* y, void* stream);
/* TensorLayerNorm (layers.py:1497-1563, `steerable_norm != ""`): per atom and degree block l,
 * s_f = |X_l[:, f]|, c_f = max(s_f, eps), n_f = (c_f - min_f c) / (max_f c - min_f c) (denominator 1 if equal),
 * Y = relu(n_f) * X / c_f * weight_f.  X, Y are [N,D,F]; eps = 1e-12 in the reference. */
int gn_tensor_norm(const float* X, const float* weight, float eps, int N, int F, int lmax, float* Y, void* stream);

/* ---- dense projections (K4/K5/K7/K8), fp32 MFMA --------------------------------------- */
/* C[r, n] = epi( sum_k A[r, k] * W[n, k] + bias[n] )      (nn.Linear layout W[out,in]; Dense, layers.py:457-529)
 *   rows:  logical row r in [0, Mrows) maps to physical row (r / row_cnt) * row_gstride + row_goff + r % row_cnt
 *          in both A (leading dim lda) and C (ldc); pass (1,1,0) for the identity.  This addresses the
 *          degree-l' rows of an [N, D, F] tensor for W_vk[l'] (gotennet.py:432-441).
 *   epi:   SiLU on output columns [act_lo, act_hi); then, if gate != NULL,
 *          C = res + value * gate  (res, gate: same addressing as C; HTR update t += SiLU(W t + b) * w,
 *          gotennet.py:445,611).  bias may be NULL.  K must be a multiple of 4. */
int gn_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
            int Mrows, int Nout, int K, int act_lo, int act_hi,
            int row_cnt, int row_gstride, int row_goff,
            const float* res, const float* gate, void* stream);

/* Extended form used by the pipeline.  Extra epilogue: pre_out (if non-NULL) receives the value BEFORE
 * the activation (same addressing as C); `res` alone (gate == NULL) gives C = res + value; gate_mode 1
 * multiplies by SiLU'(gate) instead of gate (backward through an activation, applied once per output
 * element in the PRODUCER's epilogue; res may then be NULL).  Prologue on A,
 * applied while staging, on A columns [pro_lo, pro_hi): pro_mode 1: A <- SiLU(A) (activations are stored
 * pre-activation); pro_mode 2: A <- A * SiLU'(a_pre) (backward through a SiLU; a_pre has A's row addressing
 * with leading dim ldp); and A <- A * a_gate on every column when a_gate != NULL (leading dim ldg). */
int gn_gemm_ex(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
               int Mrows, int Nout, int K, int act_lo, int act_hi,
               int row_cnt, int row_gstride, int row_goff,
               const float* res, const float* gate, int gate_mode, float* pre_out,
               int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
               const float* a_gate, int ldg, int act_kind /* GN_ACT_* */, void* stream);

/* Several INDEPENDENT gn_gemm_ex problems (Dense products, layers.py:457-529; call sites gotennet.py:400-407, 432-441,
 * 611, 728, 738) in one launch (no problem may read what another writes).  The atom-sized
 * products of a layer (x / v, EQ / EK_l / X W_vu^T, and the matching input-gradient products) fill a fraction of the
 * 256 CUs one at a time; a group walks all their tiles with one persistent grid.  n <= 4; every field has the meaning
 * of the gn_gemm_ex argument of the same name. */
typedef struct gn_gemm_desc {
    const float* A; int lda;
    const float* W; const float* bias;
    float* C; int ldc;
    int M, N, K;
    int act_lo, act_hi;
    int row_cnt, row_gstride, row_goff;
    const float* res; const float* gate; int gate_mode;
    float* pre_out;
    int pro_mode, pro_lo, pro_hi;
    const float* a_pre; int ldp;
    const float* a_gate; int ldg;
    /* K-segmented A: when a_seg != 0 the logical A columns [s a_seg, (s+1) a_seg) come from A, A2, A3 (s = 0, 1, 2;
     * same lda and row addressing; a_seg % 32 == 0; no prologue): C = res + A W_0^T + A2 W_1^T + A3 W_2^T with the
     * three weights concatenated along K.  Used for gX = gX + gXp W_vu + gEQ W_vq + gEK_l W_vk_l. */
    const float* A2; const float* A3; int a_seg;
    int act_kind;                /* GN_ACT_*: activation of the [act_lo, act_hi) columns, of gate_mode 1 and of the prologues */
} gn_gemm_desc;
int gn_gemm_group(const gn_gemm_desc* problems, int n, void* stream);

/* 3 x bf16 split variant of the same Dense products (layers.py:457-529): same contract as
 * gn_gemm_ex / gn_gemm_group, but every weight is passed as the bf16 planes written ONCE per weight by
 * gn_split_bf16x3 (W = hi + mid + lo exactly, each plane bf16; stored in the order the matrix cores consume them:
 * [column block of 32][k-step of 16, padded to an even count][plane][lane][8], gn_split_bf16x3_size(N, K) bf16
 * elements).  The product is accumulated in fp32 from the six plane pairs of order <= 2 on the bf16 matrix cores
 * (v_mfma_f32_32x32x16_bf16): fp32-class error (<= 1e-6 relative to an fp64 product) at 16/6 of the exact-fp32
 * MFMA rate.  A is split on the fly; K must be a multiple of 4 as for gn_gemm_ex. */
long gn_split_bf16x3_size(int N, int K);
int gn_split_bf16x3(const float* w /* [N][K] fp32 */, int N, int K, unsigned short* out, void* stream);
int gn_gemm_split(const float* A, int lda, const unsigned short* W3, const float* bias, float* C, int ldc,
                  int Mrows, int Nout, int K, int act_lo, int act_hi,
                  int row_cnt, int row_gstride, int row_goff,
                  const float* res, const float* gate, int gate_mode, float* pre_out,
                  int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                  const float* a_gate, int ldg, int act_kind, void* stream);
/* gn_gemm_group on the split path: every problems[i].W points to gn_split_bf16x3 planes (cast to const float*). */
int gn_gemm_group_split(const gn_gemm_desc* problems, int n, void* stream);

/* 2 x fp16 split variant with block exponents: the same contract again, every weight passed as the buffer written
 * ONCE per weight by gn_split_f16x2 (a 256-byte header holding the tensor's binary exponent, then two fp16 planes
 * hi + lo of w * 2^-exponent in the same fragment-major order; gn_split_f16x2_size(N, K) 16-bit elements; the amax is
 * found on the device, no host read-back).  A is scaled per (staging wave, K-slab of 32 columns) by the running maximum
 * of that block's binary exponent (exact powers of two, accumulators rescaled when it grows): wave q of four stages rows
 * 8q..8q+7 of every 32-row MFMA tile of the workgroup tile, so 16 rows (64-row tiles) or 32 rows (128-row tiles) share
 * an exponent -- results are bit-reproducible but depend at the 1e-7 level on which rows are neighbours (INTEGRATION.md:
 * batch-position contract; per-row bound: tests/test_hip_parity.py::test_fp16_block_exponent_per_row_bound).  Each scaled
 * operand is split into two fp16 planes, and
 * the product accumulated in fp32 from THREE plane pairs on v_mfma_f32_32x32x16_f16 -- half the matrix work of the
 * bf16 split; 22 significand bits per operand: <= 2e-7 of the output's max-norm against an fp64 product. */
long gn_split_f16x2_size(int N, int K);
int gn_split_f16x2(const float* w /* [N][K] fp32 */, int N, int K, unsigned short* out, void* stream);
int gn_gemm_f16x2(const float* A, int lda, const unsigned short* W2, const float* bias, float* C, int ldc,
                  int Mrows, int Nout, int K, int act_lo, int act_hi,
                  int row_cnt, int row_gstride, int row_goff,
                  const float* res, const float* gate, int gate_mode, float* pre_out,
                  int pro_mode, int pro_lo, int pro_hi, const float* a_pre, int ldp,
                  const float* a_gate, int ldg, int act_kind, void* stream);
int gn_gemm_group_f16x2(const gn_gemm_desc* problems, int n, void* stream);

/* ---- K6 GATA message / softmax / aggregate -------------------------------------------- */
/* Attention weights (gotennet.py:497-511): s[e,h] = sum_{c in head h} q[i,c] k[j,c] t_attn[e,c];
 * a = exp(s - max) / (sum + 1e-16) over the incoming edges of i (PyG softmax), then
 * a *= 1/sqrt(F)  or  sqrt(outdeg[j])/sqrt(F) when outdeg != NULL (scale_edge).  a is [E,H].
 * q,k are rows of ldqk floats; t_attn rows of ldt floats hold the PRE-activation W_re t + b
 * (the kernel applies SiLU while loading). */
int gn_attn_softmax(const float* q, const float* k, int ldqk, const float* t_attn, int ldt,
                    const int* rowptr, const int* src, const int* outdeg,
                    int N, int F, int H, float* a, int act /* GN_ACT_*: t_attn = act(.) */, void* stream);

/* Message + segmented reduction + residual (gotennet.py:516-559, 613-640, 426-427):
 *   o[c]   = t_filter[e,c] * x[j,c] * cut[e] + a[e, c / (M F / H)] * v[j,c],  c in [0, M F)
 *   h_out[i]     = h_in[i] + sum_e o[0:F]
 *   X_out[i,m,:] = X_in[i,m,:] + sum_e ( rl[e,m] * o[dblk(l(m))] + X_in[j,m,:] * o[tblk(l(m))] )
 * with dblk/tblk the F-wide block of the direction / tensor gate of degree l (sep_dir / sep_tensor).
 * x, v rows of ldxv floats; t_filter rows of ldt floats.  X_out must not alias X_in.
 * X_in == NULL (lmax <= 4) means "X_in is identically zero" -- the first interaction of GotenNet.forward, which starts
 * from X = 0 (gotennet.py:992): the tensor-gate blocks of t_filter / x / v are then NOT READ (they may be unwritten)
 * and X_out = the aggregated update; same bits as passing a zero tensor. */
int gn_message_aggregate(const float* x, const float* v, int ldxv, const float* t_filter, int ldt,
                         const float* a, const float* rl, const float* cut,
                         const int* rowptr, const int* src,
                         const float* h_in, const float* X_in, float* h_out, float* X_out,
                         int N, int F, int H, int lmax, int sep_dir, int sep_tensor, void* stream);

/* ---- EQFF node chains as one kernel each (gotennet.py:716-748 after X_p = X W_vu^T) ------------------------------ */
/* Forward: n = sqrt(sum_D X_p^2 + eps); [m1 | m2] = W_1 SiLU(W_0 [h | n] + b_0) + b_1; h += m1; X += m2 * X_p -- the
 * sequence gn_eqff_context -> gn_gemm(gamma_m.0) -> gn_gemm(gamma_m.1) -> gn_eqff_update as ONE launch (16 atoms per
 * workgroup, both products on MFMA in the plane arithmetics, W0p / W1p = planes written by gn_split_bf16x3 /
 * gn_split_f16x2 of gamma_m.0.weight [F, 2F] and gamma_m.1.weight [2F, F]).  ctx_out [N,2F], pre_out [N,F] (the
 * pre-activation of the hidden layer), mm_out [N,2F]: what the backward needs, or NULL.
 * Backward: gn_eqff_backward_a -> W_1^T -> W_0^T -> gn_eqff_backward_b as one launch: given g_h, g_X [N,D,F] of the
 * block's outputs returns g_Xp [N,D,F] (gradient w.r.t. X_p, not yet through W_vu) and g_h1 = g_h + d/dh through gamma_m;
 * W1Tp / W0Tp = planes of the TRANSPOSED weights ([F, 2F] and [2F, F]).  g_Xp must not alias g_X.
 * Supported: F in {128, 256}, SiLU, arith 1 (3 x bf16) or 2 (2 x fp16); otherwise callers keep the launch sequence. */
int gn_eqff_fused_supported(int F, int act, int arith);
int gn_eqff_fused_forward(const float* Xp, const void* W0p, const float* b0, const void* W1p, const float* b1,
                          float eps, int N, int F, int D, float* h, float* X,
                          float* ctx_out, float* pre_out, float* mm_out, int arith, void* stream);
int gn_eqff_fused_backward(const float* gh, const float* gX, const float* mm, const float* Xp, const float* ctx,
                           const float* pre_g1, const void* W1Tp, const void* W0Tp, int N, int F, int D,
                           float* gXp, float* gh1, int arith, void* stream);

/* ---- K7 HTR edge weights -------------------------------------------------------------- */
/* w[e,f] = sum_l sum_m P(EQ[i])_m * P(EK[j])_m with P(a) = a - (a . rl_l) rl_l per degree block
 * (gotennet.py:351-364, 580-609; sep_htr=True, rejection on).  EQ, EK are [N,D,F]; w is [E,F]. */
/* `mode` selects the reference's non-default edge-update variants (gotennet.py:139-190, 285-291, 580-599):
 *   GN_HTR_JOINT  sep_htr=False: one rejection block over all D rows instead of one per degree;
 *   GN_HTR_NOREJ  "norej": no vector rejection;
 *   GN_HTR_GATE_* gamma_w applied to w: "gated" sigmoid, "gatedt" tanh, "act" SiLU.
 * mode = 0 is the default path.  w_raw (optional, may be NULL) receives w before the gate (the backward needs it
 * when a gate is set). */
#define GN_HTR_JOINT 1
#define GN_HTR_NOREJ 2
#define GN_HTR_GATE_SIGMOID (1 << 2)
#define GN_HTR_GATE_TANH (2 << 2)
#define GN_HTR_GATE_SILU (3 << 2)
#define GN_HTR_DIRECT 16 /* gn_htr_backward only: g_t_out already is dL/dw [E,F]; pre_t, w, w_raw, g_pre_t unused */
int gn_htr_edge(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                int N, int F, int lmax, int mode, float* w_raw, float* w, void* stream);

/* ---- K8 EQFF node-local pieces -------------------------------------------------------- */
/* ctx[n, 0:F] = h[n]; ctx[n, F:2F] = sqrt(sum_m Xp[n,m,:]^2 + eps)   (gotennet.py:731-735) */
int gn_eqff_context(const float* h, const float* Xp, float eps, int N, int F, int D, float* ctx, void* stream);
/* h += m[:, 0:F];  X += m[:, F:2F] (broadcast over D) * Xp           (gotennet.py:741-746); m is [N,2F] */
int gn_eqff_update(const float* m, const float* Xp, int N, int F, int D, float* h, float* X, void* stream);

/* ---- K10 force backward (input gradients only; what torch.autograd.grad does for the reference at
 *      outputs.py:365-375).  Needs the by-source (CSC) view: tgt_by_src[pp] is the TARGET of by-source entry pp
 *      (= dst[perm[pp]], stored so that the source passes chase one index less per edge) and
 *      perm[colptr[j] .. colptr[j+1]) lists the CSR
 *      edge ids whose source is j.  F <= 256.  Every kernel WRITES its g_rl [E,D] / g_cut [E] contribution to
 *      the slice it is given (plain stores, no read-modify-write); gn_edge_geometry_backward sums the slices. */

/* HTR (gotennet.py:561-611) backward: g_t_out = dL/dt' [E,F], pre_t = W_t t + b and w (saved), w.r.t. EQ, EK,
 * rl, and g_pre_t = g_t_out * w * SiLU'(pre_t) [E,F] (the operand of the W_t^T product). */
int gn_htr_backward(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                    const float* EQ, const float* EK,
                    const float* rl, const int* rowptr, const int* src, const int* tgt_by_src,
                    const int* colptr, const int* perm, int N, int F, int lmax, int mode,
                    float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act /* GN_ACT_* of gamma_t */, void* stream);

/* GATA message/softmax/aggregate (gotennet.py:452-559, 613-640) backward.  Inputs: saved x, v [N,MF];
 * eproj [E,(1+M)F] = (pre-activation of t_attn | t_filter); a [E,H]; qk rows with q at column 0 and k at
 * column F; X_in [N,D,F]; upstream g_h1 [N,F], g_X1 [N,D,F].  Outputs: g_eproj [E,(1+M)F] (gradient w.r.t.
 * the t_attn pre-activation | t_filter), g_s [E,H] scratch, g_nproj rows (ldn) with g_q at column 0 and g_k
 * at column F, g_x, g_v [N,MF], g_X_out = g_X1 + (tensor-gate path), g_rl [E,D] and g_cut [E] slices.
 * X_in == NULL (lmax <= 4, SiLU): X_in identically zero (first interaction); one g_cut slice is written (not one per degree group).  The tensor-gate blocks of eproj / x / v
 * are not read, the tensor-gate columns of g_eproj are NOT WRITTEN (take the K-prefix (2 + ND) F of the W_e^T product
 * that consumes it), those blocks of g_x / g_v are written as zeros, g_X_out is not written (may be NULL). */
int gn_message_backward(const float* x, const float* v, int ldxv, const float* eproj, int lde, const float* a,
                        const float* qk, int ldqk, const float* X_in, const float* rl, const float* cut,
                        const int* outdeg, const float* g_h1, const float* g_X1,
                        const int* rowptr, const int* src, const int* tgt_by_src, const int* colptr, const int* perm,
                        float* g_eproj, float* g_s, float* g_nproj, int ldn, float* g_x, float* g_v,
                        float* g_X_out, float* g_rl, float* g_cut, float* ga_parts, long E,
                        int N, int F, int H, int lmax, int sep_dir, int sep_tensor, int act, void* stream);
/* Number of degree groups G the message backward uses for these flags (1 = monolithic kernels, lmax >= 5, and every
 * activation other than GN_ACT_SILU -- those run the degree-sliced kernels; lmax 3..4 with sep_dir and sep_tensor
 * and SiLU: {scalar,1,2}, {3}, {4}).  g_cut must then hold G consecutive [E] slices and ga_parts
 * G x [E,H] floats of workspace.  With G = 1, ga_parts ([E,H] floats) selects the single-read form of the register-tiled
 * kernels (the by-source kernel does the per-edge work, then attention backward, then g_k: t_filter is read once);
 * ga_parts = NULL keeps the by-target / by-source pair (t_filter read by both).  The same holds for X_in == NULL (the first
 * interaction) at every lmax <= 4: with ga_parts ([E,H] floats) the single-read kernel runs in its form without the tensor-gate
 * blocks, without it the zero-X_in forms of the pair.  Same gradients either way (summation order aside).
 * GN_LMAX_MAX in `lmax` (the reference's aggr = "max", gotennet.py:638-639): ga_parts is instead a workspace of
 * E x (1 + D) x F floats -- the upstream gradient of every per-edge MESSAGE, routed to the arg-max edge(s) of each output
 * element (evenly among exact ties, as torch's amax) before the degree-sliced backward kernels run; G = 1, X_in required. */
int gn_message_backward_groups(int lmax, int sep_dir, int sep_tensor, int act);

/* EQFF (gotennet.py:716-748) backward, node-local halves around the two gamma_m GEMMs:
 * a: g_m = [g_h | sum_m g_X Xp], g_Xp = g_X * m2;   b: g_Xp += g_ctx[:,F:] Xp / n, g_h1 = g_h + g_ctx[:, :F].
 * g_X == NULL in (a): dL/dX of the block's output is identically zero (an energy head that reads h only,
 * outputs.py:333-346): g_m = [g_h | 0], g_Xp = 0, nothing is read in its place. */
int gn_eqff_backward_a(const float* g_h, const float* g_X, const float* m, const float* Xp,
                       int N, int F, int D, float* g_m, float* g_Xp, void* stream);
int gn_eqff_backward_b(const float* g_ctx, const float* ctx, const float* Xp, const float* g_h,
                       int N, int F, int D, float* g_Xp, float* g_h1, void* stream);

/* EdgeInit (layers.py:1709-1710) backward: g_feat[e, F:2F] = g_t0 (h_i + h_j); g_h[n] += in- and out-edge sums
 * of g_t0 * feat[e, F:2F].  feat/g_feat are the [E, 2F] radial projections (W_ndp | W_erp). */
int gn_edge_init_backward(const float* g_t0, const float* h, const float* feat, int ldf,
                          const int* rowptr, const int* src, const int* colptr, const int* perm,
                          int N, int F, float* g_feat, float* g_h, void* stream);
/* NodeInit aggregate (layers.py:1666-1675) backward: g_feat[e, 0:F] and the g_cut [E] slice from g_ctx[:, F:2F]. */
int gn_node_init_backward(const float* g_ctx, const int* z, const float* feat, int ldf, const float* cut,
                          const float* A_nbr, const int* rowptr, const int* src, int N, int F,
                          float* g_feat, float* g_cut, void* stream);
int gn_layernorm_silu_backward(const float* x, const float* gamma, const float* beta, float eps,
                               const float* g_out, int N, int F, float* g_x, int act, void* stream);
/* input-gradients of gn_layernorm / gn_tensor_norm (x / X are the un-normalised inputs).  torch.max / torch.min
 * route their gradient to one channel: the first extremal one. */
int gn_layernorm_backward(const float* x, const float* gamma, float eps,
                          const float* g_out, int N, int F, float* g_x, void* stream);
int gn_tensor_norm_backward(const float* X, const float* weight, const float* g_Y, float eps, int N, int F,
                            int lmax, float* g_X, void* stream);

/* ---- element-wise pieces of the composed edge update (gotennet.py:236-291: gamma_t as a 2-layer MLP "mlp"/"mlpa",
 *      gamma_w = [LayerNorm "ln"] [act "linwa"] W_edp "linw" [LayerNorm "postln"] [gate]); n = element count, n % 4 = 0.
 *      kind: 0 identity, 1 sigmoid, 2 tanh, 3 SiLU. */
int gn_gate(const float* x, int kind, long n, float* y, void* stream);
int gn_gate_backward(const float* g, const float* x, int kind, long n, float* g_x, void* stream);
/* t' = t + act(pre) * wg  ->  g_pre = g act'(pre) wg,  g_wg = g act(pre);  act in {0, 3}. */
int gn_edge_gate_backward(const float* g, const float* pre, int act, const float* wg, long n,
                          float* g_pre, float* g_wg, void* stream);

/* Edge geometry (K1) backward: (sum of the n_rl slices g_rl [n_rl,E,D], sum of the n_cut slices g_cut [n_cut,E],
 * g_phi [E,R]) -> g_vec [E,3] through the unit vector and harmonics, g_diff [E] through cutoff and radial
 * basis.  Self-loops get zeros. */
int gn_edge_geometry_backward(const float* edge_vec, const float* edge_diff, const int* src, const int* dst,
                              int E, int lmax, int R, int basis, const float* means, const float* betas, float cutoff,
                              const float* g_rl, int n_rl, const float* g_cut, int n_cut, const float* g_phi,
                              float* g_vec, float* g_diff, void* stream);
/* out[n] = sign * ( sum_{src(e)=n} gv_e - sum_{dst(e)=n} gv_e ), gv = g_vec + g_diff * edge_vec/|edge_vec|
 * (Distance.forward, layers.py:1593-1600: edge_vec = pos[j]-pos[i], edge_weight = |edge_vec|). sign=-1: forces. */
int gn_pos_scatter(const float* g_vec, const float* g_diff, const float* edge_vec,
                   const int* rowptr, const int* colptr, const int* perm, int N, float sign,
                   float* out, void* stream);

/* ---- K10 energy head (Atomwise, outputs.py:323-376; SchnetMLP layers.py:225-273, 2 layers, SiLU) ------- */
/* y_n = scale * (sum_k SiLU(pre1[n,k]) W2[k] + b2) + shift (+ atomref[z_n]); energy[b] = sum_{n in molecule b} y_n + mol_shift.
 * Atomwise (outputs.py:323-376) standardises per atom: scale = stddev, shift = mean, mol_shift = 0.  AtomwiseV3
 * (outputs.py:96-229) scales per atom and adds its mean AFTER the aggregation: scale = stddev, shift = 0, mol_shift = mean.
 * pre1 = W1 h + b1 comes from gn_gemm.  mol_ptr [n_mol+1] int32.  head_grad: g_pre1 = scale W2 SiLU'(pre1). */
int gn_head_energy(const float* pre1, const float* W2, float b2, float scale, float shift, float mol_shift,
                   const float* atomref, const int* z, const int* mol_ptr, int n_mol, int Hd,
                   float* y, float* energy, int mean /* aggregation_mode "mean": energy / atoms of the molecule */,
                   float* atom_scale /* [N] or NULL: d energy / d y_n (1 or 1 / atoms) for gn_head_grad */,
                   int act /* GN_ACT_* of the head MLP (GN_ACT_NONE: no hidden layer) */, void* stream);
int gn_head_grad(const float* pre1, const float* W2, float scale, const float* atom_scale /* [N] or NULL */, int N, int Hd,
                 float* g_pre1, int act, void* stream);

/* ---- vector-representation read-outs of the QM9 task (QM9Task.py:168-187) ---------------------------------------
 * GatedEquivariantBlock (outputs.py:24-93) around two gn_gemm products:
 *   vmix [N*3, ldv] = mix_vectors(vectors): V at column 0, W at column w_off, each n_vout wide;
 *   gn_geb_context:  ctx [N, ldc] = [scalars (n_sin) | ||V||_2 over the 3 components (n_vout) | zeros];
 *   x [N, ldx] = scalar_net(ctx) = [s (n_sout) | gate (n_vout) | padding]            (gn_gemm, twice);
 *   gn_geb_gate:     s_out = sactivation(s) (sact = GN_ACT_* or -1 for none), v_out [N*3, ldo] = gate * W. */
int gn_geb_context(const float* s, int lds, int n_sin, const float* vmix, int ldv, int n_vout, int N,
                   float* ctx, int ldc, void* stream);
int gn_geb_gate(const float* x, int ldx, int n_sout, int n_vout, const float* vmix, int ldv, int w_off, int N,
                int sact, float* s_out, int lds, float* v_out, int ldo, void* stream);
/* Dipole (outputs.py:430-468): y_b = sum_{n in molecule b} (mu_n + pos_n q_n), q_n = scale q_n + shift when
 * standardise; y [n_mol,3], or [n_mol] = |y_b| when magnitude; y_vec [n_mol,3] = sum mu_n (or NULL). */
int gn_dipole_reduce(const float* mu, int ldm, const float* q, int ldq, const float* pos, const int* mol_ptr,
                     int n_mol, float scale, float shift, int standardise, int magnitude, float* y, float* y_vec,
                     void* stream);
/* ElectronicSpatialExtentV2 (outputs.py:516-545): c_b = mass-weighted centroid, y_b = sum_n |pos_n - c_b|^2 x_n;
 * mass [n_mass] indexed by atomic number. */
int gn_ese_reduce(const float* x, const float* pos, const int* z, const float* mass, int n_mass, const int* mol_ptr,
                  int n_mol, float* y, void* stream);

/* ---- adjacent: radius graph (Distance.forward, layers.py:1588-1604) ----------------------- */
/* torch_cluster.radius_graph(pos, r, batch, loop=True, max_num_neighbors) semantics: edges j->i with
 * ||pos_j - pos_i||^2 < cutoff^2 (fp32) inside one molecule (batch sorted, int64), target-major,
 * sources ascending, first max_nbr sources per target.  Pass 1 counts deg[i]; the caller scans it
 * into rowptr[N+1] (int64) and sizes the outputs; pass 2 writes edge_index int64 [2,E],
 * edge_vec = pos[j]-pos[i] [E,3] and edge_diff [E] (norm; 0 on self-loops). */
int gn_radius_count(const float* pos, const int64_t* batch, int N, float cutoff, int max_nbr,
                    int* deg, void* stream);
int gn_radius_fill(const float* pos, const int64_t* batch, int N, float cutoff, int max_nbr,
                   const int64_t* rowptr, int64_t E, int64_t* edge_index, float* edge_vec,
                   float* edge_diff, void* stream);
/* Edge vectors of a FIXED edge list for new positions (static-topology MD steps replayed from a hipGraph; the
 * arithmetic of Distance.forward, layers.py:1593-1600):
 * edge_vec[e] = pos[src[e]] - pos[dst[e]], edge_diff[e] = |edge_vec[e]| (0 on self-loops); the arithmetic of
 * gn_radius_fill, so a replayed step is bit-identical to an eager one on the same edge list. */
int gn_edge_vectors(const float* pos, const int* src, const int* dst, int E, float* edge_vec, float* edge_diff,
                    void* stream);


#ifdef __cplusplus
}
#endif
#endif /* GOTENNET_HIP_H */
