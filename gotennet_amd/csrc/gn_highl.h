// gn_highl.h -- pieces shared by the tuned kernels (gn_gata.hip, gn_backward.hip, gn_options.hip) and the
// degree-sliced kernels that serve lmax = 5..8 (gn_highl.hip).
#pragma once
#include "gn_common.h"

namespace gn {

// cross-slot fixed-order reduction of ROWS float4 accumulators; `wr(row, sum)` is called by
// exactly one slot per row.  red: >= min(ROWS, 9) * 1024 floats of LDS.
template <int ROWS, typename Writer>
__device__ __forceinline__ void reduce_rows(float4 (&acc)[ROWS], float* red, int slot, int c0, int F, int ns, Writer wr) {
    constexpr int CH = ROWS < 9 ? ROWS : 9;
#pragma unroll
    for (int base = 0; base < ROWS; base += CH) {
        if (base) __syncthreads();
#pragma unroll
        for (int r = 0; r < CH; ++r)
            if (base + r < ROWS) st4(&red[r * 1024 + slot * F + c0], acc[base + r]);
        __syncthreads();
        for (int r = slot; r < CH && base + r < ROWS; r += ns) wr(base + r, red4(red + r * 1024, c0, F, ns));
    }
}

struct MsgBwdArgs {
    // saved forward tensors
    const float* x; const float* v; int ldxv;          // [N, M F]
    const float* eproj; int lde;                       // [E, (1+M) F]: pre_ta | t_filter
    const float* a;                                    // [E, H] attention weights (softmax * norm)
    const float* qk; int ldqk;                         // q at col 0, k at col F
    const float* X_in;                                 // [N, D, F] layer input X
    const float* rl; const float* cut;
    const int* outdeg;                                 // scale_edge (or NULL)
    // upstream gradients
    const float* g_h1; const float* g_X1;              // [N,F], [N,D,F]
    // graph
    const int* rowptr; const int* src; const int* dst;   // dst: target of the pp-th BY-SOURCE entry (= CSR dst[perm[pp]])
    const int* colptr; const int* perm;
    // outputs
    float* g_eproj;                                    // [E, (1+M) F]: d/d(W_re t + b) | d/d t_filter
    float* g_s;                                        // [E, H] scratch: g_a then g_s
    float* g_nproj; int ldn;                           // [N, 4F]: g_q at col 0, g_k at col F
    float* g_x; float* g_v;                            // [N, M F]
    float* g_X_out;                                    // [N, D, F] = g_X1 + source part
    float* g_rl; float* g_cut;                         // this call's slice (written, not accumulated)
    int N, F, H;
    float inv_sqrt_f;
    int act;                                           // GN_ACT_*: t_attn = act(W_re t + b)
    int mean;                                          // aggr = "mean" (degree-sliced kernels only): messages scaled by 1 / in-degree
    const float* g_edge;                               // aggr = "max" (degree-sliced kernels only): [E, 1 + D, F] upstream gradient of
                                                       // every MESSAGE (row 0: scalar, rows 1..D: tensor), routed to the arg-max
                                                       // edges by hl_max_route_kernel; NULL: the per-target rows g_h1 / g_X1
};

// gamma_w (gotennet.py:285-291): 0 identity, 1 nn.Sigmoid ("gated"), 2 nn.Tanh ("gatedt"), 3 nn.SiLU ("act")
__device__ __forceinline__ float gate1(float x, int kind) {
    switch (kind) {
        case 1: return 1.0f / (1.0f + expf(-x));
        case 2: return tanhf(x);
        case 3: return silu(x);
        default: return x;
    }
}
__device__ __forceinline__ float dgate1(float x, int kind) {
    switch (kind) {
        case 1: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f - s); }
        case 2: { const float t = tanhf(x); return 1.0f - t * t; }
        case 3: return dsilu(x);
        default: return 1.0f;
    }
}
__device__ __forceinline__ float4 gate4(float4 v, int k) { return make_float4(gate1(v.x, k), gate1(v.y, k), gate1(v.z, k), gate1(v.w, k)); }
__device__ __forceinline__ float4 dgate4(float4 v, int k) { return make_float4(dgate1(v.x, k), dgate1(v.y, k), dgate1(v.z, k), dgate1(v.w, k)); }

}  // namespace gn

// lmax > 4, or the caller OR-ed GN_LMAX_SLICED into the lmax ARGUMENT of the entry point (an explicit per-call request
// for the degree-sliced kernel family at lmax <= 4: the tests hold the two families against each other).  No
// process-wide switch: the choice travels with the call.
bool gn_use_highl(int lmax_arg);
int gn_highl_message(const float* x, const float* v, int ldxv, const float* t_filter, int ldt, const float* a,
                     const float* rl, const float* cut, const int* rowptr, const int* src, const float* h_in,
                     const float* X_in, float* h_out, float* X_out, int N, int F, int H, int lmax, int sep_dir,
                     int sep_tensor, int aggr /* 0 add, 1 mean, 2 max */, hipStream_t st);
int gn_highl_message_backward(const gn::MsgBwdArgs& p, int lmax, int sep_dir, int sep_tensor, hipStream_t st);
int gn_highl_htr_edge(const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                      int N, int F, int lmax, int mode, float* w_raw, float* w, hipStream_t st);
int gn_highl_htr_backward(const float* g_t_out, const float* pre_t, const float* w, const float* w_raw,
                          const float* EQ, const float* EK, const float* rl, const int* rowptr, const int* src,
                          const int* dst, const int* colptr, const int* perm, int N, int F, int lmax, int mode,
                          float* gEQ, float* gEK, float* g_rl, float* g_pre_t, int act, hipStream_t st);
