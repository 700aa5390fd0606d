// gn_heads.hip -- the two vector-representation read-outs the reference's QM9 task builds on top of the path
// (models/tasks/QM9Task.py:168-187): Dipole (outputs.py:379-468: two GatedEquivariantBlocks, outputs.py:24-93) and
// ElectronicSpatialExtentV2 (outputs.py:471-545).  Node-local and tiny next to the interaction layers (a few [N,F]
// rows per atom): plain one-thread-per-element kernels, per-molecule sums in a fixed order (one workgroup per molecule,
// wave-level butterflies: no atomics).  The GEMMs of the blocks go through gn_gemm like every other projection.
#include "gn_common.h"

namespace gn {

// GatedEquivariantBlock, first half (outputs.py:78-82):  ctx = [scalars | ||V||_2 over the 3 components | 0-padding]
// vmix [N*3, ldv] holds mix_vectors(vectors): V at column 0, W at column w_off (each n_vout wide).
__global__ void geb_context_kernel(const float* __restrict__ s, int lds, int n_sin, const float* __restrict__ vmix,
                                   int ldv, int n_vout, int N, float* __restrict__ ctx, int ldc) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * ldc) return;
    const int n = (int)(idx / ldc), c = (int)(idx % ldc);
    float v = 0.f;
    if (c < n_sin) {
        v = s[(size_t)n * lds + c];
    } else if (c < n_sin + n_vout) {
        const int f = c - n_sin;
        const float a = vmix[((size_t)n * 3 + 0) * ldv + f], b = vmix[((size_t)n * 3 + 1) * ldv + f],
                    d = vmix[((size_t)n * 3 + 2) * ldv + f];
        v = sqrtf(a * a + b * b + d * d);            // torch.norm(vectors_V, dim=-2)
    }
    ctx[idx] = v;
}

// second half (outputs.py:84-91):  x = scalar_net(ctx) = [s_out | gate];  v_out = gate * W;  s_out = sactivation(s_out)
__global__ void geb_gate_kernel(const float* __restrict__ x, int ldx, int n_sout, int n_vout,
                                const float* __restrict__ vmix, int ldv, int w_off, int N, int sact,
                                float* __restrict__ s_out, int lds, float* __restrict__ v_out, int ldo) {
    const int per = n_sout + 3 * n_vout;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * per) return;
    const int n = (int)(idx / per), c = (int)(idx % per);
    if (c < n_sout) {
        const float v = x[(size_t)n * ldx + c];
        s_out[(size_t)n * lds + c] = sact < 0 ? v : act1(v, sact);
    } else {
        const int m = (c - n_sout) / n_vout, f = (c - n_sout) % n_vout;
        v_out[((size_t)n * 3 + m) * ldo + f] = x[(size_t)n * ldx + n_sout + f] * vmix[((size_t)n * 3 + m) * ldv + w_off + f];
    }
}

// Dipole read-out (outputs.py:449-468), one workgroup per molecule:
//   y_b = sum_n ( mu_n + pos_n * q_n ),  q_n = stddev * s_n + mean (when standardised),  mu_n = v_n[:, 0]
//   y_vec_b = sum_n mu_n;  magnitude: |y_b|
__global__ __launch_bounds__(64) void dipole_reduce_kernel(
    const float* __restrict__ mu, int ldm, const float* __restrict__ q, int ldq, const float* __restrict__ pos,
    const int* __restrict__ mol_ptr, float scale, float shift, int standardise, int magnitude,
    float* __restrict__ y, float* __restrict__ y_vec) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
    float d[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
    for (int n = n0 + lane; n < n1; n += 64) {
        float c = q[(size_t)n * ldq];
        if (standardise) c = scale * c + shift;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float a = mu[((size_t)n * 3 + m) * ldm];
            v[m] += a;
            d[m] += a + pos[(size_t)n * 3 + m] * c;
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) { d[m] = wave_sum(d[m]); v[m] = wave_sum(v[m]); }
    if (lane == 0) {
        if (magnitude) y[b] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        else { y[3 * b] = d[0]; y[3 * b + 1] = d[1]; y[3 * b + 2] = d[2]; }
        if (y_vec) { y_vec[3 * b] = v[0]; y_vec[3 * b + 1] = v[1]; y_vec[3 * b + 2] = v[2]; }
    }
}

// ElectronicSpatialExtentV2 read-out (outputs.py:526-545), one workgroup per molecule:
//   c_b = sum_n m_n pos_n / sum_n m_n,   y_b = sum_n |pos_n - c_b|^2 x_n
__global__ __launch_bounds__(64) void ese_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ pos, const int* __restrict__ z,
    const float* __restrict__ mass, int n_mass, const int* __restrict__ mol_ptr, float* __restrict__ y) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n0 = mol_ptr[b], n1 = mol_ptr[b + 1];
    float mp[3] = {0.f, 0.f, 0.f}, ms = 0.f;
    for (int n = n0 + lane; n < n1; n += 64) {
        const int zn = z[n];
        const float m = (zn >= 0 && zn < n_mass) ? mass[zn] : 0.f;
        ms += m;
#pragma unroll
        for (int k = 0; k < 3; ++k) mp[k] += m * pos[(size_t)n * 3 + k];
    }
    ms = wave_sum(ms);
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = ms > 0.f ? wave_sum(mp[k]) / ms : 0.f;      // (massless molecule: rejected by the host; no 0 / 0 here)
    float s = 0.f;
    for (int n = n0 + lane; n < n1; n += 64) {
        const float dx = pos[(size_t)n * 3] - c[0], dy = pos[(size_t)n * 3 + 1] - c[1], dz = pos[(size_t)n * 3 + 2] - c[2];
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);      // torch.norm(...) ** 2, as the reference writes it
        s += (r * r) * x[n];
    }
    s = wave_sum(s);
    if (lane == 0) y[b] = s;
}

}  // namespace gn

// ====================================================================================== C ABI
extern "C" int gn_geb_context(const float* s, int lds, int n_sin, const float* vmix, int ldv, int n_vout, int N,
                              float* ctx, int ldc, void* stream) {
    if (N < 0 || n_sin <= 0 || n_vout <= 0 || ldc < n_sin + n_vout || lds < n_sin || ldv < n_vout) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * ldc;
    hipLaunchKernelGGL(gn::geb_context_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s, lds, n_sin, vmix, ldv, n_vout, N, ctx, ldc);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_geb_gate(const float* x, int ldx, int n_sout, int n_vout, const float* vmix, int ldv, int w_off, int N,
                           int sact, float* s_out, int lds, float* v_out, int ldo, void* stream) {
    if (N < 0 || n_sout <= 0 || n_vout <= 0 || ldx < n_sout + n_vout || lds < n_sout || ldo < n_vout ||
        ldv < w_off + n_vout || sact < -1 || sact >= GN_ACT_COUNT)
        return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    const size_t tot = (size_t)N * (n_sout + 3 * n_vout);
    hipLaunchKernelGGL(gn::geb_gate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, n_sout, n_vout, vmix, ldv, w_off, N, sact, s_out, lds, v_out, ldo);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_dipole_reduce(const float* mu, int ldm, const float* q, int ldq, const float* pos, const int* mol_ptr,
                                int n_mol, float scale, float shift, int standardise, int magnitude, float* y,
                                float* y_vec, void* stream) {
    if (n_mol < 0 || ldm <= 0 || ldq <= 0 || !y) return GN_ERR_BAD_ARG;
    if (n_mol == 0) return GN_OK;
    hipLaunchKernelGGL(gn::dipole_reduce_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream,
                       mu, ldm, q, ldq, pos, mol_ptr, scale, shift, standardise, magnitude, y, y_vec);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_ese_reduce(const float* x, const float* pos, const int* z, const float* mass, int n_mass,
                             const int* mol_ptr, int n_mol, float* y, void* stream) {
    if (n_mol < 0 || n_mass <= 0 || !y) return GN_ERR_BAD_ARG;
    if (n_mol == 0) return GN_OK;
    hipLaunchKernelGGL(gn::ese_reduce_kernel, dim3(n_mol), dim3(64), 0, (hipStream_t)stream,
                       x, pos, z, mass, n_mass, mol_ptr, y);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
