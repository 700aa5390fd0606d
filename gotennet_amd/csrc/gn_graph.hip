// gn_graph.hip -- radius graph + edge vectors (reference Distance.forward, layers.py:1588-1604,
// over torch_cluster.radius_graph(pos, r, batch, loop=True, max_num_neighbors)).
//
// Molecules are small (tens to hundreds of atoms) and contiguous in `batch`, so each
// target atom scans its own molecule: O(atoms-per-molecule) per thread, no cell list.
// Output layout = radius_graph's: target-major, sources ascending, strict d^2 < r^2
// in fp32 (x,y,z accumulated in that order), at most max_nbr sources per target
// (the first max_nbr in source order), self-loop included.
#include "gn_common.h"

namespace gn {

__device__ __forceinline__ bool within(const float* __restrict__ pos, int i, int j, float r2) {
    const float dx = pos[3 * i] - pos[3 * j];
    const float dy = pos[3 * i + 1] - pos[3 * j + 1];
    const float dz = pos[3 * i + 2] - pos[3 * j + 2];
    float d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d < r2;
}

__device__ __forceinline__ int molecule_start(const int64_t* __restrict__ batch, int i) {
    const int64_t b = batch[i];
    int s = i;
    while (s > 0 && batch[s - 1] == b) --s;
    return s;
}

__global__ void radius_count_kernel(const float* __restrict__ pos, const int64_t* __restrict__ batch, int N,
                                    float r2, int max_nbr, int* __restrict__ deg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t b = batch[i];
    int c = 0;
    for (int j = molecule_start(batch, i); j < N && batch[j] == b && c < max_nbr; ++j) c += within(pos, i, j, r2) ? 1 : 0;
    deg[i] = c;
}

// rowptr = exclusive scan of deg (length N+1)
__global__ void radius_fill_kernel(const float* __restrict__ pos, const int64_t* __restrict__ batch, int N,
                                   float r2, int max_nbr, const int64_t* __restrict__ rowptr, int64_t E,
                                   int64_t* __restrict__ edge_index, float* __restrict__ edge_vec,
                                   float* __restrict__ edge_diff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t b = batch[i];
    int64_t e = rowptr[i];
    int c = 0;
    for (int j = molecule_start(batch, i); j < N && batch[j] == b && c < max_nbr; ++j) {
        if (!within(pos, i, j, r2)) continue;
        edge_index[e] = j;           // row 0: source
        edge_index[E + e] = i;       // row 1: target
        const float vx = pos[3 * j] - pos[3 * i], vy = pos[3 * j + 1] - pos[3 * i + 1], vz = pos[3 * j + 2] - pos[3 * i + 2];
        edge_vec[3 * e] = vx; edge_vec[3 * e + 1] = vy; edge_vec[3 * e + 2] = vz;
        edge_diff[e] = (j == i) ? 0.0f : sqrtf(vx * vx + vy * vy + vz * vz);
        ++e; ++c;
    }
}

// edge vectors of a FIXED edge list for new positions (same arithmetic as radius_fill_kernel)
__global__ void edge_vectors_kernel(const float* __restrict__ pos, const int* __restrict__ src, const int* __restrict__ dst,
                                    int E, float* __restrict__ edge_vec, float* __restrict__ edge_diff) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int j = src[e], i = dst[e];
    const float vx = pos[3 * j] - pos[3 * i], vy = pos[3 * j + 1] - pos[3 * i + 1], vz = pos[3 * j + 2] - pos[3 * i + 2];
    edge_vec[3 * e] = vx; edge_vec[3 * e + 1] = vy; edge_vec[3 * e + 2] = vz;
    edge_diff[e] = (j == i) ? 0.0f : sqrtf(vx * vx + vy * vy + vz * vz);
}

// ---- by-source (CSC) view of a target-major edge list, stable: perm lists the CSR edge ids of every source in increasing
// order (= what torch.sort(src, stable=True) gives), without a sort: count -> scan -> scatter (integer atomics: the ORDER
// inside a bucket is arbitrary at this point, its CONTENT is not) -> rank every bucket entry among its bucket's ids.
__global__ void csc_count_kernel(const int* __restrict__ src, int E, int* __restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(cnt + src[e], 1);
}
// one workgroup: colptr = exclusive scan of cnt (N + 1 entries), cursor = a copy of colptr[0..N) for the scatter.
// `cursor` MAY alias `cnt` (the launcher scans in place: a thread reads cnt[n] before it writes cursor[n], and no other
// thread touches entry n) -- neither pointer is __restrict__
__global__ __launch_bounds__(1024) void csc_scan_kernel(const int* cnt, int N, int* __restrict__ colptr, int* cursor) {
    __shared__ int part[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < N; base += 1024) {
        const int n = base + t;
        const int v = n < N ? cnt[n] : 0;
        int incl = v;                                            // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += part[w];
        if (n < N) {
            colptr[n] = before + incl - v;
            cursor[n] = before + incl - v;
        }
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
    if (t == 0) colptr[N] = carry;
}
__global__ void csc_scatter_kernel(const int* __restrict__ src, int E, int* __restrict__ cursor, int* __restrict__ tmp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) tmp[atomicAdd(cursor + src[e], 1)] = e;
}
// one wave per source: entry j of the bucket goes to position #{k : id_k < id_j} (ids are distinct) = the order a stable
// sort by source gives.  O(d^2 / 64) comparisons per bucket: short buckets (a molecular neighbour list: d <= 64) compare against
// uniform global loads; longer ones (hub atoms of an uncapped dense graph) stage the bucket through a per-wave LDS
// window of 1024 ids, one coalesced pass per window, and count against LDS broadcasts -- the same ranks, ~10x fewer cycles at
// d = 4096
__global__ __launch_bounds__(256) void csc_rank_kernel(const int* __restrict__ colptr, const int* __restrict__ tmp,
                                                       const int* __restrict__ dst, int N, int* __restrict__ perm,
                                                       int* __restrict__ tgt_by_src) {
    constexpr int WIN = 1024;
    __shared__ int win[4][WIN];
    const int wv = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (s >= N) return;
    const int p0 = colptr[s], d = colptr[s + 1] - p0;
    if (d <= 64) {
        if (lane < d) {
            const int id = tmp[p0 + lane];
            int r = 0;
            for (int k = 0; k < d; ++k) r += tmp[p0 + k] < id ? 1 : 0;
            perm[p0 + r] = id;
            tgt_by_src[p0 + r] = dst[id];
        }
        return;
    }
    for (int j0 = 0; j0 < d; j0 += 64) {             // (every lane runs every trip: the window loads are wave-wide)
        const int j = j0 + lane;
        const int id = j < d ? tmp[p0 + j] : 0x7fffffff;
        int r = 0;
        for (int w0 = 0; w0 < d; w0 += WIN) {
            const int wn = d - w0 < WIN ? d - w0 : WIN;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < wn; k += 64) win[wv][k] = tmp[p0 + w0 + k];
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < wn; ++k) r += win[wv][k] < id ? 1 : 0;
        }
        if (j < d) {
            perm[p0 + r] = id;
            tgt_by_src[p0 + r] = dst[id];
        }
    }
}

// offsets of the molecules in a SORTED batch vector: mol_ptr[m] = first atom with batch >= m  (m = 0 .. n_mol).
// Precondition: batch non-decreasing (molecules contiguous, what a PyG Batch gives).  For ANY input every entry 0 .. n_mol is
// written with a value in [0, N] (the sequence -1, batch[0], .., batch[N-1], n_mol crosses every m at least once; negative and
// too-large entries are clamped), so a bad batch vector gives wrong molecule sums, never an out-of-bounds access
__global__ void molecule_ptr_kernel(const int64_t* __restrict__ batch, int N, int n_mol, int* __restrict__ mol_ptr) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    int64_t lo = n == 0 ? 0 : batch[n - 1] + 1;                  // molecules lo .. hi start at atom n
    if (lo < 0) lo = 0;
    const int64_t hi = n == N ? n_mol : batch[n];
    for (int64_t m = lo; m <= hi && m <= n_mol; ++m) mol_ptr[m] = n;
}

}  // namespace gn

extern "C" int gn_build_csc(const int* src, const int* dst, int E, int N, int* colptr, int* perm, int* tgt_by_src,
                            int* work, void* stream) {
    if (E < 0 || N < 0 || !colptr || (E > 0 && (!src || !dst || !perm || !tgt_by_src || !work))) return GN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* cnt = work;                                             // [N] counts, then the scatter cursor
    int* tmp = work + N;                                         // [E] bucket contents before ranking
    if (N > 0 && hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)N, st) != hipSuccess) return (int)hipGetLastError();
    if (E > 0) hipLaunchKernelGGL(gn::csc_count_kernel, dim3((E + 255) / 256), dim3(256), 0, st, src, E, cnt);
    hipLaunchKernelGGL(gn::csc_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, N, colptr, cnt);
    if (E > 0) {
        hipLaunchKernelGGL(gn::csc_scatter_kernel, dim3((E + 255) / 256), dim3(256), 0, st, src, E, cnt, tmp);
        hipLaunchKernelGGL(gn::csc_rank_kernel, dim3((N + 3) / 4), dim3(256), 0, st, colptr, tmp, dst, N, perm, tgt_by_src);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_molecule_ptr(const int64_t* batch, int N, int n_mol, int* mol_ptr, void* stream) {
    if (N < 0 || n_mol < 0 || !mol_ptr || (N > 0 && !batch)) return GN_ERR_BAD_ARG;
    hipLaunchKernelGGL(gn::molecule_ptr_kernel, dim3((N + 256) / 256), dim3(256), 0, (hipStream_t)stream, batch, N, n_mol, mol_ptr);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_edge_vectors(const float* pos, const int* src, const int* dst, int E, float* edge_vec,
                               float* edge_diff, void* stream) {
    if (E < 0) return GN_ERR_BAD_ARG;
    if (E == 0) return GN_OK;
    hipLaunchKernelGGL(gn::edge_vectors_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       pos, src, dst, E, edge_vec, edge_diff);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_radius_count(const float* pos, const int64_t* batch, int N, float cutoff, int max_nbr,
                               int* deg, void* stream) {
    if (N < 0 || max_nbr <= 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::radius_count_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       pos, batch, N, cutoff * cutoff, max_nbr, deg);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" int gn_radius_fill(const float* pos, const int64_t* batch, int N, float cutoff, int max_nbr,
                              const int64_t* rowptr, int64_t E, int64_t* edge_index, float* edge_vec,
                              float* edge_diff, void* stream) {
    if (N < 0 || max_nbr <= 0 || E < 0) return GN_ERR_BAD_ARG;
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(gn::radius_fill_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       pos, batch, N, cutoff * cutoff, max_nbr, rowptr, E, edge_index, edge_vec, edge_diff);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
