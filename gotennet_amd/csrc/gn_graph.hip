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

}  // namespace gn

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
