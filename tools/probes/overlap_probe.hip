// overlap_probe.hip -- can an fp16 MFMA stream and an HBM stream share the chip without adding their times?
// (1) two kernels on two HIP streams: register-resident MFMA stream (random operands: the power-capped case) and a float4 copy;
// (2) ONE kernel whose workgroups hold 4 MFMA waves + 4 copy waves (what a wave-specialised projection kernel would be).
// Prints each alone, both together, and the sum / max they would take if the times added / overlapped perfectly.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/overlap_probe tools/probes/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int NT = 8;

__device__ __forceinline__ void mfma_body(const v8h* __restrict__ src, float* __restrict__ out, int iters, int lane_id) {
    v8h a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[lane_id * 4 + i]; b[i] = src[lane_id * 4 + 2 + i]; }
    v16f c[NT];
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 1], b[(t >> 1) & 1], c[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    if (s == 12345.678f) out[lane_id] = s;
}
__device__ __forceinline__ void copy_body(const float4* __restrict__ in, float4* __restrict__ o, size_t n4, size_t start, size_t stride) {
    for (size_t i = start; i < n4; i += stride * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; v[u] = j < n4 ? in[j] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; if (j < n4) o[j] = v[u]; }
    }
}
__global__ __launch_bounds__(256, 2) void k_mfma(const v8h* src, float* out, int iters) { mfma_body(src, out, iters, threadIdx.x); }
__global__ __launch_bounds__(256) void k_copy(const float4* in, float4* o, size_t n4) {
    copy_body(in, o, n4, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
// waves 0-3: MFMA stream; waves 4-7: copy.  PRIO: the copy waves raise their issue priority (s_setprio 3)
template <int PRIO>
__global__ __launch_bounds__(512, 2) void k_both(const v8h* src, float* out, int iters, const float4* in, float4* o, size_t n4) {
    if (threadIdx.x < 256) mfma_body(src, out, iters, threadIdx.x);
    else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        copy_body(in, o, n4, (size_t)blockIdx.x * 256 + (threadIdx.x - 256), (size_t)gridDim.x * 256);
    }
}
// the copy waves FIRST in the workgroup (waves 0-3), the MFMA waves after them
__global__ __launch_bounds__(512, 2) void k_both_rev(const v8h* src, float* out, int iters, const float4* in, float4* o, size_t n4) {
    if (threadIdx.x >= 256) mfma_body(src, out, iters, threadIdx.x - 256);
    else copy_body(in, o, n4, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 600;
    const size_t mb = argc > 2 ? atoi(argv[2]) : 512;
    const int fill = argc > 3 ? atoi(argv[3]) : 0;           // 0 random operands (power-capped MFMA stream), 1 zeros (nameplate rate)
    const size_t n4 = mb * 1024 * 1024 / 16;
    const size_t bytes = 256 * 4 * 16;
    std::vector<_Float16> h(bytes / 2);
    srand(1);
    for (auto& x : h) x = fill ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX) * 4.f - 2.f);
    void *d, *o; float4 *ci, *co;
    hipMalloc(&d, bytes); hipMalloc(&o, 512 * 4); hipMalloc(&ci, n4 * 16); hipMalloc(&co, n4 * 16);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemset(ci, 1, n4 * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto L_m = [&](hipStream_t s) { hipLaunchKernelGGL(k_mfma, dim3(512), dim3(256), 0, s, (const v8h*)d, (float*)o, iters); };
    auto L_c = [&](hipStream_t s) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, ci, co, n4); };
    auto L_b = [&](hipStream_t s) { hipLaunchKernelGGL(k_both<0>, dim3(512), dim3(512), 0, s, (const v8h*)d, (float*)o, iters, ci, co, n4); };
    auto L_p = [&](hipStream_t s) { hipLaunchKernelGGL(k_both<1>, dim3(512), dim3(512), 0, s, (const v8h*)d, (float*)o, iters, ci, co, n4); };
    auto L_r = [&](hipStream_t s) { hipLaunchKernelGGL(k_both_rev, dim3(512), dim3(512), 0, s, (const v8h*)d, (float*)o, iters, ci, co, n4); };
    auto timed = [&](auto f, int reps) {
        f(); hipDeviceSynchronize();
        double best = 1e30;
        for (int r = 0; r < reps; ++r) { const double t0 = now_us(); f(); hipDeviceSynchronize(); best = std::min(best, now_us() - t0); }
        return best;
    };
    for (int round = 0; round < 2; ++round) {
        const double tm = timed([&] { for (int k = 0; k < 4; ++k) L_m(s1); }, 6) / 4;
        const double tc = timed([&] { for (int k = 0; k < 4; ++k) L_c(s2); }, 6) / 4;
        const double tb = timed([&] { for (int k = 0; k < 4; ++k) { L_m(s1); L_c(s2); } }, 6) / 4;
        const double tw = timed([&] { for (int k = 0; k < 4; ++k) L_b(s1); }, 6) / 4;
        const double tp = timed([&] { for (int k = 0; k < 4; ++k) L_p(s1); }, 6) / 4;
        const double tr = timed([&] { for (int k = 0; k < 4; ++k) L_r(s1); }, 6) / 4;
        printf("%s operands, iters %d, copy %zu MiB:  MFMA alone %7.1f us | copy alone %7.1f us (%.2f TB/s) | two streams %7.1f us | one kernel, 4 + 4 waves %7.1f us"
               " (copy waves at s_setprio 3: %7.1f; copy waves first: %7.1f) | sum %7.1f, max %7.1f\n", fill ? "zero  " : "random", iters, mb, tm, tc, 2.0 * n4 * 16 / tc / 1e6, tb, tw, tp, tr, tm + tc, tm > tc ? tm : tc);
    }
    return 0;
}
