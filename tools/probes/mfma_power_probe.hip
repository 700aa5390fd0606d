// mfma_power_probe.hip -- does the int8 (or fp8) matrix pipe sit further below the socket power cap than the fp16 pipe?
// Register-resident MFMA streams (no memory traffic inside the loop), every CU busy at 2 workgroups x 4 waves, operands
// random / ones / zeros.  Prints the time of the same number of instructions per pipe and the rate it implies.
// the 2 x fp16 split executes 3 fp16 terms per product; an int8 slicing needs 6 terms at twice the nominal
// rate -- it only wins if the int8 stream runs closer to ITS nameplate than the fp16 stream does to its own.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power_probe tools/probes/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int NT = 8;      // independent accumulator tiles per wave (back-to-back issue without dependency stalls)

__global__ __launch_bounds__(256, 2) void f16_stream(const v8h* __restrict__ src, float* __restrict__ out, int iters) {
    v8h a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[threadIdx.x * 4 + i]; b[i] = src[threadIdx.x * 4 + 2 + i]; }
    v16f c[NT];
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 1], b[(t >> 1) & 1], c[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void i8_stream(const v4i* __restrict__ src, int* __restrict__ out, int iters) {
    v4i a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[threadIdx.x * 4 + i]; b[i] = src[threadIdx.x * 4 + 2 + i]; }
    v16i c[NT];
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) c[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 1], b[(t >> 1) & 1], c[t], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    if (s == 123456789) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void f8_stream(const v8i* __restrict__ src, float* __restrict__ out, int iters) {
    v8i a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[threadIdx.x * 4 + i]; b[i] = src[threadIdx.x * 4 + 2 + i]; }
    v16f c[NT];
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t & 1], b[(t >> 1) & 1], c[t], 0, 0, 0, 127, 0, 127);
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, grid = 512, reps = 20;
    const size_t bytes = 256 * 4 * 32;
    std::vector<unsigned char> h(bytes);
    void *d, *o;
    hipMalloc(&d, bytes); hipMalloc(&o, grid * 256 * 4);
    const double n_inst = (double)grid * 4 * iters * NT;                 // wave-level MFMA instructions per launch
    const char* fill_name[3] = {"random", "ones", "zeros"};
    for (int fill = 0; fill < 3; ++fill) {
        // fp16: random = normal-ish fp16 values in [-2, 2); int8 / fp8: random bytes
        srand(1);
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 0) {                                             // fp16 payload
                _Float16* p = (_Float16*)h.data();
                for (size_t i = 0; i < bytes / 2; ++i)
                    p[i] = fill == 0 ? (_Float16)((rand() / (float)RAND_MAX) * 4.f - 2.f) : (fill == 1 ? (_Float16)1.f : (_Float16)0.f);
            } else if (pass == 1) {                                      // int8 payload
                for (size_t i = 0; i < bytes; ++i) h[i] = fill == 0 ? (unsigned char)(rand() & 0xff) : (fill == 1 ? 1 : 0);
            } else {                                                     // fp8 e4m3 payload (no NaN patterns: clear the top exponent/mantissa combination)
                for (size_t i = 0; i < bytes; ++i) {
                    unsigned char b = fill == 0 ? (unsigned char)(rand() & 0xff) : (fill == 1 ? 0x38 : 0);
                    if ((b & 0x7f) == 0x7f) b &= 0xfe;
                    h[i] = b;
                }
            }
            hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
            float ms;
            double flop_per_inst;
            const char* name;
            if (pass == 0) { ms = time_ms([&] { hipLaunchKernelGGL(f16_stream, dim3(grid), dim3(256), 0, 0, (const v8h*)d, (float*)o, iters); }, reps); flop_per_inst = 2.0 * 32 * 32 * 16; name = "fp16 32x32x16"; }
            else if (pass == 1) { ms = time_ms([&] { hipLaunchKernelGGL(i8_stream, dim3(grid), dim3(256), 0, 0, (const v4i*)d, (int*)o, iters); }, reps); flop_per_inst = 2.0 * 32 * 32 * 32; name = "int8 32x32x32"; }
            else { ms = time_ms([&] { hipLaunchKernelGGL(f8_stream, dim3(grid), dim3(256), 0, 0, (const v8i*)d, (float*)o, iters); }, reps); flop_per_inst = 2.0 * 32 * 32 * 64; name = "fp8  32x32x64"; }
            printf("%-7s %s: %8.1f us per launch, %7.1f T(FL)OP/s, %6.2f cycles per wave-instruction at 2.4 GHz nameplate\n",
                   fill_name[fill], name, ms * 1e3, n_inst * flop_per_inst / (ms * 1e-3) / 1e12,
                   (ms * 1e-3) * 2.4e9 / ((double)iters * NT * 2));      // 2 waves per SIMD share the pipe
        }
    }
    return 0;
}
