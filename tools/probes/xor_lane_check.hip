// xor_lane_check.hip -- the VALU cross-lane exchanges of gn_common.h against __shfl_xor, bit for bit (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gotennet_amd/csrc -o tools/probes/xor_lane_check tools/probes/xor_lane_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "gn_common.h"

template <int K>
__device__ void mgs_ref(float (&v)[K], int width, int lp) {        // the round-4 formulation (ds_bpermute shuffles)
    int off = width >> 1;
    for (int live = K; live >= 2; live >>= 1, off >>= 1) {
        const bool up = (lp & off) != 0;
        for (int i = 0; i < live / 2; ++i) {
            const float a = v[i], b = v[i + live / 2];
            v[i] = (up ? b : a) + __shfl_xor(up ? a : b, off, 64);
        }
    }
    for (; off > 0; off >>= 1) v[0] += __shfl_xor(v[0], off, 64);
}

__global__ void check(const float* in, int* bad) {
    const int t = threadIdx.x;
    const float v = in[t];
    int nb = 0;
    nb += __float_as_int(gn::xor_lane<1>(v)) != __float_as_int(__shfl_xor(v, 1, 64));
    nb += __float_as_int(gn::xor_lane<2>(v)) != __float_as_int(__shfl_xor(v, 2, 64));
    nb += __float_as_int(gn::xor_lane<4>(v)) != __float_as_int(__shfl_xor(v, 4, 64));
    nb += __float_as_int(gn::xor_lane<8>(v)) != __float_as_int(__shfl_xor(v, 8, 64));
    nb += __float_as_int(gn::xor_lane<16>(v)) != __float_as_int(__shfl_xor(v, 16, 64));
    nb += __float_as_int(gn::xor_lane<32>(v)) != __float_as_int(__shfl_xor(v, 32, 64));
    for (int w = 1; w <= 64; w <<= 1) {
        float r = v;
        for (int o = 1; o < w; o <<= 1) r += __shfl_xor(r, o, 64);
        nb += __float_as_int(gn::group_sum(v, w)) != __float_as_int(r);
    }
    {
        float r = v;
        for (int o = 1; o < 64; o <<= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
        nb += __float_as_int(gn::wave_max(v)) != __float_as_int(r);
    }
    for (int st = 1; st <= 32; st <<= 1) {
        float r = v, m = v;
        for (int o = st; o < 64; o <<= 1) { r += __shfl_xor(r, o, 64); m = fmaxf(m, __shfl_xor(m, o, 64)); }
        nb += __float_as_int(gn::stride_sum(v, st)) != __float_as_int(r);
        nb += __float_as_int(gn::stride_max(v, st)) != __float_as_int(m);
    }
#define MGS(K, W)                                                                         \
    {                                                                                     \
        float a[K], b[K];                                                                 \
        for (int k = 0; k < K; ++k) a[k] = b[k] = in[64 + k * 64 + t] * (1.f + k);        \
        gn::multi_group_sum<K>(a, W, t % W);                                              \
        mgs_ref<K>(b, W, t % W);                                                          \
        if ((t % W) % (W / K) == 0) nb += __float_as_int(a[0]) != __float_as_int(b[0]);   \
    }
    MGS(8, 64) MGS(8, 32) MGS(8, 16) MGS(8, 8) MGS(4, 64) MGS(4, 4) MGS(16, 64) MGS(16, 16) MGS(32, 64) MGS(32, 32) MGS(2, 64) MGS(2, 2)
    if (nb) atomicAdd(bad, nb);
}

int main() {
    float h[64 * 33];
    srand(3);
    for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 37.f;
    float* d; int* bad; int hb = 0;
    hipMalloc(&d, sizeof(h)); hipMalloc(&bad, 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d, bad);
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("xor_lane / group_sum / wave_max / multi_group_sum vs __shfl_xor: %d mismatching lanes\n", hb);
    return hb != 0;
}
