// Diagnostic (not part of the library): what clock does this MI355X hold under a dense bf16 MFMA load on random data,
// and what MFMA rate does that clock allow?  A bare v_mfma_f32_16x16x32_bf16 loop (operands in registers, 16 independent
// accumulators per wave) on every CU, 1 or 2 waves per SIMD, run for >= 2 s of back-to-back launches; the in-kernel
// clock is delta(s_memtime) / delta(s_memrealtime) x 100 MHz around the loop (MI355X_MICROARCH.md, DVFS give-back (6)).
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ seed, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    // 4 + 4 different random fragments per lane, so that consecutive MFMAs see different operands
    const int nthr = gridDim.x * blockDim.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, seed[tid + (2 * i) * nthr]);
        b[i] = __builtin_bit_cast(bf16x8, seed[tid + (2 * i + 1) * nthr]);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) sink[tid] = t;          // keeps the loop alive, (almost) never stores
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    for (int wps = 1; wps <= 2; ++wps) {                 // waves per SIMD
        const int threads = 256 * wps, grid = cus, iters = 40000;
        const size_t n = (size_t)grid * threads;
        std::vector<uint32_t> h(n * 8 * 4);
        uint32_t x = 12345u;
        for (auto& v : h) {                              // random bf16 pairs in [-2, 2): sign, exponent 0x3e..0x3f region
            x = x * 1664525u + 1013904223u;
            const uint32_t lo = 0x3e00u + ((x >> 8) & 0x1ffu) + ((x & 1u) << 15);
            const uint32_t hi = 0x3e00u + ((x >> 20) & 0x1ffu) + ((x & 2u) << 14);
            v = lo | (hi << 16);
        }
        uint4* dseed; float* dsink; unsigned long long* dst;
        CK(hipMalloc(&dseed, n * 32 * 4)); CK(hipMalloc(&dsink, n * 4)); CK(hipMalloc(&dst, n / 64 * 16));
        CK(hipMemcpy(dseed, h.data(), n * 32 * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0.f, total = 0.f;
        int launches = 0;
        while (total < 2500.f) {                          // >= 2 s of back-to-back launches, time the last one
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(threads), 0, 0, dseed, dsink, dst, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            total += ms; ++launches;
        }
        std::vector<unsigned long long> st(n / 64 * 2);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (size_t w = 0; w < n / 64; ++w) { ghz.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1); cyc.push_back((double)st[2 * w]); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double flops = 2.0 * 16 * 16 * 32 * 16.0 * iters * (double)(n / 64);
        printf("bare 16x16x32 bf16 MFMA loop, random operands, %d CUs, %d wave(s) per SIMD: %.2f ms per launch (%d launches), "
               "%.0f TFLOP/s, in-kernel clock median %.3f GHz (min %.3f max %.3f), %.2f cycles per MFMA per SIMD\n",
               cus, wps, ms, launches, flops / ms * 1e-9, ghz[ghz.size() / 2], ghz.front(), ghz.back(),
               cyc[cyc.size() / 2] / (16.0 * iters) / wps);
        CK(hipFree(dseed)); CK(hipFree(dsink)); CK(hipFree(dst));
    }
    return 0;
}
