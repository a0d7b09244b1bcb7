// Probe: in-place x += v over a [78848, 512] fp32 matrix, three ways -- (a) float4 load + add + store, (b) no-return
// global_atomic_add_f32 with consecutive lanes on consecutive dwords, (c) the same with each lane owning 4 consecutive
// dwords (the accumulator layout of the GEMM epilogue).  Prints us per pass and TB/s (8 B per element of HBM traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void rmw_kernel(float* x, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(x)[i];
        a.x += v; a.y += v; a.z += v; a.w += v;
        reinterpret_cast<float4*>(x)[i] = a;
    }
}
__global__ void atomic_dword_kernel(float* x, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        unsafeAtomicAdd(x + i, v);
}
__global__ void atomic_lane4_kernel(float* x, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) unsafeAtomicAdd(x + i * 4 + j, v);
    }
}

int main() {
    const long rows = 78848, cols = 512, n = rows * cols;
    float* x;
    CK(hipMalloc(&x, n * 4));
    CK(hipMemset(x, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8, block = 512, reps = 20;
    for (int mode = 0; mode < 3; ++mode) {
        for (int r = 0; r < reps + 3; ++r) {
            if (r == 3) CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(rmw_kernel, dim3(grid), dim3(block), 0, 0, x, n / 4, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(atomic_dword_kernel, dim3(grid), dim3(block), 0, 0, x, n, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(atomic_lane4_kernel, dim3(grid), dim3(block), 0, 0, x, n / 4, 1.0f);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("mode %d (%s): %.1f us per pass, %.2f TB/s\n", mode,
               mode == 0 ? "load+add+store float4" : mode == 1 ? "atomic add, lane = dword" : "atomic add, lane = 4 dwords", us,
               n * 8.0 / us * 1e-6);
    }
    std::vector<float> h(16);
    CK(hipMemcpy(h.data(), x, 64, hipMemcpyDeviceToHost));
    printf("x[0] = %.1f (expect %d)\n", h[0], 3 * (reps + 3));
    CK(hipFree(x));
    return 0;
}
