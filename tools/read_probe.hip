// Diagnostic (not part of the library): pure-read HBM rate of a streaming kernel as a function of workgroups per CU and
// 16-B loads in flight per lane; 2 GiB buffer, every element read once.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ p, long nvec, float* __restrict__ out) {
    float acc = 0.f;
    const long stride = (long)gridDim.x * 256 * U;
    for (long g = (long)blockIdx.x * 256 * U + threadIdx.x; g + 256L * (U - 1) < nvec; g += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (NT) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p + g + 256L * u)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = p[g + 256L * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1234.5f) out[threadIdx.x] = acc;
}
// contiguous chunk per workgroup (each wave sweeps its own 1/4 of it), instead of the grid-stride interleave
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const float4* __restrict__ p, long nvec, float* __restrict__ out) {
    float acc = 0.f;
    const long per = nvec / gridDim.x;
    const float4* q = p + (long)blockIdx.x * per;
    for (long g = threadIdx.x; g + 256L * (U - 1) < per; g += 256L * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[g + 256L * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1234.5f) out[threadIdx.x] = acc;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

int main() {
    const long bytes = 2L << 30, nvec = bytes / 16;
    float4* p; float* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(p, 1, bytes));
    int cus = 256;
    for (int wg : {1, 2, 4, 8, 16, 32}) {
        const int grid = cus * wg;
        float a = timeit([&] { hipLaunchKernelGGL((rd<4, false>), dim3(grid), dim3(256), 0, 0, p, nvec, out); });
        float b = timeit([&] { hipLaunchKernelGGL((rd<8, false>), dim3(grid), dim3(256), 0, 0, p, nvec, out); });
        float c = timeit([&] { hipLaunchKernelGGL((rd<16, false>), dim3(grid), dim3(256), 0, 0, p, nvec, out); });
        float d = timeit([&] { hipLaunchKernelGGL((rd<8, true>), dim3(grid), dim3(256), 0, 0, p, nvec, out); });
        float e = timeit([&] { hipLaunchKernelGGL((rd_chunk<8>), dim3(grid), dim3(256), 0, 0, p, nvec, out); });
        printf("%2d WG/CU: U=4 %.2f  U=8 %.2f  U=16 %.2f  U=8 nt %.2f  U=8 chunked %.2f TB/s\n", wg, bytes / a * 1e-9, bytes / b * 1e-9,
               bytes / c * 1e-9, bytes / d * 1e-9, bytes / e * 1e-9);
    }
    return 0;
}
