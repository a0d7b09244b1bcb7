// Diagnostic (not part of the library): does the SHAPE of a wave's 16-B-per-lane store instruction matter for the rate at
// which a CU can push an output tile to HBM?  Every workgroup (512 threads, one per CU) writes 256 x 256 bf16 tiles of a
// [M, 2048] matrix, as the tile256 GEMM epilogue does, in three shapes per wave-instruction:
//   A: 16 rows x 64 B  (the epilogue today: a lane holds 8 consecutive columns of one of 16 rows)
//   B:  8 rows x 128 B (whole 128-B lines)
//   C:  2 rows x 512 B
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void st(uint16_t* __restrict__ out, int N, int units_per_wg, int ncb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    for (int u = 0; u < units_per_wg; ++u) {
        const long unit = (long)u * gridDim.x + blockIdx.x;
        const long row0 = (unit / ncb) * 256 + wr * 128;
        const int col0 = (int)(unit % ncb) * 256 + wc * 64;          // this wave: 128 rows x 64 columns (128 B per row)
        if (SHAPE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<uint4*>(out + (row0 + 16 * t + (lane & 15)) * N + col0 + 8 * (lane >> 4) + 32 * h) = v;
        } else if (SHAPE == 1) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                *reinterpret_cast<uint4*>(out + (row0 + 8 * t + (lane >> 3)) * N + col0 + 8 * (lane & 7)) = v;
        } else {
            // the four waves of a wave row write 2 rows x 512 B together: wave wc takes rows 2*(4 t + wc) .. +1
#pragma unroll
            for (int t = 0; t < 16; ++t)
                *reinterpret_cast<uint4*>(out + (row0 + 2 * (4 * t + wc) + (lane >> 5)) * N + (col0 - wc * 64) + 8 * (lane & 31)) = v;
        }
    }
}

int main() {
    const int N = 2048, ncb = N / 256, M = 78848;
    const long units = (long)(M / 256) * ncb;
    const int upw = (int)(units / 256);
    for (int grid : {32, 64, 128, 256}) {
    uint16_t* out;
    CK(hipMalloc(&out, (size_t)M * N * 2));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int shape = 0; shape < 3; ++shape) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(a));
            if (shape == 0) hipLaunchKernelGGL(st<0>, dim3(grid), dim3(512), 0, 0, out, N, upw, ncb);
            else if (shape == 1) hipLaunchKernelGGL(st<1>, dim3(grid), dim3(512), 0, 0, out, N, upw, ncb);
            else hipLaunchKernelGGL(st<2>, dim3(grid), dim3(512), 0, 0, out, N, upw, ncb);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 1 && ms < best) best = ms;
        }
        const double bytes = (double)upw * grid * 256 * 256 * 2;
        printf("%3d workgroups, shape %c: %.1f us for %.0f MB  %.2f TB/s  (%.1f GB/s per CU, %.1f us per 128-KiB tile)\n", grid, "ABC"[shape], best * 1e3,
               bytes / 1e6, bytes / best / 1e9, bytes / best / 1e6 / grid, best * 1e3 / upw);
    }
    }
    return 0;
}
