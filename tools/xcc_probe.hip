// Which XCD does workgroup blockIdx.x of a launch land on?  Every product GEMM deals its units "XCD-aware" on the
// assumption blockIdx.x % 8 == XCD; this probe reads the hardware's XCC_ID register per workgroup for the launch shapes
// the GEMMs use (256 x 512 threads with 160 KiB LDS: one per CU; 512 x 512 threads with 64 KiB: two per CU).
// build: hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o tools/xcc_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(int* xcc, int* cu) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x & 0xf;
        cu[blockIdx.x] = h;
    }
    // stay resident long enough that the whole grid is placed before anyone leaves
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}

int main() {
    int shapes[3][2] = {{256, 160 * 1024}, {512, 64 * 1024}, {232, 64 * 1024}};
    for (auto& s : shapes) {
        int n = s[0], ldsb = s[1];
        int *dx, *dc;
        hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(n), dim3(512), ldsb, 0, dx, dc);
            hipDeviceSynchronize();
        }
        std::vector<int> x(n), c(n);
        hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        int match = 0;
        for (int i = 0; i < n; ++i) match += (x[i] == i % 8);
        printf("grid %d, %d KiB LDS: blockIdx %% 8 == XCC_ID for %d of %d workgroups\n  first 32 XCC ids:", n, ldsb / 1024, match, n);
        for (int i = 0; i < 32 && i < n; ++i) printf(" %d", x[i]);
        printf("\n");
        int hist[16] = {0};
        for (int i = 0; i < n; ++i) hist[x[i] & 15]++;
        printf("  workgroups per XCC:");
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("\n");
        hipFree(dx); hipFree(dc);
    }
    return 0;
}
