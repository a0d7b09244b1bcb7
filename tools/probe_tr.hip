// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds lds[i] = i (16-bit); every lane passes
// the address of (row q = (lane&15)>>2 of its 16-lane group's block, cols 4*(lane&3)..+3) with 64-element rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    // block for group grp: rows 4*grp .. 4*grp+3, columns 0..15, row pitch 64 elements
    const short* addr = lds + (4 * grp + q) * 64 + 4 * p;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)addr);
    for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) printf("lane %2d: %5d %5d %5d %5d   (row,col)= (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
        h[l*4]/64, h[l*4]%64, h[l*4+1]/64, h[l*4+1]%64, h[l*4+2]/64, h[l*4+2]%64, h[l*4+3]/64, h[l*4+3]%64);
    return 0;
}
