// "Drip" variant of the activation-stationary bf16 GEMM (see gemm.hip for the operand layout and the unit deal).
//
// Why it exists (profiles/r01_gemm_cfc_timeline_*.txt): in gemm.hip a workgroup finishes a 128 x 256 unit and
// then stores the whole tile.  All workgroups run the same schedule, so the chip alternates between store
// bursts (HBM saturated, MFMA idle) and compute phases (HBM idle), and -- vmcnt retiring in order -- the first
// weight loads of the next unit wait behind the burst.  Here the finished tile is kept in a second accumulator
// set and leaves ONE 16 x 32 chunk per LDS piece, behind the MFMAs of the next unit:
//
//  * ONE 512-thread workgroup per CU, 8 waves, <= 256 VGPRs: acc[2][8] (current unit) + prev[2][8] (previous);
//  * activations stream through a ring of THREE [128 x 128-k] LDS pieces filled by LDS-DMA two pieces ahead,
//    every wave issuing 4 of the 32 wave-instructions of a piece;
//  * the packed weight fragments of half a piece (2 k-steps) sit in one of two register sets and are fetched
//    one piece ahead by inline-asm loads -- every VMEM operation of the loop is issued from inline asm or is a
//    store, so the in-order vmcnt queue is counted BY HAND (hipcc's own counts cannot see the asm operations):
//
//        piece j:  wait fills(j) [vmcnt 20] | barrier | fills(j+2) [4] |
//                  wait W(2j)   [vmcnt 8]  | 32 MFMA | load W(2j+2) [4] |
//                  wait W(2j+1) [vmcnt 8]  | 32 MFMA | load W(2j+3) [4] | drip chunk(s): stores
//
//    counts = operations issued AFTER the awaited one, stores counted as 0: a predicated store may not issue, and
//    counting low only waits for the next-oldest loads, which are needed next anyway.  Every piece issues exactly
//    4 + 4 + 4 loads (addresses are clamped past the end), so the counts hold from the first piece to the last;
//  * stores are the YOUNGEST operations of a piece: no wait in the loop ever has to retire a store of the same
//    piece, and the loads of the next piece are never queued behind more than 2-4 stores.
#include <stdlib.h>

#include "common.h"

namespace neat {

constexpr int D_THREADS = 512;
constexpr int D_WAVES = 8;
constexpr int D_NI = 2;
constexpr int D_BKH = 128;
constexpr int D_KSTEPS = D_BKH / 32;          // 4
constexpr int D_CPR = D_BKH / 8;              // 16-B chunks per LDS row
constexpr int D_BUF = 128 * D_BKH * 2;        // 32 KiB per piece
constexpr int D_RING = 3;
constexpr int D_SLAB_OFF = D_RING * D_BUF;    // 96 KiB
constexpr int D_BIAS_OFF = D_SLAB_OFF + D_WAVES * D_NI * 1024;   // 112 KiB
constexpr int D_CB = D_WAVES * D_NI * 16;     // 256 columns per unit

__device__ __forceinline__ void d_glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 d_gload16(const uint4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void d_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// values produced by an asm load may be used only after the wait: make them opaque at this point so that no use
// is scheduled above it
__device__ __forceinline__ void d_pin(u32x4 (&w)[2][D_NI]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < D_NI; ++b) asm volatile("" : "+v"(w[a][b]));
}

// two k-steps: 2 x (8 activation fragments from LDS, 16 MFMAs)
__device__ __forceinline__ void d_mma_half(f32x4 (&acc)[D_NI][8], const u32x4 (&w)[2][D_NI], const char* abase,
                                           int h, int fr, int fq) {
    constexpr int rowbytes = D_CPR * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kk = 2 * h + ks;
        bf16x8 wf[D_NI];
#pragma unroll
        for (int i = 0; i < D_NI; ++i) wf[i] = __builtin_bit_cast(bf16x8, w[ks][i]);
        const int coff = ((kk * 4 + fq) ^ fr) * 16;
#pragma unroll
        for (int tg = 0; tg < 8; tg += 4) {
            bf16x8 af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[t] = *reinterpret_cast<const bf16x8*>(abase + (tg + t) * 16 * rowbytes + coff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < D_NI; ++i)
                    acc[i][tg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[t], acc[i][tg + t], 0, 0, 0);
        }
    }
}

struct DUnit {
    int sub, cb, nt0, nvalid;   // sub-panel, column block, first column tile of this wave, live tiles (may be <= 0)
};

template <int EPI>
__global__ __launch_bounds__(D_THREADS, 2) void gemm_drip_kernel(const bf16_t* __restrict__ A,
                                                                  const uint4* __restrict__ Wp,
                                                                  const float* __restrict__ bias, void* out0,
                                                                  void* out1, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // same XCD-aware unit deal as gemm.hip
    const int nsub = (M + 127) >> 7;
    const int ncb = (N + D_CB - 1) / D_CB;
    const int G = gridDim.x;
    const int X = (G & 7) == 0 ? 8 : 1;
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G / X;
    const int nsx = (nsub - xcd + X - 1) / X;
    const int myunits = (nsx * ncb - wg + WX - 1) / WX;
    if (myunits <= 0) return;
    const int nkh = K / D_BKH;
    const int ksK = K >> 5;
    const int J = myunits * nkh;

    if constexpr (EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_GELU) {
        float* lb = reinterpret_cast<float*>(lds + D_BIAS_OFF);
        for (int i = threadIdx.x; i < N; i += D_THREADS) lb[i] = bias[i];
    }
    __syncthreads();   // also drains the bias loads: from here on every load is issued from inline asm

    auto unit_at = [&](int t) {
        DUnit u;
        const int v = wg + t * WX;
        const int i = v / ncb;
        u.cb = v - i * ncb;
        u.sub = xcd + X * i;
        u.nt0 = (u.cb * D_CB >> 4) + wave * D_NI;
        u.nvalid = (N >> 4) - u.nt0;
        return u;
    };
    // LDS-DMA of data piece jd into ring slot js: 4 wave-instructions per wave
    auto fill = [&](int jd, int js) {
        const int t = jd / nkh;
        const int kh = jd - t * nkh;
        const DUnit u = unit_at(t);
        const int row0 = u.sub << 7;
        const uint32_t buf = lds_base + (js % D_RING) * D_BUF;
        const bf16_t* src = A + kh * D_BKH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = wave + 8 * r;
            const int row = q * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            int gr = row0 + row;
            gr = gr < M ? gr : M - 1;
            d_glds16(src + (long)gr * K + c * 8, buf + q * 1024);
        }
    };
    // the 4 packed-W fragments of half-piece q (2 k-steps x NI tiles); idle / missing tiles re-read a valid one
    auto wload = [&](u32x4 (&w)[2][D_NI], int q) {
        int jj = q >> 1;
        jj = jj < J ? jj : J - 1;
        const int h = q & 1;
        const int t = jj / nkh;
        const int kh = jj - t * nkh;
        const DUnit u = unit_at(t);
        const int nv = u.nvalid > 0 ? u.nvalid : 1;
        const int n0 = u.nvalid > 0 ? u.nt0 : 0;
        const uint4* wp = Wp + ((long)n0 * ksK + (long)kh * D_KSTEPS + 2 * h) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < D_NI; ++i) w[ks][i] = d_gload16(wp + (long)(i < nv ? i : nv - 1) * ksK * 64 + ks * 64);
    };

    // one 16-row x 32-column chunk of a finished tile: transpose through the wave's slab, finish, store 16 B/lane
    char* stg = lds + D_SLAB_OFF + wave * (D_NI * 1024);
    auto sw = [](int r) { return (r & 3) ^ ((r & 4) ? 5 : 0); };
    const int er = lane >> 2, eq = lane & 3;
    const char* rd0 = stg + er * 128 + (((2 * eq) ^ sw(er)) << 4);
    const char* rd1 = stg + er * 128 + (((2 * eq + 1) ^ sw(er)) << 4);
    char* wr0 = stg + fr * 128 + (((0 * 4 + fq) ^ sw(fr)) << 4);
    char* wr1 = stg + fr * 128 + (((1 * 4 + fq) ^ sw(fr)) << 4);
    auto chunk = [&](const f32x4 p0, const f32x4 p1, int t, const DUnit& u) {
        *reinterpret_cast<f32x4*>(wr0) = p0;
        *reinterpret_cast<f32x4*>(wr1) = p1;
        const float4 a0 = *reinterpret_cast<const float4*>(rd0);
        const float4 a1 = *reinterpret_cast<const float4*>(rd1);
        const int ncols = u.nvalid < D_NI ? u.nvalid * 16 : D_NI * 16;       // <= 0 for an idle wave
        const int n = u.nt0 * 16 + eq * 8;
        const int m = (u.sub << 7) + t * 16 + er;
        if (m >= M || eq * 8 >= ncols) return;
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if constexpr (EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_GELU) {
            const float* lb = reinterpret_cast<const float*>(lds + D_BIAS_OFF) + n;
            const float4 b0 = *reinterpret_cast<const float4*>(lb);
            const float4 b1 = *reinterpret_cast<const float4*>(lb + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        const long o = (long)m * N + n;
        if constexpr (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16) {
            *reinterpret_cast<uint4*>((bf16_t*)out0 + o) =
                make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        } else {  // NEAT_EPI_BIAS_GELU: h rounded to bf16 once, GELU of the rounded value (as gemm.hip)
            uint32_t hp[4], up[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hp[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                up[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
            }
            *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
            *reinterpret_cast<uint4*>((bf16_t*)out1 + o) = make_uint4(up[0], up[1], up[2], up[3]);
        }
    };

    f32x4 acc[D_NI][8], prev[D_NI][8];
    auto drip = [&](int t, const DUnit& u) {      // t is wave-uniform: the switch is scalar branches, no indexing
        switch (t) {
            case 0: chunk(prev[0][0], prev[1][0], 0, u); break;
            case 1: chunk(prev[0][1], prev[1][1], 1, u); break;
            case 2: chunk(prev[0][2], prev[1][2], 2, u); break;
            case 3: chunk(prev[0][3], prev[1][3], 3, u); break;
            case 4: chunk(prev[0][4], prev[1][4], 4, u); break;
            case 5: chunk(prev[0][5], prev[1][5], 5, u); break;
            case 6: chunk(prev[0][6], prev[1][6], 6, u); break;
            default: chunk(prev[0][7], prev[1][7], 7, u); break;
        }
    };

    // prologue: 4 + 4 + 4 + 4 loads
    u32x4 w0[2][D_NI], w1[2][D_NI];
    fill(0, 0);
    fill(1 < J ? 1 : J - 1, 1);
    wload(w0, 0);
    wload(w1, 1);

    DUnit cur = unit_at(0), done = cur;
    bool have_prev = false;
    int kh = 0, tu = 0;
    for (int j = 0; j < J; ++j) {
        // fills(j) of THIS wave have landed: 20 loads were issued after them (W(2j-2..2j+1), fills(j+1)); then the
        // barrier makes every wave's part visible and frees ring slot (j+2) % 3 = (j-1) % 3
        if (j == 0) d_wait_vm<0>(); else d_wait_vm<20>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill(j + 2 < J ? j + 2 : J - 1, j + 2);

        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < D_NI; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* abase = lds + (j % D_RING) * D_BUF + fr * (D_CPR * 16);
        d_wait_vm<8>();                      // W(2j): behind it W(2j+1) and fills(j+2)
        d_pin(w0);
        d_mma_half(acc, w0, abase, 0, fr, fq);
        wload(w0, 2 * j + 2);
        d_wait_vm<8>();                      // W(2j+1): behind it fills(j+2) and W(2j+2)
        d_pin(w1);
        d_mma_half(acc, w1, abase, 1, fr, fq);
        wload(w1, 2 * j + 3);

        // drip: the chunks of the previous unit that belong to this piece slot (8 chunks over nkh slots)
        if (have_prev) {
            const int c_lo = (8 * kh + nkh - 1) / nkh, c_hi = (8 * (kh + 1) + nkh - 1) / nkh;
            for (int c = c_lo; c < c_hi; ++c) drip(c, done);
        }
        if (kh == nkh - 1) {
#pragma unroll
            for (int i = 0; i < D_NI; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) prev[i][t] = acc[i][t];
            done = cur;
            have_prev = true;
            kh = 0;
            ++tu;
            if (tu < myunits) cur = unit_at(tu);
        } else {
            ++kh;
        }
    }
    // the last unit has no successor to hide behind
    d_wait_vm<0>();
#pragma unroll
    for (int c = 0; c < 8; ++c) chunk(prev[0][c], prev[1][c], c, done);
}

static int d_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}

template <int EPI>
static int launch_drip(const uint16_t* A, const uint16_t* Wp, const float* bias, void* out0, void* out1, int M,
                       int N, int K, hipStream_t s) {
    const int ldsb = D_BIAS_OFF + N * 4;
    static int attr_set = 0;   // per template instance: largest LDS size requested so far
    if (attr_set < ldsb) {
        int rc = check_hip(hipFuncSetAttribute((const void*)gemm_drip_kernel<EPI>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ldsb),
                           "hipFuncSetAttribute(gemm_drip)");
        if (rc) return rc;
        attr_set = ldsb;
    }
    const int cus = d_num_cus();
    const long units = (long)cdiv(M, 128) * cdiv(N, D_CB);
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < cus ? want : cus);
    hipLaunchKernelGGL((gemm_drip_kernel<EPI>), dim3(grid), dim3(D_THREADS), ldsb, s, A,
                       reinterpret_cast<const uint4*>(Wp), bias, out0, out1, M, N, K);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16(drip)");
    return NEAT_OK;
}

// Shapes the drip kernel takes: store-only epilogues with both outputs present, K in whole 128-pieces and at
// least 4 of them (<= 2 chunks per piece), bias vector within the LDS left over, enough units to fill the chip.
bool gemm_drip_ok(int M, int N, int K, int epilogue, const void* out0) {
    if (!(epilogue == NEAT_EPI_BF16 || epilogue == NEAT_EPI_BIAS_BF16 || epilogue == NEAT_EPI_BIAS_GELU)) return false;
    if (epilogue == NEAT_EPI_BIAS_GELU && !out0) return false;
    if (K % D_BKH != 0 || K < 4 * D_BKH || N > 8192) return false;
    return (long)cdiv(M, 128) * cdiv(N, D_CB) >= d_num_cus();
}

int launch_gemm_drip(const uint16_t* A, const uint16_t* Wp, const float* bias, void* out0, void* out1, int M,
                     int N, int K, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case NEAT_EPI_BF16: return launch_drip<NEAT_EPI_BF16>(A, Wp, bias, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_BF16: return launch_drip<NEAT_EPI_BIAS_BF16>(A, Wp, bias, out0, out1, M, N, K, s);
        default: return launch_drip<NEAT_EPI_BIAS_GELU>(A, Wp, bias, out0, out1, M, N, K, s);
    }
}

}  // namespace neat
