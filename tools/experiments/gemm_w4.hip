// Experiment (round 2): the GEMM as FOUR waves per CU with the whole 512-register file per SIMD.
//
//   unit 256 x 256, waves 2 (M) x 2 (N), wave tile 128 x 128 = 4 x 4 accumulator tiles of v_mfma_f32_32x32x16_bf16
//   (256 accumulator registers); the finished tile is PARKED as packed bf16 (128 registers) and drips out under the next
//   unit's MFMAs, a few stores per K-tile, so the epilogue (11 of 24 us per c_fc unit in gemm_tile256_kernel, where no
//   register is left to park anything) disappears from the critical path.  Both operands by LDS-DMA into two 64-k K-tile
//   buffers (A 32 KiB + W 32 KiB each); one barrier per K-tile; program order inside a K-tile: parked-tile stores, then the
//   next K-tile's 16 LDS-DMAs, then nothing but fragment reads and MFMAs, so a plain vmcnt(0) at the K-tile boundary
//   waits only for operations that are a whole K-tile old.
//
// Standalone harness: builds to a binary that checks against a host reference on a small problem and times the c_fc shape.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float sigmoid1702(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896341f)));
}
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid1702(x); }

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
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

constexpr int W_THREADS = 256;
constexpr int W_KT = 32;                   // k extent of a K-tile
constexpr int W_PART = 256 * W_KT * 2;     // bytes of one operand's K-tile image: 256 rows x 64 B
constexpr int W_BUF = 2 * W_PART;          // A then W
constexpr int W_NBUF = 4;                  // ring: K-tile g+2 is staged while K-tile g is consumed
constexpr int W_BIAS = W_NBUF * W_BUF;     // fp32 bias copy behind the ring
constexpr int W_MAXN = 8192;
constexpr int W_LDS = W_BIAS + W_MAXN * 4;

// chunk swizzle of a 128-B row (conflict-free for the ds_read_b128 lane groups of the 32x32x16 operand read, natural and
// interleaved row order; searched exhaustively, see tools/experiments/README.md)
__host__ __device__ __forceinline__ int w_swz(int row) { return (row >> 2) & 3; }

enum { EPI_BF16 = 0, EPI_BIAS_GELU = 3 };

// W image: K-tile kt of column block cb is 32 KiB = 256 rows (columns of the block, natural order) x 8 chunks; chunk p of
// row n holds k = kt*64 + (p ^ swz(n))*8 .. +7
__global__ void pack_w4_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K) {
    const long total = (long)N * K / 8;
    const int nkt = K / W_KT;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const int pos = (int)(u & 1023);             // chunk index inside the 16-KiB K-tile image (256 rows x 4 chunks)
        const long t = u >> 10;
        const int kt = (int)(t % nkt), cb = (int)(t / nkt);
        const int row = pos >> 2;
        const int c = (pos & 3) ^ w_swz(row);
        const int n = cb * 256 + row;
        const int k0 = kt * W_KT + c * 8;
        uint32_t w[4];
        for (int j = 0; j < 4; ++j) w[j] = pack2bf(src[(long)n * K + k0 + 2 * j], src[(long)n * K + k0 + 2 * j + 1]);
        reinterpret_cast<uint4*>(dst)[u] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <int EPI>
__global__ __launch_bounds__(W_THREADS, 1) void gemm_w4_kernel(const bf16_t* __restrict__ A, const char* __restrict__ Wp,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ out0,
                                                               bf16_t* __restrict__ out1, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    const int nrb = (M + 255) >> 8, ncb = N >> 8;
    const int G_ = gridDim.x;
    const int X = (G_ & 7) == 0 ? 8 : 1;
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G_ / X;
    const int nrx = (nrb - xcd + X - 1) / X;
    const int U = (nrx * ncb - wg + WX - 1) / WX;
    if (U <= 0) return;
    const int nkt = K / W_KT;
    const long wcb_bytes = (long)nkt * W_PART;
    auto unit_of = [&](int t, int& row0, int& cb) {
        const int v = wg + t * WX;
        const int i = v / ncb;
        cb = v - i * ncb;
        row0 = (xcd + X * i) << 8;
    };
    if constexpr (EPI == EPI_BIAS_GELU) {
        for (int i = threadIdx.x * 4; i < N; i += W_THREADS * 4)
            *reinterpret_cast<float4*>(lds + W_BIAS + i * 4) = *reinterpret_cast<const float4*>(bias + i);
    }

    // ---- staging: a K-tile is 16 + 16 wave-instructions of 1 KiB (16 rows x 64 B); this wave issues 4 + 4 of them
    // instruction q (0..3) of this wave covers image rows (wave*4 + q)*16 + (lane>>2), chunk lane&3
    const uint32_t st_lds = lds_base + wave * 4096;
    auto stage_piece = [&](int slot, int pbuf, int row0, int cb, int kt) {
        const int q = slot & 3;
        if (slot < 4) {
            const int row = (wave * 4 + q) * 16 + (lane >> 2);
            int gr = row0 + row;
            gr = gr < M ? gr : M - 1;
            glds16(reinterpret_cast<const char*>(A) + ((long)gr * K + kt * W_KT) * 2 + ((lane & 3) ^ w_swz(row)) * 16,
                   st_lds + pbuf * W_BUF + q * 1024);
        } else {
            glds16(Wp + cb * wcb_bytes + (long)kt * W_PART + wave * 4096 + lane * 16 + q * 1024,
                   st_lds + pbuf * W_BUF + W_PART + q * 1024);
        }
    };
    auto stage = [&](int pbuf, int row0, int cb, int kt) {
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) stage_piece(sl, pbuf, row0, cb, kt);
    };

    // ---- fragment read offsets: activation rows natural; weight rows interleaved so that a lane's accumulator registers
    // 8g2 .. 8g2+7 are 8 CONSECUTIVE output columns (one 16-B bf16 store): tile row i = 8g + 4h + q -> column
    // 16(g>>1) + 8h + 4(g&1) + q
    const int arow0 = wr * 128 + l31;
    const int wrow0 = wc * 128 + 16 * ((l31 >> 3) >> 1) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
    // rows arow0 + 32 t / wrow0 + 32 i: swz(row + 32 t) = swz(row) (32 = 2^5, swz uses bits 1..3), so one XOR serves all tiles
    const int aswz = w_swz(arow0), wswz = w_swz(wrow0);

    f32x16 acc[4][4];                       // [weight (column) tile i][activation (row) tile t]
    uint4 park[4][4][2];                    // the previous unit's tile as packed bf16: [i][t][g2]: 8 consecutive columns
    bool have_park = false;
    int p_row0 = 0, p_cb = 0;

    auto init_acc = [&](int cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 b0;
#pragma unroll
            for (int r = 0; r < 16; ++r) b0[r] = 0.0f;
            if constexpr (EPI == EPI_BIAS_GELU) {
                // register r = 4 g + q of lane half lh is column 16(g>>1) + 8 lh + 4(g&1) + q of the 32-column tile
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = cb * 256 + wc * 128 + 32 * i + 16 * (g >> 1) + 8 * lh + 4 * (g & 1);
                    const float4 b4 = *reinterpret_cast<const float4*>(lds + W_BIAS + n * 4);
                    b0[4 * g + 0] = b4.x; b0[4 * g + 1] = b4.y; b0[4 * g + 2] = b4.z; b0[4 * g + 3] = b4.w;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = b0;
        }
    };

    // one parked 16-B piece: columns cb*256 + wc*128 + 32 i + 16 g2 + 8 lh .. +7 of row row0 + wr*128 + 32 t + l31
    auto drip = [&](int i, int t, int g2) {
        const int m = p_row0 + wr * 128 + 32 * t + l31;
        const long o = (long)m * N + p_cb * 256 + wc * 128 + 32 * i + 16 * g2 + 8 * lh;
        const uint4 hq = park[i][t][g2];
        if constexpr (EPI == EPI_BF16) {
            *reinterpret_cast<uint4*>(out0 + o) = hq;
        } else {
            const uint32_t hp[4] = {hq.x, hq.y, hq.z, hq.w};
            uint32_t up[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                up[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
            }
            *reinterpret_cast<uint4*>(out0 + o) = hq;
            *reinterpret_cast<uint4*>(out1 + o) = make_uint4(up[0], up[1], up[2], up[3]);
        }
    };

    int c_row0, c_cb;
    unit_of(0, c_row0, c_cb);
    int n_row0 = c_row0, n_cb = c_cb, n_kt = 0, n_t = 0;       // cursor of the NEXT K-tile to stage
    auto advance = [&]() {
        if (n_kt + 1 < nkt) {
            ++n_kt;
        } else if (n_t + 1 < U) {
            n_kt = 0;
            ++n_t;
            unit_of(n_t, n_row0, n_cb);
        } else {
            n_kt = -1;                                             // nothing left to stage
        }
    };
    // prologue: K-tiles 0 and 1 (a unit has at least 2: K >= 64)
    stage(0, n_row0, n_cb, n_kt);
    advance();
    stage(1, n_row0, n_cb, n_kt);
    advance();

    // volatile: pins the conversion order tile by tile, so that only one accumulator tile's worth of values is ever
    // copied out of the accumulator file at a time
    auto cvt2 = [](float lo, float hi) {
        uint32_t r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    };
    const int gper = (16 + nkt - 1) / nkt;                        // parked groups (one accumulator tile each) per K-tile
    int pb = 0;
    bool staged_prev = true;                                      // the prologue staged K-tile 1 after K-tile 0
    for (int tu = 0; tu < U; ++tu) {
        init_acc(c_cb);
        for (int kt = 0; kt < nkt; ++kt) {
            // this wave's share of the K-tile has landed: everything but the 8 youngest operations (= the pieces of the
            // K-tile after it, staged during the previous K-tile; parked stores are issued BEFORE a K-tile's pieces)
            if (staged_prev) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the last K-tiles: nothing younger to hide behind
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const char* ab = lds + pb * W_BUF;
            const char* wb = ab + W_PART;
            bf16x8 af[2][4], wf[2][4];
            auto read_frags = [&](int ks, bf16x8 (&fa)[4], bf16x8 (&fw)[4]) {
                const int ca = (((2 * ks + lh) ^ aswz) << 4), cw = (((2 * ks + lh) ^ wswz) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const bf16x8*>(ab + (arow0 + 32 * t) * 64 + ca);
#pragma unroll
                for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wb + (wrow0 + 32 * i) * 64 + cw);
            };
            // side work, one piece per group of 4 MFMAs (8 slots): slot s stages piece s of K-tile g+2 into the ring slot
            // that K-tile g-2 occupied (every wave passed the barrier above after its reads of g-2 ... g-1);
            // the parked groups of this K-tile go first
            const bool do_stage = n_kt >= 0;
            const int s_row0 = n_row0, s_cb = n_cb, s_kt = n_kt, s_pb = (pb + 2) & 3;
            auto side = [&](int slot) {
                if (slot == 0 && have_park) for (int j = 0; j < gper; ++j) {
                    const int grp = kt * gper + j;                // wave-uniform -> one scalar jump per group
                    switch (grp) {
#define W4_CASE(G) case G: drip(G >> 2, G & 3, 0); drip(G >> 2, G & 3, 1); break;
                        W4_CASE(0) W4_CASE(1) W4_CASE(2) W4_CASE(3) W4_CASE(4) W4_CASE(5) W4_CASE(6) W4_CASE(7)
                        W4_CASE(8) W4_CASE(9) W4_CASE(10) W4_CASE(11) W4_CASE(12) W4_CASE(13) W4_CASE(14) W4_CASE(15)
#undef W4_CASE
                        default: break;
                    }
                }
                if (do_stage) stage_piece(slot, s_pb, s_row0, s_cb, s_kt);
            };
            read_frags(0, af[0], wf[0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks + 1 < 2) read_frags(ks + 1, af[(ks + 1) & 1], wf[(ks + 1) & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][i], af[ks & 1][t], acc[i][t], 0, 0, 0);
                    side(ks * 4 + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (do_stage) advance();
            staged_prev = do_stage;
            pb = (pb + 1) & 3;
        }
#ifdef W4_NOPARK
        {   // main loop only (timing experiment): keep the accumulators alive, store nothing
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) sink += acc[i][t][0] + acc[i][t][15];
            if (sink == 1234.5f) out0[threadIdx.x] = 0;
            if (tu + 1 < U) unit_of(tu + 1, c_row0, c_cb);
            continue;
        }
#endif
        // the unit is complete: park it as bf16 (what was parked from the unit before has been stored by now)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x16 a = acc[i][t];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
                    park[i][t][g2] = make_uint4(cvt2(a[8 * g2 + 0], a[8 * g2 + 1]), cvt2(a[8 * g2 + 2], a[8 * g2 + 3]),
                                                cvt2(a[8 * g2 + 4], a[8 * g2 + 5]), cvt2(a[8 * g2 + 6], a[8 * g2 + 7]));
            }
        have_park = true;
        p_row0 = c_row0;
        p_cb = c_cb;
        if (tu + 1 < U) unit_of(tu + 1, c_row0, c_cb);
    }
    // the last unit's tile
#ifndef W4_NOPARK
#pragma unroll
    for (int pcs = 0; pcs < 32; ++pcs) drip(pcs >> 3, (pcs >> 1) & 3, pcs & 1);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ host harness
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <int EPI>
static void launch(const bf16_t* A, const bf16_t* Wp, const float* bias, bf16_t* o0, bf16_t* o1, int M, int N, int K) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)gemm_w4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS)); attr = true; }
    const long units = (long)((M + 255) / 256) * (N / 256);
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < 256 ? want : 256);
    hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3(grid), dim3(W_THREADS), W_LDS, 0, A, reinterpret_cast<const char*>(Wp), bias, o0, o1, M, N, K);
    CK(hipGetLastError());
}

static int check(int M, int N, int K, int epi) {
    std::vector<float> hw((size_t)N * K), hb(N);
    std::vector<uint16_t> ha((size_t)M * K);
    srand(1);
    for (auto& v : ha) v = f2bf((float)(rand() % 17 - 8) / 8.0f);
    for (auto& v : hw) v = (float)(rand() % 15 - 7) / 16.0f;
    for (auto& v : hb) v = (float)(rand() % 9 - 4) / 4.0f;
    bf16_t *dA, *dWp, *o0, *o1; float *dW, *dB;
    CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dWp, hw.size() * 2)); CK(hipMalloc(&dW, hw.size() * 4)); CK(hipMalloc(&dB, N * 4));
    CK(hipMalloc(&o0, (size_t)M * N * 2)); CK(hipMalloc(&o1, (size_t)M * N * 2));
    CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(o0, 0xff, (size_t)M * N * 2)); CK(hipMemset(o1, 0xff, (size_t)M * N * 2));
    hipLaunchKernelGGL(pack_w4_kernel, dim3(1024), dim3(256), 0, 0, dW, dWp, N, K);
    if (epi == EPI_BF16) launch<EPI_BF16>(dA, dWp, dB, o0, o1, M, N, K); else launch<EPI_BIAS_GELU>(dA, dWp, dB, o0, o1, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> r0((size_t)M * N), r1((size_t)M * N);
    CK(hipMemcpy(r0.data(), o0, r0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), o1, r1.size() * 2, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int m = 0; m < M; m += 7)
        for (int n = 0; n < N; ++n) {
            double s = epi == EPI_BF16 ? 0.0 : hb[n];
            for (int k = 0; k < K; ++k) s += (double)bf2f(ha[(size_t)m * K + k]) * (double)bf2f(f2bf(hw[(size_t)n * K + k]));
            const float got = bf2f(r0[(size_t)m * N + n]);
            if (fabs(got - (float)s) > 0.02 * fabs(s) + 0.02) { if (bad < 5) printf("  mismatch m=%d n=%d got %f want %f\n", m, n, got, s); ++bad; }
            if (epi == EPI_BIAS_GELU) {
                const float h = bf2f(f2bf((float)s)), u = h / (1.0f + expf(-1.702f * h));
                const float gu = bf2f(r1[(size_t)m * N + n]);
                if (fabs(gu - u) > 0.02 * fabs(u) + 0.02) { if (bad < 5) printf("  gelu mismatch m=%d n=%d got %f want %f\n", m, n, gu, u); ++bad; }
            }
        }
    printf("check M=%d N=%d K=%d epi=%d: %ld mismatches\n", M, N, K, epi, bad);
    hipFree(dA); hipFree(dWp); hipFree(dW); hipFree(dB); hipFree(o0); hipFree(o1);
    return bad != 0;
}

static void bench(int M, int N, int K, int epi, const char* name) {
    bf16_t *dA, *dWp, *o0, *o1; float *dW, *dB;
    CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dWp, (size_t)N * K * 2)); CK(hipMalloc(&dW, (size_t)N * K * 4)); CK(hipMalloc(&dB, N * 4));
    CK(hipMalloc(&o0, (size_t)M * N * 2)); CK(hipMalloc(&o1, (size_t)M * N * 2));
    std::vector<uint16_t> ha((size_t)M * K); std::vector<float> hw((size_t)N * K), hb(N, 0.1f);
    srand(2);
    for (auto& v : ha) v = f2bf((float)rand() / RAND_MAX * 2 - 1);
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_w4_kernel, dim3(1024), dim3(256), 0, 0, dW, dWp, N, K);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, sum = 0;
    const int R = 12;
    for (int r = 0; r < R + 2; ++r) {
        CK(hipEventRecord(a));
        if (epi == EPI_BF16) launch<EPI_BF16>(dA, dWp, dB, o0, o1, M, N, K); else launch<EPI_BIAS_GELU>(dA, dWp, dB, o0, o1, M, N, K);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-22s M=%d N=%d K=%d  avg %.1f us  min %.1f us  %.1f TFLOP/s\n", name, M, N, K, sum / R * 1e3, best * 1e3, 2.0 * M * N * K / (sum / R * 1e-3) / 1e12);
    hipFree(dA); hipFree(dWp); hipFree(dW); hipFree(dB); hipFree(o0); hipFree(o1);
}

int main() {
    int bad = 0;
#ifndef W4_NOPARK
    bad |= check(768, 512, 256, EPI_BF16);
    bad |= check(1024, 1024, 512, EPI_BIAS_GELU);
#endif
    if (bad) { printf("FAILED\n"); return 1; }
    bench(78848, 2048, 512, EPI_BIAS_GELU, "c_fc  BIAS_GELU");
    bench(78848, 1536, 512, EPI_BF16, "qkv   (BF16 epi)");
    bench(59136, 512, 2048, EPI_BF16, "dfc   BF16");
    bench(59136, 512, 1536, EPI_BF16, "dqkv  BF16");
    bench(59136, 512, 512, EPI_BF16, "do    BF16");
    bench(16384, 2048, 2048, EPI_BF16, "square-ish BF16");
    return 0;
}
