// bf16 MFMA GEMM with fused epilogues for the encoder linears.
// Replaces nn.Linear / nn.MultiheadAttention projections (reference open_clip/transformer.py:209-213,
// 227-229) and their data-gradient GEMMs (weights are frozen: there is no wgrad on this path).
//
//   out[M,N] = epilogue( A[M,K] . W[N,K]^T )      A = activations (row-major), W = weights (packed)
//
// Shape regime of this path: M = 4*B*77 ~ 8e4 rows, K = 512..2048, N = 512..2048.  At these sizes the
// activation/output streams are as expensive as the MFMAs (arithmetic intensity sits at the HBM/MFMA
// ridge), so the kernel is built around the ACTIVATION stream ("activation-stationary"):
//
//  * work unit = 128 rows x 256 columns; persistent grid of TWO 512-thread workgroups per CU (16 waves,
//    4 per SIMD, <= 128 VGPRs each); units are dealt so that all column blocks of a sub-panel run on one
//    XCD (one L2): measured with FETCH_SIZE, the plain round-robin deal fetched every panel N/256 times.  Two independent workgroups per CU drift out
//    of phase, so one's epilogue (a per-CU store stream of ~10 B/clk) overlaps the other's MFMAs;
//  * the unit's 128-row sub-panel streams through LDS in [128 x 128-k] pieces, double buffered
//    (2 x 32 KiB = all the LDS a workgroup owns, so two of them fit a CU), filled by global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip).  The LDS image is
//    lane-linear; the bank-conflict XOR swizzle (16-B chunk ^= row & 15) is applied to the per-lane SOURCE
//    address and to the ds_read_b128 address.  ONE raw barrier per piece, with no vmcnt drain, so epilogue
//    stores keep flowing behind it;
//  * the frozen weights are pre-packed once (neat_pack_weight_bf16) in MFMA-fragment order, so a wave
//    loads each W fragment as one contiguous, fully coalesced 1-KiB global read (L2-resident: |W| <= 2 MB)
//    straight into VGPRs -- no LDS traffic, no barrier, reused by the 8 row-tiles of the sub-panel;
//  * W is the MFMA *A* operand and the activations the *B* operand, so an accumulator quad holds 4
//    CONSECUTIVE output columns n of one row m: bias / residual / saved-activation loads and the output
//    stores of the epilogue are 8-16 B per lane, and a wave (64 columns) writes whole 128-B lines;
//  * the LDS-DMA fills are issued from inline asm by a rotating half of the waves (one per SIMD) so that
//    the in-order vmcnt queue of the OTHER wave on each SIMD only ever holds L2-latency weight loads:
//    while a filling wave waits out its HBM-latency fills, its SIMD partner keeps the MFMA pipe busy;
//  * epilogue: each 16 x 32 accumulator tile is transposed through a wave-private LDS slab (aliasing the piece
//    consumed last) so that every lane moves 16 B per global access (bias, residual, saved activation,
//    output) and a row's bytes are contiguous.
//
// Rows >= M are clamped on load and masked on store; N needs % 16, K needs % 128.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace neat {

constexpr int P_THREADS = 512;
constexpr int P_WAVES = 8;                // per workgroup; two workgroups per CU -> 4 waves per SIMD
constexpr int P_NI = 2;                   // 16-column tiles per wave: a unit is 128 rows x 256 columns
constexpr int P_BKH = 128;                // k extent of one LDS piece
// LDS: two [128 x BKH] bf16 pieces = 64 KiB, so that TWO workgroups fit a CU's 160 KiB (with separate epilogue
// slabs a workgroup needed exactly 80 KiB and only one was ever resident); the per-wave epilogue slabs
// (16 rows x NI*16 fp32) alias the piece that was consumed last
constexpr int lds_piece_bytes(int bkh) { return 128 * bkh * 2; }
constexpr int lds_total_bytes(int bkh, int ni) { return 2 * lds_piece_bytes(bkh); }   // epilogue slabs alias a piece

// one LDS-DMA wave-instruction: 64 lanes x 16 B -> lds_dst .. lds_dst+1023 (lane-linear)
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

// One LDS piece: ksteps x (NI weight fragments from global, 8 activation fragments from LDS, NI*8 MFMAs),
// straight-line (no per-tile branches: they would serialise every ds_read behind the previous MFMAs).
template <int NI, int BKH>
__device__ __forceinline__ void piece_mma(f32x4 (&acc)[NI][8], const uint4* __restrict__ wp, const long (&woff)[NI],
                                          const char* abase, int fr, int fq) {
    constexpr int ksteps = BKH >> 5;
    constexpr int rowbytes = (BKH >> 3) * 16;
    uint4 wn[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wn[i] = wp[woff[i]];
#pragma unroll
    for (int kk = 0; kk < ksteps; ++kk) {
        bf16x8 wf[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = __builtin_bit_cast(bf16x8, wn[i]);
        if (kk + 1 < ksteps) {
#pragma unroll
            for (int i = 0; i < NI; ++i) wn[i] = wp[woff[i] + (kk + 1) * 64];
        }
        const int coff = ((kk * 4 + fq) ^ fr) * 16;
#pragma unroll
        for (int tg = 0; tg < 8; tg += 4) {      // 4 row tiles at a time keeps the fragment registers at 16
            bf16x8 af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[t] = *reinterpret_cast<const bf16x8*>(abase + (tg + t) * 16 * rowbytes + coff);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    acc[i][tg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[t], acc[i][tg + t], 0, 0, 0);
        }
    }
}

// Work unit = (128-row sub-panel, column block of 8 waves x NI x 16 columns); units are dealt round-robin to
// the persistent workgroups.  Each unit streams its sub-panel through LDS in K/BKH pieces.
template <int EPI, int BKH, int NI, int OCC>
__global__ __launch_bounds__(P_THREADS, 2 * OCC) void gemm_panel_kernel(
    const bf16_t* __restrict__ A, const uint4* __restrict__ Wp, const float* __restrict__ bias, const float* res,
    const bf16_t* __restrict__ aux, void* out0, void* out1, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ksteps = BKH >> 5;                 // MFMA k-steps per piece (4 or 8)
    constexpr int cpr = BKH >> 3;                    // 16-B chunks per LDS row (16 or 32)
    constexpr int rpi = 64 / cpr;                    // rows per LDS-DMA wave-instruction (4 or 2)
    constexpr int ninstr = 128 * cpr / 64;           // wave-instructions per piece
    constexpr int CB = P_WAVES * NI * 16;            // columns per unit
    constexpr int P_BUF_BYTES = lds_piece_bytes(BKH);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;  // LDS byte address

    // Unit -> workgroup deal, XCD-aware.  Workgroups are placed on the 8 XCDs round-robin by blockIdx (used for
    // speed only, never for correctness), and each XCD has its own L2.  All column blocks of one 128-row
    // sub-panel are therefore dealt to workgroups of ONE XCD (sub-panel s belongs to XCD s % 8): the panel is
    // fetched from the fabric once instead of once per column block.
    const int nsub = (M + 127) >> 7;
    const int ncb = (N + CB - 1) / CB;
    const int G = gridDim.x;
    const int X = (G & 7) == 0 ? 8 : 1;              // XCDs the deal is spread over
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G / X;
    const int nsx = (nsub - xcd + X - 1) / X;        // sub-panels xcd, xcd+X, ... owned by this XCD
    const int myunits = (nsx * ncb - wg + WX - 1) / WX;           // local units wg, wg+WX, ...
    if (myunits <= 0) return;                                     // uniform
    auto unit_of = [&](int t, int& sub, int& cb) {                // t-th unit of this workgroup
        const int v = wg + t * WX;
        const int i = v / ncb;
        cb = v - i * ncb;
        sub = xcd + X * i;
    };
    const int nkh = K / BKH;                         // pieces along K
    const int ksK = K >> 5;                          // k-steps in the whole K (packed-W stride)
    const int J = myunits * nkh;                     // pieces this workgroup consumes

    auto issue_piece = [&](int j) {
        const int t = j / nkh;
        const int kh = j - t * nkh;
        int sub_, cb_;
        unit_of(t, sub_, cb_);
        const int row0 = sub_ << 7;
        const uint32_t buf = lds_base + (j & 1) * P_BUF_BYTES;
        const bf16_t* src = A + kh * BKH;
        for (int q = (wave & 3); q < ninstr; q += 4) {
            const int row = q * rpi + lane / cpr;                  // row inside the piece
            const int c = (lane % cpr) ^ (row & 15);
            int gr = row0 + row;
            gr = gr < M ? gr : M - 1;
            glds16(src + (long)gr * K + c * 8, buf + q * 1024);
        }
    };

    if ((wave >> 2) == 0) issue_piece(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f32x4 acc[NI][8];
    for (int j = 0; j < J; ++j) {
        const int tu = j / nkh;
        const int kh = j - tu * nkh;
        int sub, cb;
        unit_of(tu, sub, cb);

        // piece j is in LDS (its fillers waited vmcnt(0) before arriving here) and nobody reads buffer
        // (j+1)&1 any more.  Raw barrier: no vmcnt(0), so epilogue stores keep draining behind it.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (j + 1 < J && (wave >> 2) == ((j + 1) & 1)) issue_piece(j + 1);

        const int nt0 = (cb * CB >> 4) + wave * NI;                 // first of this wave's NI column tiles
        const int nvalid = (N >> 4) - nt0;                          // how many of them exist (<=0: idle)
        if (kh == 0) {
            // the accumulators start from the bias (lane: columns tile*16 + 4*fq .. +3 of every row tile), which
            // takes the bias add out of the epilogue
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_BIAS_GELU) {
                    if (i < nvalid) b0 = *reinterpret_cast<const f32x4*>(bias + (nt0 + i) * 16 + fq * 4);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[i][t] = b0;
            }
        }
        if (nvalid > 0) {
            const char* buf = lds + (j & 1) * P_BUF_BYTES;
            const uint4* wp = Wp + ((long)nt0 * ksK + (long)kh * ksteps) * 64 + lane;
            long woff[NI];                                          // missing column tiles re-read the last valid one
#pragma unroll
            for (int i = 0; i < NI; ++i) woff[i] = (long)(i < nvalid ? i : nvalid - 1) * ksK * 64;
            piece_mma<NI, BKH>(acc, wp, woff, buf + fr * (cpr * 16), fr, fq);
        }
        // the fills this wave issued for piece j+1 have had the whole k-loop to land; retire them BEFORE the
        // epilogue stores so that the next barrier needs no vmcnt wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kh == nkh - 1) {                                        // uniform over the workgroup
            // every wave is done reading piece j: its buffer now holds the epilogue slabs, until the fill of
            // piece j+2 is issued behind the NEXT barrier (which each wave reaches after its epilogue)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (nvalid > 0) {
                // Epilogue.  The accumulator holds 4 consecutive columns per lane but 16 different ROWS per
                // wave-instruction (32-B requests).  Each 16 x (NI*16) tile is therefore transposed through a
                // wave-private LDS slab (XOR-swizzled 16-B chunks, conflict-free both ways) so that consecutive
                // lanes cover one row and every lane moves 16 B per global access.
                // two slabs per wave (even / odd chunk), so that the transposing write of chunk t+1 does not wait
                // for the reads of chunk t
                char* stg = lds + (j & 1) * P_BUF_BYTES + wave * (2 * NI * 1024);
                constexpr int SLAB2 = NI * 1024;
                constexpr int CPW = NI * 4;                             // 16-B fp32 chunks per row of the slab
                const int ncols = nvalid < NI ? nvalid * 16 : NI * 16;  // valid columns of this wave's slab
                const int rowm0 = sub << 7;
                // FULL = whole unit inside the matrix: straight-line code without per-chunk conditions, which lets
                // the compiler run the LDS round trip of chunk t+1 under the arithmetic and stores of chunk t
                auto epilogue = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    if constexpr (EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_F32) {
                        // fp32 outputs: a lane owns 4 columns (16 B), 8 lanes cover a 128-B row segment
                        constexpr int RPI = 64 / CPW;                   // rows per read instruction
                        const int cl = lane % CPW, rq = lane / CPW;
                        const int n = nt0 * 16 + cl * 4;
                        const bool col_ok = cl * 4 < ncols;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            char* sl = stg + (t & 1) * SLAB2;
#pragma unroll
                            for (int i = 0; i < NI; ++i)
                                *reinterpret_cast<f32x4*>(sl + fr * (CPW * 16) + (((i * 4 + fq) ^ (fr & (CPW - 1))) << 4)) = acc[i][t];
#pragma unroll
                            for (int r = 0; r < 16 / RPI; ++r) {
                                const int row = r * RPI + rq;
                                const float4 a4 = *reinterpret_cast<const float4*>(sl + row * (CPW * 16) + ((cl ^ (row & (CPW - 1))) << 4));
                                const int m = rowm0 + t * 16 + row;
                                if constexpr (!FULL) {
                                    if (m >= M || !col_ok) continue;
                                }
                                const long o = (long)m * N + n;
                                if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                                    const float4 rr = *reinterpret_cast<const float4*>(res + o);
                                    *reinterpret_cast<float4*>((float*)out0 + o) = make_float4(a4.x + rr.x, a4.y + rr.y, a4.z + rr.z, a4.w + rr.w);
                                } else {
                                    *reinterpret_cast<float4*>((float*)out0 + o) = a4;
                                }
                            }
                        }
                    } else {
                        // bf16 outputs: a lane owns 8 columns (16 B of bf16), 4 lanes cover the wave's 32 columns,
                        // one wave-instruction covers the whole 16-row tile.  Slab swizzle sw(r) is conflict-free
                        // for the ds_write_b128 lane groups (8 contiguous lanes) and the ds_read_b128 ones.
                        static_assert(NI == 2, "bf16 epilogue assumes a 32-column slab");
                        auto sw = [](int r) { return (r & 3) ^ ((r & 4) ? 5 : 0); };
                        const int er = lane >> 2, eq = lane & 3;
                        const int n = nt0 * 16 + eq * 8;
                        const bool col_ok = eq * 8 < ncols;
                        const char* rd0 = stg + er * 128 + (((2 * eq) ^ sw(er)) << 4);
                        const char* rd1 = stg + er * 128 + (((2 * eq + 1) ^ sw(er)) << 4);
                        char* wr = stg + fr * 128;
                        const int wsw = sw(fr);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
#pragma unroll
                            for (int i = 0; i < NI; ++i)
                                *reinterpret_cast<f32x4*>(wr + (t & 1) * SLAB2 + (((i * 4 + fq) ^ wsw) << 4)) = acc[i][t];
                            const float4 a0 = *reinterpret_cast<const float4*>(rd0 + (t & 1) * SLAB2);
                            const float4 a1 = *reinterpret_cast<const float4*>(rd1 + (t & 1) * SLAB2);
                            const int m = rowm0 + t * 16 + er;
                            if constexpr (!FULL) {
                                if (m >= M || !col_ok) continue;
                            }
                            const long o = (long)m * N + n;
                            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                            if constexpr (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16) {
                                *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(
                                    pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                            } else if constexpr (EPI == NEAT_EPI_BIAS_GELU) {
                                // the saved pre-activation is the bf16-rounded value; GELU is applied to the same
                                // rounded value so that forward and backward see one h
                                uint32_t hp[4], up[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    hp[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                                    const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                                    up[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
                                }
                                if (FULL || out0) *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                                *reinterpret_cast<uint4*>((bf16_t*)out1 + o) = make_uint4(up[0], up[1], up[2], up[3]);
                            } else {  // NEAT_EPI_GELU_BWD
                                const uint4 hq = *reinterpret_cast<const uint4*>(aux + o);
                                const uint32_t hp[4] = {hq.x, hq.y, hq.z, hq.w};
                                uint32_t dp[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                                    dp[q] = pack2bf(v[2 * q] * quick_gelu_grad(h0), v[2 * q + 1] * quick_gelu_grad(h1));
                                }
                                *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(dp[0], dp[1], dp[2], dp[3]);
                            }
                        }
                    }
                };
                const bool full = rowm0 + 128 <= M && ncols == NI * 16 && (EPI != NEAT_EPI_BIAS_GELU || out0 != nullptr);
                if (full) epilogue(std::true_type{});
                else epilogue(std::false_type{});
            }
        }
    }
}

// f32 weights -> bf16 in MFMA-fragment order.  Logical matrix Wl[n][k] (N x K): Wl = src (row-major [N,K])
// or, with transposed != 0, Wl[n][k] = src[k*N + n] (src row-major [K,N]: the dgrad operand W^T).
// dst[((nt*(K/32) + ks)*64 + lane)*8 + j] = Wl[nt*16 + (lane&15)][ks*32 + (lane>>4)*8 + j]
__global__ void pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                   int transposed) {
    const long total = (long)(N / 16) * (K / 32) * 64;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(u & 63);
        const long tile = u >> 6;
        const int ks = (int)(tile % (K / 32));
        const int nt = (int)(tile / (K / 32));
        const int n = nt * 16 + (lane & 15);
        const int k0 = ks * 32 + (lane >> 4) * 8;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 2 * j;
            const float a = transposed ? src[(long)k * N + n] : src[(long)n * K + k];
            const float b = transposed ? src[(long)(k + 1) * N + n] : src[(long)n * K + k + 1];
            w[j] = pack2bf(a, b);
        }
        reinterpret_cast<uint4*>(dst)[u] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

static int g_num_cu = 0;
static int num_cus() {
    if (g_num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_num_cu = n;
    }
    return g_num_cu;
}

template <int EPI, int BKH, int NI, int OCC>
static int launch_gemm_c(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                         const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int grid, hipStream_t s) {
    constexpr int ldsb = lds_total_bytes(BKH, NI);
    static bool attr_set = false;  // per-template-instance: allow > 64 KiB dynamic LDS
    if (!attr_set) {
        int rc = check_hip(hipFuncSetAttribute((const void*)gemm_panel_kernel<EPI, BKH, NI, OCC>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ldsb),
                           "hipFuncSetAttribute(gemm)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_panel_kernel<EPI, BKH, NI, OCC>), dim3(grid), dim3(P_THREADS), ldsb, s, A,
                       reinterpret_cast<const uint4*>(Wp), bias, res, aux, out0, out1, M, N, K);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16");
    return NEAT_OK;
}

template <int EPI>
static int launch_gemm(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                       const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s) {
    const int cus = num_cus();
    const long units = (long)cdiv(M, 128) * cdiv(N, P_WAVES * P_NI * 16);
    // persistent: two workgroups per CU; a multiple of 8 so that the XCD-aware deal applies
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < 2 * cus ? want : 2 * cus);
    return launch_gemm_c<EPI, P_BKH, P_NI, 2>(A, Wp, bias, res, aux, out0, out1, M, N, K, grid, s);
}

}  // namespace neat

using namespace neat;

extern "C" int neat_pack_weight_bf16(const float* src, uint16_t* dst, int N, int K, int transposed, void* stream) {
    NEAT_REQUIRE(src && dst, NEAT_ERR_ARG, "neat_pack_weight_bf16: null pointer");
    NEAT_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, NEAT_ERR_SHAPE,
                 "neat_pack_weight_bf16: N=%d K=%d (need N %% 16 == 0, K %% 32 == 0)", N, K);
    const long total = (long)(N / 16) * (K / 32) * 64;
    long g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                       K, transposed);
    NEAT_LAUNCH_CHECK("neat_pack_weight_bf16");
    return NEAT_OK;
}

extern "C" int neat_gemm_bf16(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                              const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                              void* stream) {
    NEAT_REQUIRE(A && W, NEAT_ERR_ARG, "neat_gemm_bf16: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0 && N % 16 == 0, NEAT_ERR_SHAPE,
                 "neat_gemm_bf16: M=%d N=%d K=%d unsupported (need K %% 128 == 0, N %% 16 == 0)", M, N, K);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case NEAT_EPI_BF16:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16: out0 is null");
            return launch_gemm<NEAT_EPI_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_BF16:
            NEAT_REQUIRE(out0 && bias, NEAT_ERR_ARG, "neat_gemm_bf16: BIAS_BF16 needs bias, out0");
            return launch_gemm<NEAT_EPI_BIAS_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_RES_F32:
            NEAT_REQUIRE(out0 && bias && res, NEAT_ERR_ARG, "neat_gemm_bf16: BIAS_RES_F32 needs bias, res, out0");
            return launch_gemm<NEAT_EPI_BIAS_RES_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_GELU:
            NEAT_REQUIRE(out1 && bias, NEAT_ERR_ARG, "neat_gemm_bf16: BIAS_GELU needs bias, out1");
            return launch_gemm<NEAT_EPI_BIAS_GELU>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_GELU_BWD:
            NEAT_REQUIRE(out0 && aux, NEAT_ERR_ARG, "neat_gemm_bf16: GELU_BWD needs aux, out0");
            return launch_gemm<NEAT_EPI_GELU_BWD>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_F32:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16: out0 is null");
            return launch_gemm<NEAT_EPI_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        default:
            return set_error(NEAT_ERR_ARG, "neat_gemm_bf16: unknown epilogue %d", epilogue);
    }
}
