// Residual GEMM with the FOLLOWING LayerNorm folded into its epilogue (round-3 prototype of the byte-cutting fusion).
// Reference: x = x + out_proj(attn) followed by ln_2(x), and x = x + c_proj(mlp) followed by the next block's ln_1(x)
// (open_clip/transformer.py:241-242, 15-30).
//
//   x_out[M,N] (f32) = A[M,K] . W[N,K]^T + bias + res          y[M,N] (bf16) = LayerNorm(x_out) * gamma + beta
//   mean[M], rstd[M] (f32) = the row statistics the backward pass needs
//
// Unfused, LayerNorm is its own launch that re-reads x_out from HBM (4 B per element) to write y (2 B).  Folding it in needs
// a whole ROW in one workgroup at epilogue time, so the unit is 64 rows x N columns (N = 512: 8 waves x 64 columns,
// accumulator 4 column tiles x 4 row tiles = 64 VGPRs -- the same 32 Ki outputs per unit as the panel kernel's 128 x 256):
//   * the 64-row sub-panel streams through LDS in [64 x 128-k] pieces (16 KiB, double buffered) filled by LDS-DMA by a
//     rotating half of the waves, exactly as in gemm_panel_kernel; weight fragments come straight from L2 in the packed
//     MFMA-fragment layout of neat_pack_weight_bf16 (a fragment now serves 4 row tiles instead of 8, and every A fragment
//     read from LDS serves 4 MFMAs instead of 2);
//   * epilogue: the lane adds its residual values to the accumulators (a lane holds 4 consecutive columns of one row per
//     tile), stores x_out, and the row statistics are reduced in two rounds (mean, then the centred sum of squares -- the
//     two-pass form of csrc/ln.hip) across the 4 lanes that share a row by shuffles and across the 8 waves through LDS.
// N is fixed at 512 here (ViT-B text towers, the benchmark configuration); other widths take the two launches.
//
// The same 64 x 512 structure WITHOUT the LayerNorm (MODE W_RES: fp32 residual epilogue; W_BF16: plain bf16 output through
// wave-private LDS slabs, whole 128-B lines) serves the N = 512 GEMMs of launches that put at most one 128 x 256 unit on a
// CU -- the packed text batches, the MCQ scorer (neat_gemm_bf16 sends them here, gemm_wide512_applies).  Every A fragment
// read from LDS feeds four MFMAs instead of two (the panel kernel's 32-column wave tile keeps the LDS pipe as busy as the
// matrix pipe), but every weight fragment is read twice as often: the shape wins only while W is L2-resident (one GEMM
// repeated: out-projection 35.7 vs 40.8 us, c_proj 64.2 vs 81.6 us at 14 848 rows, tools/rowln_bench.py).  Inside the step
// that is what the weight warm-up at kernel start buys (see gemm_panel_kernel): with it the packed step gains 2.9 %
// (15.08 -> 14.65 ms per batch of 256, tools/experiments/wide_ab.sh); without it the same dispatch was neutral.  At 78 848
// rows the panel kernel's longer weight reuse wins (104.8 vs 131.6 us) and keeps those launches.
// Those launches run the 16-wave split-K form of the unit (gemm_wide_split_kernel below) when K % 256 == 0, K >= 512.
#include <stdlib.h>

#include "common.h"

namespace neat {

constexpr int R_THREADS = 512;
constexpr int R_N = 512;                    // row width this kernel is built for
constexpr int R_RT = 4;                     // row tiles per unit (64 rows)
constexpr int R_NI = 4;                     // column tiles per wave (64 columns)
constexpr int R_BKH = 128;                  // k extent of an LDS piece
constexpr int R_PIECE = 64 * R_BKH * 2;     // 16 KiB
constexpr int R_RED = 2 * R_PIECE;          // reduction scratch behind the two pieces: [2][8 waves][64 rows] floats
constexpr int R_SLAB = R_RED + 2 * 8 * 64 * 4;   // eight wave-private 2-KiB slabs (bf16 epilogue: 16 rows x 128 B)
constexpr int R_LDS = R_SLAB + 8 * 2048;
enum { W_LN = 0, W_RES = 1, W_BF16 = 2 };

__device__ __forceinline__ void rglds16(const void* gsrc, uint32_t lds_dst) {
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

template <int MODE>
__global__ __launch_bounds__(R_THREADS, 4) void gemm_rowln_kernel(
    const bf16_t* __restrict__ A, const uint4* __restrict__ Wp, const float* __restrict__ bias, const float* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ xout, bf16_t* __restrict__ y,
    float* __restrict__ mean, float* __restrict__ rstd, int M, int K, float eps, int warm) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int N = R_N;
    constexpr int ksteps = R_BKH >> 5;               // 4
    constexpr int cpr = R_BKH >> 3;                  // 16-B chunks per LDS row (16)
    constexpr int rpi = 64 / cpr;                    // rows per LDS-DMA wave-instruction (4)
    constexpr int ninstr = 64 * cpr / 64;            // wave-instructions per piece (16)
    constexpr int rowbytes = cpr * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    float* red = reinterpret_cast<float*>(lds + R_RED);

    const int nsub = (M + 63) >> 6;
    const int G = gridDim.x;
    const int myunits = (nsub - (int)blockIdx.x + G - 1) / G;       // units blockIdx, blockIdx + G, ...
    if (myunits <= 0) return;
    const int nkh = K / R_BKH;
    const int ksK = K >> 5;
    const int J = myunits * nkh;

    auto issue_piece = [&](int j) {
        const int t = j / nkh;
        const int kh = j - t * nkh;
        const int row0 = ((int)blockIdx.x + t * G) << 6;
        const uint32_t buf = lds_base + (j & 1) * R_PIECE;
        const bf16_t* src = A + kh * R_BKH;
        for (int q = (wave & 3); q < ninstr; q += 4) {
            const int row = q * rpi + lane / cpr;
            const int c = (lane % cpr) ^ (row & 15);
            int gr = row0 + row;
            gr = gr < M ? gr : M - 1;
            rglds16(src + (long)gr * K + c * 8, buf + q * 1024);
        }
    };

    // W warm-up (see gemm_panel_kernel): the workgroups of an XCD touch the whole packed weight between them
    typedef unsigned warm_u32x4 __attribute__((ext_vector_type(4)));
    warm_u32x4 warmv = {0u, 0u, 0u, 0u};
    if (warm) {
        const int xcd = blockIdx.x & 7;                             // round-robin placement (speed only)
        const long wg = blockIdx.x >> 3, WX = (G - xcd + 7) >> 3;   // this workgroup among the WX of its XCD
        const long wbytes = (long)N * K * 2;
        const char* wbase = reinterpret_cast<const char*>(Wp);
        for (long off = (wg * R_THREADS + threadIdx.x) * 16; off < wbytes; off += WX * R_THREADS * 16)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(warmv) : "v"(wbase + off) : "memory");
    }
    if ((wave >> 2) == 0) issue_piece(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("; warm-up retired" :: "v"(warmv));

    const int nt0 = wave * R_NI;                                     // first of this wave's 4 column tiles
    f32x4 acc[R_NI][R_RT];
    for (int j = 0; j < J; ++j) {
        const int tu = j / nkh;
        const int kh = j - tu * nkh;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (j + 1 < J && (wave >> 2) == ((j + 1) & 1)) issue_piece(j + 1);
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < R_NI; ++i) {
                f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE != W_BF16) b0 = *reinterpret_cast<const f32x4*>(bias + (nt0 + i) * 16 + fq * 4);
#pragma unroll
                for (int t = 0; t < R_RT; ++t) acc[i][t] = b0;
            }
        }
        {
            const char* abase = lds + (j & 1) * R_PIECE + fr * rowbytes;
            const uint4* wp = Wp + ((long)nt0 * ksK + (long)kh * ksteps) * 64 + lane;
            uint4 wn[R_NI];
#pragma unroll
            for (int i = 0; i < R_NI; ++i) wn[i] = wp[(long)i * ksK * 64];
#pragma unroll
            for (int kk = 0; kk < ksteps; ++kk) {
                bf16x8 wf[R_NI];
#pragma unroll
                for (int i = 0; i < R_NI; ++i) wf[i] = __builtin_bit_cast(bf16x8, wn[i]);
                if (kk + 1 < ksteps) {
#pragma unroll
                    for (int i = 0; i < R_NI; ++i) wn[i] = wp[(long)i * ksK * 64 + (kk + 1) * 64];
                }
                const int coff = ((kk * 4 + fq) ^ fr) * 16;
                bf16x8 af[R_RT];
#pragma unroll
                for (int t = 0; t < R_RT; ++t) af[t] = *reinterpret_cast<const bf16x8*>(abase + t * 16 * rowbytes + coff);
#pragma unroll
                for (int t = 0; t < R_RT; ++t)
#pragma unroll
                    for (int i = 0; i < R_NI; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[t], acc[i][t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kh == nkh - 1) {
            const int row0 = ((int)blockIdx.x + tu * G) << 6;
            if constexpr (MODE == W_BF16) {
                // bf16 output: the wave's 16 x 64 tile of a row tile is 16 rows x 128 B; through the wave-private slab it
                // leaves as whole lines (8 rows x 128 B per instruction).  16-B chunks XOR-swizzled by the row.
                char* slab = lds + R_SLAB + wave * 2048;
                bf16_t* out = y;
                const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
                for (int t = 0; t < R_RT; ++t) {
#pragma unroll
                    for (int i = 0; i < R_NI; ++i) {
                        const f32x4 v = acc[i][t];
                        // lane's 8 bytes: columns 16 i + 4 fq .. + 3 -> byte 32 i + 8 fq of the row = chunk 2 i + (fq >> 1)
                        char* p = slab + fr * 128 + (((2 * i + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8;
                        *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = rr + 8 * h;
                        const uint4 q = *reinterpret_cast<const uint4*>(slab + r * 128 + ((cc ^ (r & 7)) << 4));
                        const int m = row0 + t * 16 + r;
                        if (m < M) *reinterpret_cast<uint4*>(out + (long)m * N + wave * 64 + cc * 8) = q;
                    }
                }
                continue;
            }
            // ---- epilogue: x = acc + res, row statistics over all 512 columns, y = LN(x) ----
            // lane: rows row0 + 16 t + fr, columns 64 wave + 16 i + 4 fq .. + 3
            float s1[R_RT];
#pragma unroll
            for (int t = 0; t < R_RT; ++t) {
                int m = row0 + t * 16 + fr;
                const bool ok = m < M;
                m = ok ? m : M - 1;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < R_NI; ++i) {
                    const long o = (long)m * N + wave * 64 + i * 16 + fq * 4;
                    const float4 r = *reinterpret_cast<const float4*>(res + o);
                    f32x4 v = acc[i][t];
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                    acc[i][t] = v;
                    if (ok) *reinterpret_cast<float4*>(xout + o) = make_float4(v[0], v[1], v[2], v[3]);
                    s += (v[0] + v[1]) + (v[2] + v[3]);
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                s1[t] = s;
            }
            if constexpr (MODE == W_RES) continue;            // residual GEMM without the LayerNorm: done
            // the reduction scratch was last read two barriers ago (previous unit's second round): free
            if (fq == 0) {
#pragma unroll
                for (int t = 0; t < R_RT; ++t) red[wave * 64 + t * 16 + fr] = s1[t];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float mu[R_RT], s2[R_RT];
#pragma unroll
            for (int t = 0; t < R_RT; ++t) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += red[w * 64 + t * 16 + fr];
                mu[t] = s * (1.0f / N);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < R_NI; ++i) {
                    f32x4 v = acc[i][t];
                    v[0] -= mu[t]; v[1] -= mu[t]; v[2] -= mu[t]; v[3] -= mu[t];
                    acc[i][t] = v;
                    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                s2[t] = q;
            }
            if (fq == 0) {
#pragma unroll
                for (int t = 0; t < R_RT; ++t) red[512 + wave * 64 + t * 16 + fr] = s2[t];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            f32x4 g4[R_NI], b4[R_NI];
#pragma unroll
            for (int i = 0; i < R_NI; ++i) {
                g4[i] = *reinterpret_cast<const f32x4*>(gamma + wave * 64 + i * 16 + fq * 4);
                b4[i] = *reinterpret_cast<const f32x4*>(beta + wave * 64 + i * 16 + fq * 4);
            }
#pragma unroll
            for (int t = 0; t < R_RT; ++t) {
                float q = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) q += red[512 + w * 64 + t * 16 + fr];
                const float rs = rsqrtf(q * (1.0f / N) + eps);
                const int m = row0 + t * 16 + fr;
                if (m < M) {
#pragma unroll
                    for (int i = 0; i < R_NI; ++i) {
                        const f32x4 v = acc[i][t];
                        const float o0 = v[0] * rs * g4[i][0] + b4[i][0], o1 = v[1] * rs * g4[i][1] + b4[i][1];
                        const float o2 = v[2] * rs * g4[i][2] + b4[i][2], o3 = v[3] * rs * g4[i][3] + b4[i][3];
                        *reinterpret_cast<uint2*>(y + (long)m * N + wave * 64 + i * 16 + fq * 4) =
                            make_uint2(pack2bf(o0, o1), pack2bf(o2, o3));
                    }
                    if (wave == 0 && fq == 0) {
                        mean[m] = mu[t];
                        rstd[m] = rs;
                    }
                }
            }
            // the next unit's first reduction write happens behind at least the two barriers of its first two pieces
            // (nkh >= 2 is required below), i.e. after every wave has finished the reads above
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the same 64 x 512 unit with the K loop cut in two INSIDE the workgroup -------------------------------------------
// The launches this file serves put at most ONE unit on a CU (packed text batches, the MCQ scorer): eight waves, two per
// SIMD, and a main loop that waits out L2 latency with nothing else to run.  Here a workgroup has SIXTEEN waves: waves
// 0..7 (k-half 0) run k in [0, K/2), waves 8..15 (k-half 1) run [K/2, K), each half with its own double-buffered LDS
// pieces, the same column split (wave & 7 owns 64 columns) and the same per-wave code as gemm_rowln_kernel -- four waves
// per SIMD instead of two.  At the end of a unit the halves are added through LDS in two rounds (a wave sends two of its
// four row tiles each way, 64 KiB per round): k-half 0 ends up with the finished row tiles 0..1, k-half 1 with 2..3, both
// as (bias + half 0) + half 1 (fp32 addition commutes, so the two directions agree bit for bit; the result differs from
// the unsplit kernels' single k chain in the last bit), and each half runs the epilogue of its two row tiles.  No
// workspace, no second launch, no cross-workgroup synchronisation.  K % 256 == 0, K >= 512.
constexpr int S_THREADS = 1024;
constexpr int S_EXCH = 4 * R_PIECE;             // 64 KiB of pieces (2 k-halves x 2 buffers), then the exchange region
constexpr int S_LDS = S_EXCH + 2 * 32768;       // 128 KiB; the bf16 epilogue's wave-private slabs alias the exchange region

template <int MODE>
__global__ __launch_bounds__(S_THREADS) void gemm_wide_split_kernel(
    const bf16_t* __restrict__ A, const uint4* __restrict__ Wp, const float* __restrict__ bias, const float* res,
    float* xout, bf16_t* __restrict__ y, int M, int K, int warm) {
    static_assert(MODE == W_RES || MODE == W_BF16, "the split kernel has the plain epilogues only");
    static_assert(R_RT == 4 && R_NI == 4, "the exchange below is written for 4 x 4 accumulator tiles");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int N = R_N;
    constexpr int ksteps = R_BKH >> 5;
    constexpr int cpr = R_BKH >> 3;
    constexpr int rpi = 64 / cpr;
    constexpr int ninstr = 64 * cpr / 64;
    constexpr int rowbytes = cpr * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wave >> 3, w = wave & 7;                          // k-half, column group
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    const int nsub = (M + 63) >> 6;
    const int G = gridDim.x;
    const int myunits = (nsub - (int)blockIdx.x + G - 1) / G;
    if (myunits <= 0) return;
    const int nkh = (K >> 1) / R_BKH;                                // pieces per k-half
    const int ksK = K >> 5;
    const int J = myunits * nkh;

    auto issue_piece = [&](int j) {
        const int t = j / nkh;
        const int kh = j - t * nkh;
        const int row0 = ((int)blockIdx.x + t * G) << 6;
        const uint32_t buf = lds_base + (g * 2 + (j & 1)) * R_PIECE;
        const bf16_t* src = A + (g * nkh + kh) * R_BKH;
        for (int q = (w & 3); q < ninstr; q += 4) {
            const int row = q * rpi + lane / cpr;
            const int c = (lane % cpr) ^ (row & 15);
            int gr = row0 + row;
            gr = gr < M ? gr : M - 1;
            rglds16(src + (long)gr * K + c * 8, buf + q * 1024);
        }
    };

    typedef unsigned warm_u32x4 __attribute__((ext_vector_type(4)));
    warm_u32x4 warmv = {0u, 0u, 0u, 0u};
    if (warm) {
        const int xcd = blockIdx.x & 7;
        const long wg = blockIdx.x >> 3, WX = (G - xcd + 7) >> 3;
        const long wbytes = (long)N * K * 2;
        const char* wbase = reinterpret_cast<const char*>(Wp);
        for (long off = (wg * S_THREADS + threadIdx.x) * 16; off < wbytes; off += WX * S_THREADS * 16)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(warmv) : "v"(wbase + off) : "memory");
    }
    if ((w >> 2) == 0) issue_piece(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("; warm-up retired" :: "v"(warmv));

    const int nt0 = w * R_NI;
    f32x4 acc[R_NI][R_RT];
    for (int j = 0; j < J; ++j) {
        const int tu = j / nkh;
        const int kh = j - tu * nkh;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (j + 1 < J && (w >> 2) == ((j + 1) & 1)) issue_piece(j + 1);
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < R_NI; ++i) {
                f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE != W_BF16) {
                    if (g == 0) b0 = *reinterpret_cast<const f32x4*>(bias + (nt0 + i) * 16 + fq * 4);
                }
#pragma unroll
                for (int t = 0; t < R_RT; ++t) acc[i][t] = b0;
            }
        }
        {
            const char* abase = lds + (g * 2 + (j & 1)) * R_PIECE + fr * rowbytes;
            const uint4* wp = Wp + ((long)nt0 * ksK + (long)(g * nkh + kh) * ksteps) * 64 + lane;
            uint4 wn[R_NI];
#pragma unroll
            for (int i = 0; i < R_NI; ++i) wn[i] = wp[(long)i * ksK * 64];
#pragma unroll
            for (int kk = 0; kk < ksteps; ++kk) {
                bf16x8 wf[R_NI];
#pragma unroll
                for (int i = 0; i < R_NI; ++i) wf[i] = __builtin_bit_cast(bf16x8, wn[i]);
                if (kk + 1 < ksteps) {
#pragma unroll
                    for (int i = 0; i < R_NI; ++i) wn[i] = wp[(long)i * ksK * 64 + (kk + 1) * 64];
                }
                const int coff = ((kk * 4 + fq) ^ fr) * 16;
                bf16x8 af[R_RT];
#pragma unroll
                for (int t = 0; t < R_RT; ++t) af[t] = *reinterpret_cast<const bf16x8*>(abase + t * 16 * rowbytes + coff);
#pragma unroll
                for (int t = 0; t < R_RT; ++t)
#pragma unroll
                    for (int i = 0; i < R_NI; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[t], acc[i][t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kh != nkh - 1) continue;

        // ---- the two k-halves meet: round r moves row tile 2 + r of half 0 and row tile r of half 1 ----
        // wave (g, w) writes 4 KiB (its 4 column tiles of one row tile, lane-linear) into its group's 32-KiB area and
        // reads the partner wave's (1 - g, w).  The exchange region was last touched by the previous unit's epilogue slabs,
        // at least two piece barriers ago.
        char* mine = lds + S_EXCH + g * 32768 + (w * 4 * 64 + lane) * 16;
        const char* theirs = lds + S_EXCH + (1 - g) * 32768 + (w * 4 * 64 + lane) * 16;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < R_NI; ++i) *reinterpret_cast<f32x4*>(mine + i * 1024) = acc[i][2 + r];
            } else {
#pragma unroll
                for (int i = 0; i < R_NI; ++i) *reinterpret_cast<f32x4*>(mine + i * 1024) = acc[i][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < R_NI; ++i) acc[i][r] += *reinterpret_cast<const f32x4*>(theirs + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < R_NI; ++i) acc[i][2 + r] += *reinterpret_cast<const f32x4*>(theirs + i * 1024);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // read before the next round / the slabs overwrite it
        }
        if (g == 1) {
#pragma unroll
            for (int i = 0; i < R_NI; ++i) {
                acc[i][0] = acc[i][2];
                acc[i][1] = acc[i][3];
            }
        }
        // this wave now holds the finished row tiles 2 g + {0, 1} of its 64 columns in acc[.][0..1]
        const int row0 = (((int)blockIdx.x + tu * G) << 6) + g * 32;
        if constexpr (MODE == W_BF16) {
            char* slab = lds + S_EXCH + wave * 2048;
            const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < R_NI; ++i) {
                    const f32x4 v = acc[i][t];
                    char* p = slab + fr * 128 + (((2 * i + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8;
                    *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rrow = rr + 8 * h;
                    const uint4 q = *reinterpret_cast<const uint4*>(slab + rrow * 128 + ((cc ^ (rrow & 7)) << 4));
                    const int m = row0 + t * 16 + rrow;
                    if (m < M) *reinterpret_cast<uint4*>(y + (long)m * N + w * 64 + cc * 8) = q;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = row0 + t * 16 + fr;
                if (m < M) {
#pragma unroll
                    for (int i = 0; i < R_NI; ++i) {
                        const long o = (long)m * N + w * 64 + i * 16 + fq * 4;
                        const float4 rv = *reinterpret_cast<const float4*>(res + o);
                        const f32x4 v = acc[i][t];
                        *reinterpret_cast<float4*>(xout + o) = make_float4(v[0] + rv.x, v[1] + rv.y, v[2] + rv.z, v[3] + rv.w);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace neat

using namespace neat;

static int wide_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        cus = 256;
    return cus;
}

template <int MODE>
static int launch_wide(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res, const float* gamma,
                       const float* beta, float* x_out, uint16_t* y, float* mean, float* rstd, int M, int K, float eps,
                       hipStream_t s, const char* what) {
    static unsigned long long attr_done = 0;
    if (int rc = ensure_dynamic_lds((const void*)gemm_rowln_kernel<MODE>, R_LDS, &attr_done, "hipFuncSetAttribute(gemm_rowln)"))
        return rc;
    const int cus = wide_cus();
    const int units = cdiv(M, 64);
    const int grid = units < 2 * cus ? units : 2 * cus;
    hipLaunchKernelGGL((gemm_rowln_kernel<MODE>), dim3(grid), dim3(R_THREADS), R_LDS, s, A, reinterpret_cast<const uint4*>(Wp),
                       bias, res, gamma, beta, x_out, reinterpret_cast<bf16_t*>(y), mean, rstd, M, K, eps, gemm_warm_w_enabled());
    NEAT_LAUNCH_CHECK(what);
    return NEAT_OK;
}

// out-projection / c_proj with the following LayerNorm folded in.  A bf16 [M,K]; Wp = neat_pack_weight_bf16 of the [512,K]
// weight; bias, gamma, beta f32 [512]; res f32 [M,512] (must NOT alias x_out); outputs x_out f32 [M,512], y bf16 [M,512],
// mean / rstd f32 [M].  K % 128 == 0, K >= 256.
extern "C" int neat_gemm_bf16_res_ln(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                                     const float* gamma, const float* beta, float* x_out, uint16_t* y, float* mean,
                                     float* rstd, int M, int N, int K, float eps, void* stream) {
    NEAT_REQUIRE(A && Wp && bias && res && gamma && beta && x_out && y && mean && rstd, NEAT_ERR_ARG,
                 "neat_gemm_bf16_res_ln: null pointer");
    NEAT_REQUIRE(res != x_out, NEAT_ERR_ARG, "neat_gemm_bf16_res_ln: res must not alias x_out");
    NEAT_REQUIRE(M > 0 && N == R_N && K >= 2 * R_BKH && K % R_BKH == 0, NEAT_ERR_SHAPE,
                 "neat_gemm_bf16_res_ln: M=%d N=%d K=%d unsupported (N == %d, K %% 128 == 0, K >= 256)", M, N, K, R_N);
    return launch_wide<W_LN>(A, Wp, bias, res, gamma, beta, x_out, y, mean, rstd, M, K, eps, (hipStream_t)stream,
                             "neat_gemm_bf16_res_ln");
}

template <int MODE>
static int launch_wide_split(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res, float* x_out,
                             uint16_t* y, int M, int K, hipStream_t s, const char* what) {
    static unsigned long long attr_done = 0;
    if (int rc = ensure_dynamic_lds((const void*)gemm_wide_split_kernel<MODE>, S_LDS, &attr_done,
                                    "hipFuncSetAttribute(gemm_wide_split)"))
        return rc;
    const int cus = wide_cus();
    const int units = cdiv(M, 64);
    const int grid = units < cus ? units : cus;                      // one 16-wave workgroup per CU
    hipLaunchKernelGGL((gemm_wide_split_kernel<MODE>), dim3(grid), dim3(S_THREADS), S_LDS, s, A,
                       reinterpret_cast<const uint4*>(Wp), bias, res, x_out, reinterpret_cast<bf16_t*>(y), M, K,
                       gemm_warm_w_enabled());
    NEAT_LAUNCH_CHECK(what);
    return NEAT_OK;
}

// neat_gemm_bf16's N = 512 launches of at most one 128 x 256 unit per CU (see the header comment): called from gemm.hip.
namespace neat {
bool gemm_wide512_applies(int M, int N, int K, int epilogue) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("NEAT_GEMM_WIDE512");                // 0: never, 2 / 3: one epilogue only (kernel A/B runs)
        enabled = e ? atoi(e) : 1;
    }
    if (!enabled || N != R_N || K % R_BKH != 0 || K < R_BKH) return false;
    if (epilogue != NEAT_EPI_BIAS_RES_F32 && epilogue != NEAT_EPI_BF16) return false;
    if (enabled == 2 && epilogue != NEAT_EPI_BIAS_RES_F32) return false;      // 2: the residual epilogue only
    if (enabled == 3 && epilogue != NEAT_EPI_BF16) return false;              // 3: the bf16 epilogue only
    static long maxrows = -1;                                       // NEAT_GEMM_WIDE_MAXROWS (experiments): row limit instead of one unit per CU
    if (maxrows < 0) {
        const char* e = getenv("NEAT_GEMM_WIDE_MAXROWS");
        maxrows = e ? atol(e) : 0;
    }
    if (maxrows > 0) return M <= maxrows;
    return (long)cdiv(M, 128) * 2 <= wide_cus();
}

int launch_gemm_wide512(int epilogue, const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res, void* out0,
                        int M, int K, hipStream_t s) {
    static int split = -1;
    if (split < 0) {
        const char* e = getenv("NEAT_GEMM_WIDE_SPLIT");              // 0: the unsplit 8-wave kernel (A/B runs)
        split = e ? atoi(e) : 1;
    }
    if (split && K % (2 * R_BKH) == 0 && K >= 4 * R_BKH && (cdiv(M, 64) <= wide_cus() || split == 2)) {
        if (epilogue == NEAT_EPI_BIAS_RES_F32)
            return launch_wide_split<W_RES>(A, Wp, bias, res, (float*)out0, nullptr, M, K, s, "neat_gemm_bf16(wide split, residual)");
        return launch_wide_split<W_BF16>(A, Wp, nullptr, nullptr, nullptr, (uint16_t*)out0, M, K, s, "neat_gemm_bf16(wide split, bf16)");
    }
    if (epilogue == NEAT_EPI_BIAS_RES_F32)
        return launch_wide<W_RES>(A, Wp, bias, res, nullptr, nullptr, (float*)out0, nullptr, nullptr, nullptr, M, K, 0.f, s,
                                  "neat_gemm_bf16(wide, residual)");
    return launch_wide<W_BF16>(A, Wp, nullptr, nullptr, nullptr, nullptr, nullptr, (uint16_t*)out0, nullptr, nullptr, M, K, 0.f,
                               s, "neat_gemm_bf16(wide, bf16)");
}
}  // namespace neat
