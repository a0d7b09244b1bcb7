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
    const bf16_t* __restrict__ aux, void* out0, void* out1, int M, int N, int K, int warm) {
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

    // W warm-up.  The weights of a launch are not L2-resident when it starts (a layer alternates eight weight matrices and
    // streams its activations through the same 4-MB L2s), and the main loop meets them one 128-k slice at a time: every
    // piece starts with L2 misses that all workgroups of an XCD take together.  So the workgroups of an XCD first touch the
    // WHOLE packed weight between them, one 16-B load per lane and 8 KiB per workgroup and pass (results discarded): one
    // bulk fetch from HBM / the Infinity Cache instead of K / 128 dependent ones.  Retired by the vmcnt(0) below.
    // The loads are asynchronous: their destination registers stay reserved (one live value, tied through every asm and read
    // once after the wait) until the vmcnt(0) has retired them -- a discarded "=v" output would let the compiler hand
    // the registers to something else while the loads are still in flight.
    typedef unsigned warm_u32x4 __attribute__((ext_vector_type(4)));
    warm_u32x4 warmv = {0u, 0u, 0u, 0u};
    if (warm) {
        const long wbytes = (long)N * K * 2;
        const char* wbase = reinterpret_cast<const char*>(Wp);
        for (long off = ((long)wg * P_THREADS + threadIdx.x) * 16; off < wbytes; off += (long)WX * P_THREADS * 16)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(warmv) : "v"(wbase + off) : "memory");
    }
    if ((wave >> 2) == 0) issue_piece(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("; warm-up retired" :: "v"(warmv));

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
                        // The residual rows are loaded RES_AHEAD chunks ahead of their use (whole units only).  Loaded where
                        // they are used, every load sat alone in the in-order vmcnt queue behind the previous chunk's
                        // stores and the wave waited out one memory latency per 8 rows: 16 serial round trips per unit.
                        constexpr int RES_AHEAD = 2;
                        constexpr int NRD = 16 / RPI;
                        constexpr bool PREFETCH = FULL && EPI == NEAT_EPI_BIAS_RES_F32;
                        float4 rres[8][NRD];
                        auto load_res = [&](int t) {
#pragma unroll
                            for (int r = 0; r < NRD; ++r)
                                rres[t][r] = *reinterpret_cast<const float4*>(res + (long)(rowm0 + t * 16 + r * RPI + rq) * N + n);
                        };
                        if constexpr (PREFETCH) {
#pragma unroll
                            for (int t = 0; t < RES_AHEAD; ++t) load_res(t);
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            char* sl = stg + (t & 1) * SLAB2;
#pragma unroll
                            for (int i = 0; i < NI; ++i)
                                *reinterpret_cast<f32x4*>(sl + fr * (CPW * 16) + (((i * 4 + fq) ^ (fr & (CPW - 1))) << 4)) = acc[i][t];
                            if constexpr (PREFETCH) {
                                if (t + RES_AHEAD < 8) load_res(t + RES_AHEAD);
                            }
#pragma unroll
                            for (int r = 0; r < NRD; ++r) {
                                const int row = r * RPI + rq;
                                const float4 a4 = *reinterpret_cast<const float4*>(sl + row * (CPW * 16) + ((cl ^ (row & (CPW - 1))) << 4));
                                const int m = rowm0 + t * 16 + row;
                                if constexpr (!FULL) {
                                    if (m >= M || !col_ok) continue;
                                }
                                const long o = (long)m * N + n;
                                if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                                    float4 rr;
                                    if constexpr (PREFETCH) rr = rres[t][r];
                                    else rr = *reinterpret_cast<const float4*>(res + o);
                                    *reinterpret_cast<float4*>((float*)out0 + o) = make_float4(a4.x + rr.x, a4.y + rr.y, a4.z + rr.z, a4.w + rr.w);
                                } else {
                                    *reinterpret_cast<float4*>((float*)out0 + o) = a4;
                                }
                            }
                        }
                    } else if constexpr (NI == 1) {
                        // bf16 outputs of the 128 x 128 unit (a wave owns 16 columns): a row of the 16 x 16 tile is 32 B of
                        // bf16 = two lanes of 8 columns, so one wave-instruction serves TWO row tiles: lanes 0..31 read
                        // tile t from slab 0, lanes 32..63 tile t+1 from slab 1 (16-B chunks XOR-swizzled by row & 3).
                        const int half = lane >> 5, er = (lane >> 1) & 15, eq = lane & 1;
                        const int n = nt0 * 16 + eq * 8;
                        const char* rd = stg + half * SLAB2 + er * 64;
                        const int c0 = ((2 * eq) ^ (er & 3)) << 4, c1 = ((2 * eq + 1) ^ (er & 3)) << 4;
                        char* wr = stg + fr * 64 + ((fq ^ (fr & 3)) << 4);
                        constexpr bool PREFETCH_H = FULL && EPI == NEAT_EPI_GELU_BWD;
                        uint4 hpre[4];
                        if constexpr (PREFETCH_H) {
#pragma unroll
                            for (int tp = 0; tp < 4; ++tp)
                                hpre[tp] = *reinterpret_cast<const uint4*>(aux + (long)(rowm0 + (2 * tp + half) * 16 + er) * N + n);
                        }
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            *reinterpret_cast<f32x4*>(wr) = acc[0][2 * tp];
                            *reinterpret_cast<f32x4*>(wr + SLAB2) = acc[0][2 * tp + 1];
                            const float4 a0 = *reinterpret_cast<const float4*>(rd + c0);
                            const float4 a1 = *reinterpret_cast<const float4*>(rd + c1);
                            const int m = rowm0 + (2 * tp + half) * 16 + er;
                            if constexpr (!FULL) {
                                if (m >= M) continue;
                            }
                            const long o = (long)m * N + n;
                            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                            if constexpr (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16) {
                                *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(
                                    pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                            } else if constexpr (EPI == NEAT_EPI_BIAS_GELU) {
                                uint32_t hp[4], up[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    hp[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                                    const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                                    up[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
                                }
                                if (out0) *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                                *reinterpret_cast<uint4*>((bf16_t*)out1 + o) = make_uint4(up[0], up[1], up[2], up[3]);
                            } else {  // NEAT_EPI_GELU_BWD
                                uint4 hq;
                                if constexpr (PREFETCH_H) hq = hpre[tp];
                                else hq = *reinterpret_cast<const uint4*>(aux + o);
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
                        // GELU': the saved pre-activation is requested H_AHEAD chunks ahead of its use, for the same reason as the
                        // residual rows above
                        constexpr bool PREFETCH_H = FULL && EPI == NEAT_EPI_GELU_BWD;
                        uint4 hpre[8];
                        constexpr int H_AHEAD = 4;
                        auto load_h = [&](int t) {
                            hpre[t] = *reinterpret_cast<const uint4*>(aux + (long)(rowm0 + t * 16 + er) * N + n);
                        };
                        if constexpr (PREFETCH_H) {
#pragma unroll
                            for (int t = 0; t < H_AHEAD; ++t) load_h(t);
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
#pragma unroll
                            for (int i = 0; i < NI; ++i)
                                *reinterpret_cast<f32x4*>(wr + (t & 1) * SLAB2 + (((i * 4 + fq) ^ wsw) << 4)) = acc[i][t];
                            if constexpr (PREFETCH_H) {
                                if (t + H_AHEAD < 8) load_h(t + H_AHEAD);
                            }
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
                                uint4 hq;
                                if constexpr (PREFETCH_H) hq = hpre[t];
                                else hq = *reinterpret_cast<const uint4*>(aux + o);
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

// NEAT_GEMM_WARM_W=0 switches the weight warm-up of both kernels off (A/B measurements; read once per process)
int gemm_warm_w_enabled() {
    static int warm = -1;
    if (warm < 0) {
        const char* e = getenv("NEAT_GEMM_WARM_W");
        warm = e ? (atoi(e) != 0) : 1;
    }
    return warm;
}

template <int EPI, int BKH, int NI, int OCC>
static int launch_gemm_c(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                         const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int grid, hipStream_t s) {
    constexpr int ldsb = lds_total_bytes(BKH, NI);
    static unsigned long long attr_done = 0;  // per template instance, one bit per device: allow > 64 KiB dynamic LDS
    if (int rc = ensure_dynamic_lds((const void*)gemm_panel_kernel<EPI, BKH, NI, OCC>, ldsb, &attr_done,
                                    "hipFuncSetAttribute(gemm)"))
        return rc;
    const int warm = gemm_warm_w_enabled();
    hipLaunchKernelGGL((gemm_panel_kernel<EPI, BKH, NI, OCC>), dim3(grid), dim3(P_THREADS), ldsb, s, A,
                       reinterpret_cast<const uint4*>(Wp), bias, res, aux, out0, out1, M, N, K, warm);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16");
    return NEAT_OK;
}

// The 128 x 128 unit (NI = 1: a wave owns 16 columns) is taken for the fp32 residual epilogue when the 128 x 256 units
// would fill less than this fraction (percent) of the 2-per-CU workgroup slots.  NEAT_GEMM_NARROW_FILL overrides it
// (0 = never; 1000 + p = for every epilogue at p percent, the form that was measured and not kept).
static int narrow_fill_percent() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NEAT_GEMM_NARROW_FILL");
        v = e ? atoi(e) : 80;
    }
    return v;
}

template <int EPI>
static int launch_gemm(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                       const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s) {
    const int cus = num_cus();
    const long units = (long)cdiv(M, 128) * cdiv(N, P_WAVES * P_NI * 16);
    // N = 512 at no more than one unit per CU (packed text batches, the MCQ scorer): the 64 x 512-unit kernel
    if (gemm_wide512_applies(M, N, K, EPI))
        return launch_gemm_wide512(EPI, A, Wp, bias, res, out0, M, K, s);
    // Small residual GEMMs at N = 768 / 1024 (the MCQ scorer's vision tower at 12 608 rows, packed ViT-L-14 text batches):
    // 128 x 256 units leave the second workgroup slot of most CUs empty (99 x 3 = 297 units on 512 slots), i.e. eight waves
    // per CU.  As 128 x 128 units they are twice as many, sixteen waves on most CUs: MCQ batch 6.05 -> 5.79 ms, ViT-L-14
    // packed step 27.4 -> 26.5 ms.  Only this epilogue: the bf16 ones lose 1-2 us per launch with the narrow unit (their
    // slab serves two row tiles per instruction), which cancelled the gain when every epilogue took it.
    if ((EPI == NEAT_EPI_BIAS_RES_F32 || narrow_fill_percent() >= 1000) &&
        units * 100 < (long)(narrow_fill_percent() % 1000) * 2 * cus && N % 128 == 0) {
        const long units1 = (long)cdiv(M, 128) * cdiv(N, P_WAVES * 16);
        const long want1 = (units1 + 7) / 8 * 8;
        const int grid1 = (int)(want1 < 2 * cus ? want1 : 2 * cus);
        return launch_gemm_c<EPI, P_BKH, 1, 2>(A, Wp, bias, res, aux, out0, out1, M, N, K, grid1, s);
    }
    // persistent: two workgroups per CU; a multiple of 8 so that the XCD-aware deal applies
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < 2 * cus ? want : 2 * cus);
    return launch_gemm_c<EPI, P_BKH, P_NI, 2>(A, Wp, bias, res, aux, out0, out1, M, N, K, grid, s);
}


// =====================================================================================================
// tile256: the second GEMM structure, used whenever N % 256 == 0 and K % 128 == 0 (every linear of the
// ViT-B/32, B/16 and L/14 towers).  It follows the 256 x 256 "8-phase" plan for gfx950:
//
//  * unit = 256 rows x 256 columns, ONE 512-thread workgroup per CU (<= 256 VGPRs), waves as 2 (M) x 4 (N):
//    a wave owns 128 x 64 outputs = 8 x 4 accumulator tiles (128 VGPRs);
//  * BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4): no wave ever waits for a weight load in
//    its in-order vmcnt queue.  K advances in K-tiles of 64; a K-tile is four 16-KiB half-tiles
//    (A0 A1 B0 B1, 128 rows x 64 k each), double buffered = 128 KiB.  The half-tiles are cut by
//    CONSUMPTION order, not by position: A0 = row tiles 0..3 of both wave rows, A1 = row tiles 4..7,
//    B0 / B1 = the first / second 32 columns of every wave column;
//  * a K-tile is four phases (one accumulator quadrant, 16 MFMAs per wave, each): P0 reads B0+A0 (12
//    ds_read_b128), P1 reads B1, P2 reads A1, P3 reads nothing (B0 is still in registers); every phase
//    restages ONE half-tile that was read a phase (or two) earlier with the data of the K-tile after next,
//    so three half-tiles are always in flight; the only vmcnt wait is vmcnt(6) in P3, once per K-tile;
//  * two raw s_barriers per phase, and the two wave rows run ONE barrier apart (wr == 1 takes an extra barrier
//    at the start, wr == 0 at the end): on every SIMD one wave issues its 16 MFMAs while its partner reads
//    fragments and issues LDS-DMA;
//  * the weights are packed once into exactly this LDS image (neat_pack_weight_bf16), so a weight half-tile is
//    one contiguous 16-KiB read; the activation half-tile is read as 128-B row segments with the XOR swizzle
//    applied to the source address;
//  * epilogue straight from the accumulators, no LDS round trip: W is the MFMA A operand, so a lane holds 4
//    consecutive columns of one row; for bf16 outputs the two column tiles of a pair are read from LDS in
//    the interleaved order (8q+4i+j), which makes a lane's 8 values 8 CONSECUTIVE columns = one 16-B store,
//    64 contiguous bytes per row and instruction.  fp32 outputs keep the natural order (16 B per lane);
//  * the bias lives in LDS (fp32 copy made once per workgroup) so that no ordinary VGPR load sits in the
//    vmcnt queue between LDS-DMA operations.
//
// LDS image of a half-tile: row lr (0..127) x 8 chunks of 16 B; chunk c of row lr is stored at chunk
// c ^ swz(lr), swz(lr) = ((lr >> 1) & 7) ^ (((lr >> 4) & 1) << 1): conflict-free for the lane groups of
// ds_read_b128 with both the natural and the interleaved row order.
constexpr int T_THREADS = 512;
constexpr int T_HALF = 16384;                 // bytes of a half-tile
constexpr int T_BUF = 4 * T_HALF;             // one K-tile: A0 A1 B0 B1
constexpr int T_BIAS = 2 * T_BUF;             // fp32 bias copy behind the two K-tile buffers
constexpr int T_MAXN = 4096;                  // fp32 bias copy: 16 KiB
constexpr int T_SLAB = T_BIAS + T_MAXN * 4;   // eight wave-private 2-KiB epilogue slabs
constexpr int T_LDS = T_SLAB + 8 * 2048;      // 160 KiB
constexpr int T_A0 = 0, T_B0 = 2;                // half-tile slots of a K-tile buffer: A0 A1 B0 B1

__host__ __device__ __forceinline__ int t_swz(int lr) { return ((lr >> 1) & 7) ^ (((lr >> 4) & 1) << 1); }

__host__ __device__ constexpr bool t256_shape_ok(int N, int K) { return N % 256 == 0 && K % 128 == 0 && N <= T_MAXN; }

#define T_BARRIER()                                  \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        asm volatile("s_barrier" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

template <int EPI>
__global__ __launch_bounds__(T_THREADS, 2) void gemm_tile256_kernel(
    const bf16_t* __restrict__ A, const char* __restrict__ Wp, const float* __restrict__ bias, const float* res,
    const bf16_t* __restrict__ aux, void* out0, void* out1, int M, int N, int K, int rows0, int warm) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool PERM = !(EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_F32);   // bf16 outputs
    constexpr bool HAS_BIAS = EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_BIAS_GELU;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // unit -> workgroup deal, XCD-aware exactly like gemm_panel_kernel: all column blocks of a 256-row block
    // run on workgroups of one XCD at about the same time, so the block is fetched into that L2 once
    const int nrb = (M + 255) >> 8;
    const int ncb = N >> 8;
    const int G_ = gridDim.x;
    const int X = (G_ & 7) == 0 ? 8 : 1;
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G_ / X;
    const int nrx = (nrb - xcd + X - 1) / X;
    const int U = (nrx * ncb - wg + WX - 1) / WX;                 // units of this workgroup
    if (U <= 0) return;                                           // uniform, before any barrier
    const int nkt = K >> 6;                                       // K-tiles per unit (even: K % 128 == 0)
    const long wcb_bytes = (long)nkt * 2 * T_HALF;                // packed bytes of one column block
    auto unit_of = [&](int t, int& row0, const char*& wb) {
        const int v = wg + t * WX;
        const int i = v / ncb;
        const int cb = v - i * ncb;
        row0 = (xcd + X * i) << 8;
        wb = Wp + cb * wcb_bytes;
    };

    if constexpr (HAS_BIAS) {
        for (int i = threadIdx.x * 4; i < N; i += T_THREADS * 4)
            *reinterpret_cast<float4*>(lds + T_BIAS + i * 4) = *reinterpret_cast<const float4*>(bias + i);
    }

    // ---- staging (LDS-DMA): every wave issues 2 of the 16 wave-instructions of a half-tile ----
    // instruction q of this wave covers half-tile rows lr = (2*wave+q)*8 + (lane>>3), LDS chunk lane&7
    int st_urow[2], st_cb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lr = (2 * wave + q) * 8 + (lane >> 3);
        st_urow[q] = (lr >> 6) * 128 + (lr & 63);                 // unit row of A0's row lr (A1: + 64)
        st_cb[q] = ((lane & 7) ^ t_swz(lr)) * 16;                 // source byte offset inside the 128-B k segment
    }
    const uint32_t st_lds = lds_base + wave * 2048;
    auto stage_a = [&](int h, int pb, int row0, int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int gr = row0 + st_urow[q] + h * 64;
            gr = gr < M ? gr : M - 1;
            const char* src = reinterpret_cast<const char*>(A) + ((long)gr * K + kt * 64) * 2 + st_cb[q];
            glds16(src, st_lds + pb * T_BUF + (T_A0 + h) * T_HALF + q * 1024);
        }
    };
    auto stage_b = [&](int h, int pb, const char* wb, int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char* src = wb + (long)(kt * 2 + h) * T_HALF + (2 * wave + q) * 1024 + lane * 16;
            glds16(src, st_lds + pb * T_BUF + (T_B0 + h) * T_HALF + q * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a half-tile) ----
    int aoff[2][2], boff[2][2];                                   // [row-tile parity][kstep], [column tile][kstep]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int lr = wr * 64 + p * 16 + fr;
            aoff[p][ks] = lr * 128 + (((ks * 4 + fq) ^ t_swz(lr)) << 4);
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int lr = PERM ? wc * 32 + 8 * (fr >> 2) + 4 * ii + (fr & 3) : wc * 32 + ii * 16 + fr;
            boff[ii][ks] = lr * 128 + (((ks * 4 + fq) ^ t_swz(lr)) << 4);
        }
    }

    // ---- stream cursors (wave-uniform): s1 = K-tile g+1 (its A1), s2 = K-tile g+2 (A0, B0, B1) ----
    int c_t = 0, c_kt = 0, c_row0;
    const char* c_wb;
    unit_of(0, c_row0, c_wb);
    int s1_t = 0, s1_kt = 0, s1_row0 = c_row0;
    const char* s1_wb = c_wb;
    int s2_t = 0, s2_kt = 0, s2_row0 = c_row0;
    const char* s2_wb = c_wb;
    auto advance = [&](int& t, int& kt, int& row0, const char*& wb) {
        if (kt + 1 < nkt) {
            ++kt;
        } else if (t + 1 < U) {
            kt = 0;
            ++t;
            unit_of(t, row0, wb);
        }                                                          // else: hold the last K-tile (harmless re-reads)
    };

    // W warm-up, as in gemm_panel_kernel: the workgroups of an XCD touch the whole packed weight between them before the
    // first K-tile, so that the K-tile stream meets it in L2.  Issued BEFORE the prologue's LDS-DMA: loads retire in
    // order, so the counted vmcnt(6) below has retired these too; the destination registers stay reserved until then.
    typedef unsigned warm_u32x4 __attribute__((ext_vector_type(4)));
    warm_u32x4 warmv = {0u, 0u, 0u, 0u};
    if (warm) {
        const long wbytes = (long)N * K * 2;
        for (long off = ((long)wg * T_THREADS + threadIdx.x) * 16; off < wbytes; off += (long)WX * T_THREADS * 16)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(warmv) : "v"(Wp + off) : "memory");
    }

    // prologue: K-tile 0 whole, K-tile 1 without its A1 (7 half-tiles, 14 LDS-DMA per wave)
    stage_a(0, 0, s1_row0, 0);
    stage_b(0, 0, s1_wb, 0);
    stage_b(1, 0, s1_wb, 0);
    stage_a(1, 0, s1_row0, 0);
    advance(s1_t, s1_kt, s1_row0, s1_wb);                          // s1 -> K-tile 1
    stage_a(0, 1, s1_row0, s1_kt);
    stage_b(0, 1, s1_wb, s1_kt);
    stage_b(1, 1, s1_wb, s1_kt);
    s2_t = s1_t, s2_kt = s1_kt, s2_row0 = s1_row0, s2_wb = s1_wb;
    advance(s2_t, s2_kt, s2_row0, s2_wb);                          // s2 -> K-tile 2
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               // K-tile 0 has landed (this wave's share)
    asm volatile("; warm-up retired" :: "v"(warmv));
    __syncthreads();                                               // ... everybody's; bias copy visible
    if (wr == 1) T_BARRIER();                                      // wave row 1 runs one barrier behind row 0

    f32x4 acc[4][8];
    bf16x8 areg[4][2], breg0[2][2], breg1[2][2];

    auto read_a = [&](int pb, int rh) {
        const char* base = lds + pb * T_BUF + (T_A0 + rh) * T_HALF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                areg[tt][ks] = *reinterpret_cast<const bf16x8*>(base + (tt >> 1) * 4096 + aoff[tt & 1][ks]);
    };
    auto read_b = [&](int pb, int ch, bf16x8 (&br)[2][2]) {
        const char* base = lds + pb * T_BUF + (T_B0 + ch) * T_HALF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                br[ii][ks] = *reinterpret_cast<const bf16x8*>(base + boff[ii][ks]);
    };
    auto quadrant = [&](int rh, int ch, const bf16x8 (&br)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
                    acc[2 * ch + ii][4 * rh + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        br[ii][ks], areg[tt][ks], acc[2 * ch + ii][4 * rh + tt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // Epilogue.  Store (and load) SHAPE: straight from the accumulators a wave-instruction covers 16 rows x 64 B, and a CU
    // cannot move that shape faster than ~30 GB/s however idle HBM is; whole 128-B lines (8 rows x 128 B) go at up to
    // ~75 GB/s per CU (tools/store_probe.hip).  So every 16-row x 128-B piece of the wave's tile takes a round trip through
    // a wave-private 2-KiB LDS slab (behind the bias copy; 16-B chunks XOR-swizzled by the row: conflict-free both ways):
    // written in the accumulator layout (lane = row fr, 16-B chunk 4 j + fq), read back as lane = (row l >> 3 and + 8,
    // chunk l & 7), so that every global access of the epilogue -- outputs, residual rows, saved pre-activation -- is
    // 8 rows x 128 B per instruction.  W is the MFMA A operand, so a lane's accumulator quad is 4 consecutive columns; for
    // bf16 outputs the interleaved column order of the weight fragments makes two quads 8 consecutive columns (16 B).
    auto epilogue = [&](int row0, int col0) {
        char* slab = lds + T_SLAB + wave * 2048;
        char* ap0 = slab + fr * 128 + ((fq ^ (fr & 7)) << 4);                // accumulator-layout position, j = 0
        char* ap1 = slab + fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);          //                              j = 1
        const int rr = lane >> 3, cc = lane & 7;
        char* lp0 = slab + rr * 128 + ((cc ^ (rr & 7)) << 4);                // line-layout position, rows rr and rr + 8
        char* lp1 = slab + (rr + 8) * 128 + ((cc ^ ((rr + 8) & 7)) << 4);
        const int mbase = row0 + wr * 128 + rr;                              // + 16 t (+ 8)
        if constexpr (PERM) {
            const long ocol = col0 + wc * 64 + cc * 8;
            auto leave = [&](const uint4& a, const uint4& b, bf16_t* dst, int t, int rowlim) {
                *reinterpret_cast<uint4*>(ap0) = a;
                *reinterpret_cast<uint4*>(ap1) = b;
                const uint4 x0 = *reinterpret_cast<const uint4*>(lp0);
                const uint4 x1 = *reinterpret_cast<const uint4*>(lp1);
                const int m0 = mbase + 16 * t, m1 = m0 + 8;
                if (m0 < rowlim) *reinterpret_cast<uint4*>(dst + (long)m0 * N + ocol) = x0;
                if (m1 < rowlim) *reinterpret_cast<uint4*>(dst + (long)m1 * N + ocol) = x1;
            };
            constexpr int HA = 4;                                  // GELU': row tiles of h requested ahead of their use
            struct H2 { uint4 a, b; };
            auto load_hq = [&](int t) -> H2 {
                int m0 = mbase + 16 * t, m1 = m0 + 8;
                m0 = m0 < M ? m0 : M - 1;
                m1 = m1 < M ? m1 : M - 1;
                return H2{*reinterpret_cast<const uint4*>(aux + (long)m0 * N + ocol),
                          *reinterpret_cast<const uint4*>(aux + (long)m1 * N + ocol)};
            };
            H2 hq[8];
            if constexpr (EPI == NEAT_EPI_GELU_BWD) {
#pragma unroll
                for (int t = 0; t < HA; ++t) hq[t] = load_hq(t);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                uint4 pk[2];
                if constexpr (EPI == NEAT_EPI_GELU_BWD) {
                    // the saved pre-activation arrives as whole lines and is turned into the accumulator layout
                    if (t + HA < 8) hq[t + HA] = load_hq(t + HA);
                    *reinterpret_cast<uint4*>(lp0) = hq[t].a;
                    *reinterpret_cast<uint4*>(lp1) = hq[t].b;
                    const uint4 hh[2] = {*reinterpret_cast<const uint4*>(ap0), *reinterpret_cast<const uint4*>(ap1)};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 lo = acc[2 * h][t], hi = acc[2 * h + 1][t];
                        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const uint32_t hp[4] = {hh[h].x, hh[h].y, hh[h].z, hh[h].w};
                        uint32_t dp[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                            dp[q] = pack2bf(v[2 * q] * quick_gelu_grad(h0), v[2 * q + 1] * quick_gelu_grad(h1));
                        }
                        pk[h] = make_uint4(dp[0], dp[1], dp[2], dp[3]);
                    }
                    leave(pk[0], pk[1], (bf16_t*)out0, t, M);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 lo = acc[2 * h][t], hi = acc[2 * h + 1][t];
                        pk[h] = make_uint4(pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3]));
                    }
                    if constexpr (EPI == NEAT_EPI_BIAS_GELU) {
                        // the saved pre-activation is the bf16-rounded value (kept for the first rows0 rows only: rows without
                        // a backward pass skip it); GELU sees the same rounded value
                        if (out0) leave(pk[0], pk[1], (bf16_t*)out0, t, rows0 < M ? rows0 : M);
                        uint4 up[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t hp[4] = {pk[h].x, pk[h].y, pk[h].z, pk[h].w};
                            uint32_t uq[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                                uq[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
                            }
                            up[h] = make_uint4(uq[0], uq[1], uq[2], uq[3]);
                        }
                        leave(up[0], up[1], (bf16_t*)out1, t, M);
                    } else {
                        leave(pk[0], pk[1], (bf16_t*)out0, t, M);
                    }
                }
            }
        } else {
            // fp32 outputs: two 32-column halves per row tile (16 rows x 128 B each).  The residual rows are loaded as whole
            // lines too, two row tiles ahead of their use; res may alias out0 (in place): a lane reads exactly the positions
            // it later writes, and reads them first.
            const long ocol = col0 + wc * 64 + cc * 4;                       // + 32 p
            float4 rq[3][4];                                                 // [t % 3][2 p + row half]
            auto load_rq = [&](int t) {
                int m0 = mbase + 16 * t, m1 = m0 + 8;
                m0 = m0 < M ? m0 : M - 1;
                m1 = m1 < M ? m1 : M - 1;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    rq[t % 3][2 * pq] = *reinterpret_cast<const float4*>(res + (long)m0 * N + ocol + 32 * pq);
                    rq[t % 3][2 * pq + 1] = *reinterpret_cast<const float4*>(res + (long)m1 * N + ocol + 32 * pq);
                }
            };
            if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                load_rq(0);
                load_rq(1);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                    if (t + 2 < 8) load_rq(t + 2);
                }
                const int m0 = mbase + 16 * t, m1 = m0 + 8;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    *reinterpret_cast<f32x4*>(ap0) = acc[2 * pq][t];
                    *reinterpret_cast<f32x4*>(ap1) = acc[2 * pq + 1][t];
                    float4 x0 = *reinterpret_cast<const float4*>(lp0);
                    float4 x1 = *reinterpret_cast<const float4*>(lp1);
                    if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                        const float4 r0 = rq[t % 3][2 * pq], r1 = rq[t % 3][2 * pq + 1];
                        x0.x += r0.x, x0.y += r0.y, x0.z += r0.z, x0.w += r0.w;
                        x1.x += r1.x, x1.y += r1.y, x1.z += r1.z, x1.w += r1.w;
                    }
                    if (m0 < M) *reinterpret_cast<float4*>((float*)out0 + (long)m0 * N + ocol + 32 * pq) = x0;
                    if (m1 < M) *reinterpret_cast<float4*>((float*)out0 + (long)m1 * N + ocol + 32 * pq) = x1;
                }
            }
        }
    };

    auto init_acc = [&](int col0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS) {
                const int n = PERM ? col0 + wc * 64 + 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : col0 + wc * 64 + 16 * i + 4 * fq;
                b0 = *reinterpret_cast<const f32x4*>(lds + T_BIAS + n * 4);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[i][t] = b0;
        }
    };

    // one K-tile = two merged phases of 32 MFMAs per wave; PB = LDS buffer of this K-tile.
    //   MP0 reads B0 B1 A0 (16 ds_read_b128), MP1 reads A1 (8); B0/B1 stay in registers for MP1.
    //   MP1 restages A0 B0 B1 (read in MP0, retired by the lgkmcnt(0) before MP0's barrier) for K-tile g+2,
    //   the next MP0 restages A1: 6 + 2 LDS-DMA per wave, which balances the 8 + 16 fragment reads.
    //   Waits (in-order vmcnt, LDS-DMA only in the loop): MP1 needs A0 B0 B1 of K-tile g+1 = everything but the
    //   8 youngest (A1(g+1), A0 B0 B1(g+2)); MP0 needs A1 of its own K-tile = everything but the 8 youngest
    //   (A0 B0 B1(g+1), A1(g+1)).  A buffer is read one phase after the wait that retires it.
    auto ktile = [&](auto pb_tag) {
        constexpr int PB = decltype(pb_tag)::value;
        read_b(PB, 0, breg0);
        read_b(PB, 1, breg1);
        read_a(PB, 0);
        stage_a(1, PB ^ 1, s1_row0, s1_kt);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T_BARRIER();
        quadrant(0, 0, breg0);
        quadrant(0, 1, breg1);
        T_BARRIER();
        read_a(PB, 1);
        stage_a(0, PB, s2_row0, s2_kt);
        stage_b(0, PB, s2_wb, s2_kt);
        stage_b(1, PB, s2_wb, s2_kt);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T_BARRIER();
        quadrant(1, 1, breg1);
        quadrant(1, 0, breg0);
        T_BARRIER();
        advance(s1_t, s1_kt, s1_row0, s1_wb);
        advance(s2_t, s2_kt, s2_row0, s2_wb);
    };

    const int G = U * nkt;
    for (int g = 0; g < G; g += 2) {
        if (c_kt == 0) init_acc((int)((c_wb - Wp) / wcb_bytes) << 8);
        ktile(std::integral_constant<int, 0>{});
        ktile(std::integral_constant<int, 1>{});
        c_kt += 2;
        if (c_kt == nkt) {
            epilogue(c_row0, (int)((c_wb - Wp) / wcb_bytes) << 8);
            c_kt = 0;
            if (++c_t < U) unit_of(c_t, c_row0, c_wb);
        }
    }
    if (wr == 0) T_BARRIER();                                      // pairs with wave row 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // no LDS-DMA may land after the workgroup ends
}

// f32 weights -> bf16 in the tile256 LDS image: half-tile ((cb*nkt + kt)*2 + h) is 16 KiB = 128 rows x 8 chunks;
// row lr holds column cb*256 + (lr>>5)*64 + h*32 + (lr&31), LDS chunk p holds k = kt*64 + (p ^ swz(lr))*8 .. +7
__global__ void pack_weight_t256_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                        int transposed) {
    const long total = (long)N * K / 8;
    const int nkt = K >> 6;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const int pos = (int)(u & 1023);
        const long ht = u >> 10;
        const int h = (int)(ht & 1);
        const int kt = (int)((ht >> 1) % nkt);
        const int cb = (int)((ht >> 1) / nkt);
        const int lr = pos >> 3;
        const int c = (pos & 7) ^ t_swz(lr);
        const int n = cb * 256 + (lr >> 5) * 64 + h * 32 + (lr & 31);
        const int k0 = kt * 64 + c * 8;
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

template <int EPI>
static int launch_gemm_t256(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                            const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s,
                            int rows0 = -1) {
    static unsigned long long attr_done = 0;
    if (int rc = ensure_dynamic_lds((const void*)gemm_tile256_kernel<EPI>, T_LDS, &attr_done,
                                    "hipFuncSetAttribute(gemm_tile256)"))
        return rc;
    const long units = (long)cdiv(M, 256) * (N / 256);
    const int cus = num_cus();
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < cus ? want : cus);
    hipLaunchKernelGGL((gemm_tile256_kernel<EPI>), dim3(grid), dim3(T_THREADS), T_LDS, s, A,
                       reinterpret_cast<const char*>(Wp), bias, res, aux, out0, out1, M, N, K, rows0 < 0 ? M : rows0,
                       gemm_warm_w_enabled());
    NEAT_LAUNCH_CHECK("neat_gemm_bf16(tile256)");
    return NEAT_OK;
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

// ---- tile256 entry points ----
extern "C" int neat_pack_weight_bf16_t256(const float* src, uint16_t* dst, int N, int K, int transposed, void* stream) {
    NEAT_REQUIRE(src && dst, NEAT_ERR_ARG, "neat_pack_weight_bf16_t256: null pointer");
    NEAT_REQUIRE(N > 0 && K > 0 && t256_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_pack_weight_bf16_t256: N=%d K=%d (need N %% 256 == 0, K %% 128 == 0, N <= %d)", N, K, T_MAXN);
    long g = ((long)N * K / 8 + 255) / 256;
    hipLaunchKernelGGL(pack_weight_t256_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, N, K, transposed);
    NEAT_LAUNCH_CHECK("neat_pack_weight_bf16_t256");
    return NEAT_OK;
}

extern "C" int neat_gemm_bf16_t256(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                                   const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                                   void* stream) {
    NEAT_REQUIRE(A && W, NEAT_ERR_ARG, "neat_gemm_bf16_t256: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && t256_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_gemm_bf16_t256: M=%d N=%d K=%d unsupported (need N %% 256 == 0, K %% 128 == 0, N <= %d)", M, N, K,
                 T_MAXN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case NEAT_EPI_BF16:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_t256: out0 is null");
            return launch_gemm_t256<NEAT_EPI_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_BF16:
            NEAT_REQUIRE(out0 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_t256: BIAS_BF16 needs bias, out0");
            return launch_gemm_t256<NEAT_EPI_BIAS_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_RES_F32:
            NEAT_REQUIRE(out0 && bias && res, NEAT_ERR_ARG, "neat_gemm_bf16_t256: BIAS_RES_F32 needs bias, res, out0");
            return launch_gemm_t256<NEAT_EPI_BIAS_RES_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_GELU:
            NEAT_REQUIRE(out1 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_t256: BIAS_GELU needs bias, out1");
            return launch_gemm_t256<NEAT_EPI_BIAS_GELU>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_GELU_BWD:
            NEAT_REQUIRE(out0 && aux, NEAT_ERR_ARG, "neat_gemm_bf16_t256: GELU_BWD needs aux, out0");
            return launch_gemm_t256<NEAT_EPI_GELU_BWD>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_F32:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_t256: out0 is null");
            return launch_gemm_t256<NEAT_EPI_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        default:
            return set_error(NEAT_ERR_ARG, "neat_gemm_bf16_t256: unknown epilogue %d", epilogue);
    }
}

// c_fc of a forward pass whose LAST rows have no backward pass (the `neg` captions of the TTA step): out1 = QuickGELU(A W^T
// + bias) for all M rows, the pre-activation out0 only for the first out0_rows rows (out0 holds out0_rows rows).  Saves a
// quarter of the 323 MB pre-activation store of the step's largest GEMM.
extern "C" int neat_gemm_bf16_t256_gelu_rows(const uint16_t* A, const uint16_t* W, const float* bias, void* out0,
                                             int out0_rows, void* out1, int M, int N, int K, void* stream) {
    NEAT_REQUIRE(A && W && bias && out1, NEAT_ERR_ARG, "neat_gemm_bf16_t256_gelu_rows: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && t256_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_gemm_bf16_t256_gelu_rows: M=%d N=%d K=%d unsupported (need N %% 256 == 0, K %% 128 == 0, N <= %d)", M,
                 N, K, T_MAXN);
    NEAT_REQUIRE(out0_rows >= 0 && out0_rows <= M && (out0 || out0_rows == 0), NEAT_ERR_ARG,
                 "neat_gemm_bf16_t256_gelu_rows: out0_rows=%d outside 0..M=%d (or out0 is null)", out0_rows, M);
    return launch_gemm_t256<NEAT_EPI_BIAS_GELU>(A, W, bias, nullptr, nullptr, out0_rows ? out0 : nullptr, out1, M, N, K,
                                                (hipStream_t)stream, out0_rows);
}

// Which of the two GEMMs a caller holding BOTH weight layouts should launch (1 = tile256).  Measured inside the
// TTA step on MI355X (profiles/r01_*): tile256 wins where the epilogue only stores and the grid fills the chip for
// at least ~1.5 rounds of 256 x 256 units (c_fc 247 vs 274 us, the bf16 dgrad GEMMs 81 vs 91 us on average; the qkv
// projection: neutral at C = 512, +0.8 % of the ViT-L-14 step).  The residual / GELU' epilogues (which load as much
// as they store) stay on gemm_panel_kernel (two workgroups per CU overlap one's epilogue with the other's main loop)
// for the ViT-B weights and move to tile256 from 768 x 3072 weights on (ViT-L-14: c_proj 405 vs 445 us, the GELU'
// dgrad 352 vs 371 us; +2.1 % of that step).
extern "C" int neat_gemm_prefers_t256(int M, int N, int K, int epilogue) {
    if (M <= 0 || N <= 0 || K <= 0 || !t256_shape_ok(N, K)) return 0;
    const bool store_only = epilogue == NEAT_EPI_BF16 || epilogue == NEAT_EPI_BIAS_GELU || epilogue == NEAT_EPI_BIAS_BF16;
    // epilogues that load as much as they store: the fp32 residual one moves to tile256 from N = 768 on (measured at
    // 768 x 768: 170 vs 184 us; at N = 512 the panel kernel wins, 103 vs 112-123 us and 234 vs 255 us), GELU' from 768 x 3072 on
    if (epilogue == NEAT_EPI_BIAS_RES_F32 || epilogue == NEAT_EPI_F32) {
        if (N < 768) return 0;
    } else if (!store_only && (long)N * K < 768L * 3072) {
        return 0;
    }
    const long units = (long)cdiv(M, 256) * (N / 256);
    const int cus = num_cus();
    const long rounds = (units + cus - 1) / cus;
    static int min_fill_x2 = -1;                // NEAT_GEMM_T256_MIN_FILL_X2 (experiments): units * 2 >= this * CUs; default 3
    if (min_fill_x2 < 0) {
        const char* e = getenv("NEAT_GEMM_T256_MIN_FILL_X2");
        min_fill_x2 = e ? atoi(e) : 3;
    }
    return units * 2 >= (long)min_fill_x2 * cus && units * 4 >= 3 * rounds * cus;
}
