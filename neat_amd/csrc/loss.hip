// Coarse alignment, the three-term TTA loss with its gradient, and AdamW -- all fp32.
// Replaces reference train_TTA.py:31-42 (coarse_aligning_t2i), :45-48 (softmax_entropy),
// :53-65 (semantics_reversion_loss), :77-91 (textual_triplet_loss), :138-147 (loss + backward + step)
// and torch.optim.AdamW (main_TTA.py:271).
//
// The similarity products feed an argmax (coarse index) and the score matrix that recall@k is computed
// on, so they are exact fp32 FMA chains in k order (no bf16, no split-K, no atomics): results are
// deterministic and ties resolve to the lowest index like torch's CPU argmax / max(dim).
#include "common.h"

namespace neat {

// ---- exact-fp32 tile GEMM on the matrix cores ----
// v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per
// product, /opt/skills/guides/cdna_hip_programming.md §3), so the products below are the SAME numbers the earlier VALU
// kernels produced (acc = fmaf(a_k, b_k, acc), k ascending) at up to 3x their rate and with the VALU left free.
// Block = 64 x 64 outputs, 4 waves as 2 x 2, one 32 x 32 accumulator tile per wave and product; operands pass through
// k-major LDS slabs of 16 k (double buffered, the next slab's global loads in flight under the MFMAs).
constexpr int SG_T = 64;   // output tile
constexpr int SG_K = 16;   // k slab
constexpr int SG_LD = SG_T + 4;
constexpr int SG_SLAB = SG_K * SG_LD;       // floats per operand slab

// A [64 x 16] operand slab in registers: 4 floats per thread.
//   row-major operand (element (m,k) at src[m*ld + k], "NT"): thread -> row t>>2, k quad t&3 (one 16-B load);
//   k-major operand   (element (k,m) at src[k*ld + m], "T"):  thread -> k t>>4, rows 4*(t&15)..+3 (scalar loads: no
//   alignment assumed); k >= K reads as zero, which leaves an fmaf chain unchanged.
template <bool KMAJOR>
__device__ __forceinline__ float4 sg_fetch(const float* __restrict__ src, int rows, int K, long ld, int r0, int k0) {
    if constexpr (!KMAJOR) {
        int gr = r0 + (threadIdx.x >> 2);
        gr = gr < rows ? gr : rows - 1;
        return *reinterpret_cast<const float4*>(src + (long)gr * ld + k0 + (threadIdx.x & 3) * 4);
    } else {
        const int k = k0 + (threadIdx.x >> 4);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = r0 + (threadIdx.x & 15) * 4 + j;
            m = m < rows ? m : rows - 1;
            v[j] = k < K ? src[(long)k * ld + m] : 0.0f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}
template <bool KMAJOR>
__device__ __forceinline__ void sg_put(float* dst, float4 v) {
    if constexpr (!KMAJOR) {
        const int r = threadIdx.x >> 2, kq = threadIdx.x & 3;
        dst[(kq * 4 + 0) * SG_LD + r] = v.x;
        dst[(kq * 4 + 1) * SG_LD + r] = v.y;
        dst[(kq * 4 + 2) * SG_LD + r] = v.z;
        dst[(kq * 4 + 3) * SG_LD + r] = v.w;
    } else {
        *reinterpret_cast<float4*>(dst + (threadIdx.x >> 4) * SG_LD + (threadIdx.x & 15) * 4) = v;
    }
}
// lane's fragment element of k-pair kk: A[i = lane&31][k = 2kk + (lane>>5)] (same map for the B operand)
__device__ __forceinline__ float sg_frag(const float* slab, int kk, int off32, int lane) {
    return slab[(2 * kk + (lane >> 5)) * SG_LD + off32 + (lane & 31)];
}
// accumulator register r of a 32 x 32 tile: row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31
__device__ __forceinline__ int sg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// one problem of a launch: C[M,N] = A . B^T over K, exact fp32
struct SgProb {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long lda, ldb;
    long ldc;     // row pitch of C in floats; 0 = N
};

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(SgProb p0, SgProb p1) {
    __shared__ __attribute__((aligned(16))) float As[2][SG_SLAB], Bs[2][SG_SLAB];
    const SgProb p = blockIdx.z == 0 ? p0 : p1;
    const int m0 = blockIdx.y * SG_T, n0 = blockIdx.x * SG_T;
    if (m0 >= p.M || n0 >= p.N) return;                         // uniform per block (the two problems share one grid)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int ns = (p.K + SG_K - 1) / SG_K;
    float4 ra = sg_fetch<A_KM>(p.A, p.M, p.K, p.lda, m0, 0), rb = sg_fetch<B_KM>(p.B, p.N, p.K, p.ldb, n0, 0);
    sg_put<A_KM>(As[0], ra);
    sg_put<B_KM>(Bs[0], rb);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        if (s + 1 < ns) {
            ra = sg_fetch<A_KM>(p.A, p.M, p.K, p.lda, m0, (s + 1) * SG_K);
            rb = sg_fetch<B_KM>(p.B, p.N, p.K, p.ldb, n0, (s + 1) * SG_K);
        }
        const float* as = As[s & 1];
        const float* bs = Bs[s & 1];
#pragma unroll
        for (int kk = 0; kk < SG_K / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sg_frag(as, kk, wm, lane), sg_frag(bs, kk, wn, lane), acc, 0, 0, 0);
        if (s + 1 < ns) {
            sg_put<A_KM>(As[(s + 1) & 1], ra);
            sg_put<B_KM>(Bs[(s + 1) & 1], rb);
        }
        __syncthreads();
    }
    const int n = n0 + wn + (lane & 31);
    const long ldc = p.ldc ? p.ldc : p.N;
    if (n < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + sg_row(r, lane);
            if (m < p.M) p.C[(long)m * ldc + n] = acc[r];
        }
    }
}

// The same product for LARGE problems (the 25 014 x 5 000 x 512 retrieval similarity): 128 x 128 outputs per workgroup, 8 waves
// as 2 x 4, each with a 64 x 32 register tile = two 32 x 32 accumulators (3 LDS reads per 2 MFMAs instead of 2 per 1), and a
// global -> register -> LDS pipeline two slabs deep.  Every output is still the k-ascending fmaf chain of the kernel above:
// bit-identical results.  Four workgroups per CU (34 KiB of LDS, 71 VGPRs) take turns on the matrix pipe.  Workgroups are
// dealt to the XCDs (blockIdx % 8) in 8 x 8 super-tiles, so the 64 workgroups an XCD runs at once share 2 MiB of A rows and
// 2 MiB of B rows in its 4-MiB L2.  Measured (25 014 x 5 000): SQ_VALU_MFMA_BUSY_CYCLES = 80 % of the SIMD cycles at the
// 2.07 GHz the chip holds under it; between K = 512 and K = 1 024 the marginal rate is 150 TFLOP/s, the rest is per-tile
// (first loads 5 %, C stores 2.5 %, the partial last round 4 %); the 4-wave form (64 x 64 per wave) measured the same.
constexpr int SB_T = 128;
constexpr int SB_LD = SB_T + 4;
constexpr int SB_SLAB = SG_K * SB_LD;

template <int NW>   // waves per workgroup: 4 (wave tile 64 x 64) or 8 (64 x 32)
__global__ __launch_bounds__(NW * 64) void gemm_f32_mfma128_kernel(SgProb p, int ntm, int ntn) {
    constexpr int BT = 8 / NW;          // 32-column accumulator tiles per wave
    constexpr int H = 8 / NW;           // 16-B loads per thread, operand and slab
    __shared__ __attribute__((aligned(16))) float As[2][SB_SLAB], Bs[2][SB_SLAB];
    const int SN = (ntn + 7) >> 3, SM = (ntm + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = (j >> 6) * 8 + xcd;                            // super-tile number
    if (g >= SM * SN) return;                                    // uniform per workgroup, before any barrier
    const int tm = (g / SN) * 8 + ((j & 63) >> 3), tn = (g % SN) * 8 + (j & 7);
    if (tm >= ntm || tn >= ntn) return;
    const int m0 = tm * SB_T, n0 = tn * SB_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / (NW / 2)) * 64, wn = (wave % (NW / 2)) * (32 * BT);
    f32x16 acc[2][BT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // operand slab [128 rows x 16 k]: thread -> rows t>>2 and 64 + (t>>2), k quad t&3 (two 16-B loads per operand)
    const int lr = threadIdx.x >> 2, kq = threadIdx.x & 3;
    constexpr int RP = NW * 16;         // rows per load pass
    const float* ap[H];
    const float* bp[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        int ra = m0 + h * RP + lr, rb = n0 + h * RP + lr;
        ra = ra < p.M ? ra : p.M - 1;
        rb = rb < p.N ? rb : p.N - 1;
        ap[h] = p.A + (long)ra * p.lda + kq * 4;
        bp[h] = p.B + (long)rb * p.ldb + kq * 4;
    }
    auto put = [&](float* dst, const float4& v, int h) {            // k-major slabs, as in the kernel above
        float* d = dst + (kq * 4) * SB_LD + h * RP + lr;
        d[0] = v.x;
        d[SB_LD] = v.y;
        d[2 * SB_LD] = v.z;
        d[3 * SB_LD] = v.w;
    };
    // slab s is computed from LDS while slab s+1 waits in registers (loaded one whole slab ago: an L2 miss has 2 x 2 048
    // MFMA cycles to arrive) and slab s+2 is being requested
    float4 va[H], vb[H], na[H], nb[H];
    const int ns = p.K / SG_K;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        va[h] = *reinterpret_cast<const float4*>(ap[h]);
        vb[h] = *reinterpret_cast<const float4*>(bp[h]);
    }
    if (ns > 1) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            na[h] = *reinterpret_cast<const float4*>(ap[h] + SG_K);
            nb[h] = *reinterpret_cast<const float4*>(bp[h] + SG_K);
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        put(As[0], va[h], h);
        put(Bs[0], vb[h], h);
    }
    __syncthreads();
    const int fo = (lane >> 5) * SB_LD + (lane & 31);
    for (int s = 0; s < ns; ++s) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            va[h] = na[h];
            vb[h] = nb[h];
        }
        if (s + 2 < ns) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                na[h] = *reinterpret_cast<const float4*>(ap[h] + (s + 2) * SG_K);
                nb[h] = *reinterpret_cast<const float4*>(bp[h] + (s + 2) * SG_K);
            }
        }
        const float* as = As[s & 1] + fo + wm;
        const float* bs = Bs[s & 1] + fo + wn;
#pragma unroll
        for (int kk = 0; kk < SG_K / 2; ++kk) {
            const float a0 = as[2 * kk * SB_LD], a1 = as[2 * kk * SB_LD + 32];
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const float bv = bs[2 * kk * SB_LD + 32 * b];
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][b], 0, 0, 0);
            }
        }
        if (s + 1 < ns) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                put(As[(s + 1) & 1], va[h], h);
                put(Bs[(s + 1) & 1], vb[h], h);
            }
        }
        __syncthreads();
    }
    const long ldc = p.ldc ? p.ldc : p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const int n = n0 + wn + b * 32 + (lane & 31);
            if (n < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm + a * 32 + sg_row(r, lane);
                    if (m < p.M) p.C[(long)m * ldc + n] = acc[a][b][r];
                }
            }
        }
}

// Fused coarse alignment tile: S = q G^T (stored), A = (pos G^T)(1 - relu(neg G^T)) -> per-row
// (max, first argmax) of this 64-column tile.
__global__ __launch_bounds__(256) void coarse_tile_kernel(const float* __restrict__ q, const float* __restrict__ pos,
                                                          const float* __restrict__ neg, const float* __restrict__ G,
                                                          float* __restrict__ S, float* __restrict__ pval,
                                                          int* __restrict__ pidx, int B, int N, int D, int ntn) {
    __shared__ __attribute__((aligned(16))) float Qs[2][SG_SLAB], Ps[2][SG_SLAB], Ns[2][SG_SLAB], Gs[2][SG_SLAB];
    __shared__ float rval[2][SG_T];
    __shared__ int ridx[2][SG_T];
    const int m0 = blockIdx.y * SG_T, n0 = blockIdx.x * SG_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 aq, ap, an;
#pragma unroll
    for (int r = 0; r < 16; ++r) aq[r] = ap[r] = an[r] = 0.0f;
    const int ns = D / SG_K;
    float4 rq = sg_fetch<false>(q, B, D, D, m0, 0), rp = sg_fetch<false>(pos, B, D, D, m0, 0),
           rn = sg_fetch<false>(neg, B, D, D, m0, 0), rg = sg_fetch<false>(G, N, D, D, n0, 0);
    sg_put<false>(Qs[0], rq);
    sg_put<false>(Ps[0], rp);
    sg_put<false>(Ns[0], rn);
    sg_put<false>(Gs[0], rg);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        if (s + 1 < ns) {
            const int k0 = (s + 1) * SG_K;
            rq = sg_fetch<false>(q, B, D, D, m0, k0);
            rp = sg_fetch<false>(pos, B, D, D, m0, k0);
            rn = sg_fetch<false>(neg, B, D, D, m0, k0);
            rg = sg_fetch<false>(G, N, D, D, n0, k0);
        }
        const int b = s & 1;
#pragma unroll
        for (int kk = 0; kk < SG_K / 2; ++kk) {
            const float g = sg_frag(Gs[b], kk, wn, lane);
            aq = __builtin_amdgcn_mfma_f32_32x32x2f32(sg_frag(Qs[b], kk, wm, lane), g, aq, 0, 0, 0);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(sg_frag(Ps[b], kk, wm, lane), g, ap, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x2f32(sg_frag(Ns[b], kk, wm, lane), g, an, 0, 0, 0);
        }
        if (s + 1 < ns) {
            sg_put<false>(Qs[b ^ 1], rq);
            sg_put<false>(Ps[b ^ 1], rp);
            sg_put<false>(Ns[b ^ 1], rn);
            sg_put<false>(Gs[b ^ 1], rg);
        }
        __syncthreads();
    }
    const int n = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm + sg_row(r, lane);
        const int m = m0 + row;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (n < N) {
            if (m < B) S[(long)m * N + n] = aq[r];
            const float a = ap[r] * (1.0f - fmaxf(an[r], 0.0f));
            if (a > best) { best = a; bi = n; }
        }
        // the 32 lanes of a half-wave hold the 32 columns of this row; ties -> lowest index
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((lane & 31) == 0) { rval[wave & 1][row] = best; ridx[wave & 1][row] = bi; }
    }
    __syncthreads();
    if (threadIdx.x < SG_T) {
        const int m = m0 + threadIdx.x;
        float best = rval[0][threadIdx.x];
        int bi = ridx[0][threadIdx.x];
        const float ov = rval[1][threadIdx.x];
        const int oi = ridx[1][threadIdx.x];
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        if (m < B) {
            pval[(long)m * ntn + blockIdx.x] = best;
            pidx[(long)m * ntn + blockIdx.x] = bi;
        }
    }
}

// The image-to-text form (reference train_TTA.py:20-28, a helper the reference defines and never calls): ONE query, TWO
// galleries -- A = (q Gp^T)(1 - sigmoid(q Gn^T)) -> per-row (max, first argmax) of this 64-column tile.  Same tiling and
// exact-fp32 MFMA chains as coarse_tile_kernel; no score matrix is stored (the reference returns the gathered rows only).
__global__ __launch_bounds__(256) void coarse_i2t_tile_kernel(const float* __restrict__ q, const float* __restrict__ Gp,
                                                              const float* __restrict__ Gn, float* __restrict__ pval,
                                                              int* __restrict__ pidx, int B, int N, int D, int ntn) {
    __shared__ __attribute__((aligned(16))) float Qs[2][SG_SLAB], Ps[2][SG_SLAB], Ns[2][SG_SLAB];
    __shared__ float rval[2][SG_T];
    __shared__ int ridx[2][SG_T];
    const int m0 = blockIdx.y * SG_T, n0 = blockIdx.x * SG_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 ap, an;
#pragma unroll
    for (int r = 0; r < 16; ++r) ap[r] = an[r] = 0.0f;
    const int ns = D / SG_K;
    float4 rq = sg_fetch<false>(q, B, D, D, m0, 0), rp = sg_fetch<false>(Gp, N, D, D, n0, 0),
           rn = sg_fetch<false>(Gn, N, D, D, n0, 0);
    sg_put<false>(Qs[0], rq);
    sg_put<false>(Ps[0], rp);
    sg_put<false>(Ns[0], rn);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        if (s + 1 < ns) {
            const int k0 = (s + 1) * SG_K;
            rq = sg_fetch<false>(q, B, D, D, m0, k0);
            rp = sg_fetch<false>(Gp, N, D, D, n0, k0);
            rn = sg_fetch<false>(Gn, N, D, D, n0, k0);
        }
        const int b = s & 1;
#pragma unroll
        for (int kk = 0; kk < SG_K / 2; ++kk) {
            const float a = sg_frag(Qs[b], kk, wm, lane);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sg_frag(Ps[b], kk, wn, lane), ap, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sg_frag(Ns[b], kk, wn, lane), an, 0, 0, 0);
        }
        if (s + 1 < ns) {
            sg_put<false>(Qs[b ^ 1], rq);
            sg_put<false>(Ps[b ^ 1], rp);
            sg_put<false>(Ns[b ^ 1], rn);
        }
        __syncthreads();
    }
    const int n = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm + sg_row(r, lane);
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (n < N) {
            const float a = ap[r] * (1.0f - 1.0f / (1.0f + expf(-an[r])));
            if (a > best) { best = a; bi = n; }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((lane & 31) == 0) { rval[wave & 1][row] = best; ridx[wave & 1][row] = bi; }
    }
    __syncthreads();
    if (threadIdx.x < SG_T) {
        const int m = m0 + threadIdx.x;
        float best = rval[0][threadIdx.x];
        int bi = ridx[0][threadIdx.x];
        const float ov = rval[1][threadIdx.x];
        const int oi = ridx[1][threadIdx.x];
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        if (m < B) {
            pval[(long)m * ntn + blockIdx.x] = best;
            pidx[(long)m * ntn + blockIdx.x] = bi;
        }
    }
}

// one wavefront per caption row: (max, first argmax) over the column tiles, then the gallery row is copied out
__global__ __launch_bounds__(64) void coarse_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                          const float* __restrict__ G, int32_t* __restrict__ idx_out,
                                                          float* __restrict__ coarse, int ntn, int D) {
    const int m = blockIdx.x, lane = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = lane; t < ntn; t += 64) {
        const float v = pval[(long)m * ntn + t];
        const int i = pidx[(long)m * ntn + t];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: torch returns the NaN position; never hit with finite features
    if (lane == 0) idx_out[m] = bi;
    for (int d = lane; d < D; d += 64) coarse[(long)m * D + d] = G[(long)bi * D + d];
}

// ---- loss rows ----
__device__ __forceinline__ float blk_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float blk_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
    return t;
}

// row j of Z (= coarse_j . text_i, unscaled): entropy H_j and dL/dZ_ji written in place
__device__ __forceinline__ void entropy_row(float* __restrict__ Z, float* __restrict__ Hrow, int row, int B, float invT1,
                                            float* red) {
    float* z = Z + (long)row * B;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < B; i += blockDim.x) mx = fmaxf(mx, z[i] * invT1);
    mx = blk_max(mx, red);
    float se = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) se += expf(z[i] * invT1 - mx);
    se = blk_sum(se, red);
    const float lse = mx + logf(se);
    float h = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float lp = z[i] * invT1 - lse;
        h -= expf(lp) * lp;
    }
    h = blk_sum(h, red);
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float lp = z[i] * invT1 - lse;
        z[i] = -expf(lp) * (lp + h) / (float)B;
    }
    if (threadIdx.x == 0) Hrow[row] = h;
}

// local row i of U (= inv_i . coarse_j): hinge max and its first argmax
__device__ __forceinline__ void sr_row(const float* __restrict__ U, float* __restrict__ srm, int* __restrict__ srj, int i,
                                       int B, int row0, float* rv, int* ri) {
    const int gi = row0 + i;
    const float* u = U + (long)i * B;
    const float diag = u[gi];
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
        const float c = (j == gi) ? 0.0f : fmaxf(diag - u[j], 0.0f);
        if (c > best) { best = c; bj = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bj, o, 64);
        if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
    }
    if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (rv[w] > best || (rv[w] == best && ri[w] < bj)) { best = rv[w]; bj = ri[w]; }
        srm[i] = best;
        // gradient flows to (i, j*) only where clamp(min=0) passes (raw >= 0) and j* is not the masked diagonal
        const bool act = (bj != gi) && (diag - u[bj] >= 0.0f);
        srj[i] = act ? bj : -1;
    }
}

// ONE launch for the row passes of both [.,B] matrices: blocks 0..B-1 take the entropy rows of Z, blocks B..B+Bl-1 the
// semantics-reversion rows of U
__global__ __launch_bounds__(256) void loss_rows_kernel(float* __restrict__ Z, float* __restrict__ Hrow,
                                                        const float* __restrict__ U, float* __restrict__ srm,
                                                        int* __restrict__ srj, int B, int row0, float invT1) {
    __shared__ float red[8];
    __shared__ int ri[8];
    if ((int)blockIdx.x < B) entropy_row(Z, Hrow, blockIdx.x, B, invT1, red);
    else sr_row(U, srm, srj, (int)blockIdx.x - B, B, row0, red, ri);
}

// gradients for local row i: dtext (in: sum_j dZ[j,gi] coarse[j,:], from the tile GEMM; out: the gradient), dpos, dinv;
// triplet row term
__global__ __launch_bounds__(256) void loss_grad_kernel(
    const float* __restrict__ text, const float* __restrict__ pos, const float* __restrict__ inv,
    const float* __restrict__ coarse_all, const int* __restrict__ srj, float* __restrict__ dtext,
    float* __restrict__ dpos, float* __restrict__ dinv, float* __restrict__ trirow, int B, int row0, int D, float invT1,
    float invT2) {
    __shared__ float red[8];
    const int i = blockIdx.x, gi = row0 + i;
    const float* t = text + (long)i * D;
    const float* p = pos + (long)i * D;
    const float* v = inv + (long)i * D;
    float a = 0.f, b = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float e1 = t[d] - v[d] + 1e-6f, e2 = t[d] - p[d] + 1e-6f;
        a += e1 * e1;
        b += e2 * e2;
    }
    const float dtv = sqrtf(blk_sum(a, red));
    const float dtp = sqrtf(blk_sum(b, red));
    const float invB = 1.0f / (float)B;
    const int js = srj[i];
    const float gsr = js >= 0 ? invB * invT2 : 0.0f;
    const float* cg = coarse_all + (long)gi * D;
    const float* cj = coarse_all + (long)(js >= 0 ? js : gi) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float acc = dtext[(long)i * D + d];
        const float gtv = (t[d] - v[d] + 1e-6f) / dtv * invB;
        const float gtp = (t[d] - p[d] + 1e-6f) / dtp * invB;
        dtext[(long)i * D + d] = acc * invT1 - gtv + 5.0f * gtp;
        dinv[(long)i * D + d] = gsr * (cg[d] - cj[d]) + gtv;
        dpos[(long)i * D + d] = -5.0f * gtp;
    }
    if (threadIdx.x == 0) trirow[i] = (2.0f - dtv) + 5.0f * dtp;
}

// losses[0] = mean_j H_j (global); losses[1] = sum_local m_i / (B*T2); losses[2] = sum_local tri_i / B
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ Hrow, const float* __restrict__ srm,
                                                            const float* __restrict__ trirow, float* __restrict__ losses,
                                                            int B, int Bl, float invT2) {
    __shared__ float red[8];
    float h = 0.f, s = 0.f, t = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) h += Hrow[i];
    for (int i = threadIdx.x; i < Bl; i += blockDim.x) { s += srm[i]; t += trirow[i]; }
    h = blk_sum(h, red);
    s = blk_sum(s, red);
    t = blk_sum(t, red);
    if (threadIdx.x == 0) {
        losses[0] = h / B;
        losses[1] = s / B * invT2;
        losses[2] = t / B;
    }
}

// All scalar factors are formed in double on the host exactly like torch.optim.AdamW does in Python
// (1 - beta2 is 0.001, not 1 - float(0.999)) and rounded to fp32 once.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float decay, float omb1, float b2, float omb2, float eps,
                             float step_size, float sqrt_bc2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * omb1;                 // lerp_
        const float vi = v[i] * b2 + gi * gi * omb2;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

template <bool A_KM, bool B_KM>
static void launch_sg(const SgProb& p0, const SgProb* p1, hipStream_t s) {
    const int M = p1 ? (p0.M > p1->M ? p0.M : p1->M) : p0.M, N = p1 ? (p0.N > p1->N ? p0.N : p1->N) : p0.N;
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KM, B_KM>), dim3(cdiv(N, SG_T), cdiv(M, SG_T), p1 ? 2 : 1), dim3(256), 0, s,
                       p0, p1 ? *p1 : p0);
}

int launch_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, long ldc, hipStream_t s) {
    NEAT_REQUIRE(A && B && C, NEAT_ERR_ARG, "gemm_f32_nt: null pointer");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && K % SG_K == 0, NEAT_ERR_SHAPE, "gemm_f32_nt: K=%d must be a multiple of 16", K);
    NEAT_REQUIRE(ldc >= N, NEAT_ERR_SHAPE, "gemm_f32_nt: ldc=%ld < N=%d", ldc, N);
    const int ntm = cdiv(M, SB_T), ntn = cdiv(N, SB_T);
    if ((long)ntm * ntn >= 512) {                                // a full chip of 128 x 128 tiles: the large-problem kernel
        const int nsup = cdiv(ntm, 8) * cdiv(ntn, 8);
        hipLaunchKernelGGL(gemm_f32_mfma128_kernel<8>, dim3(cdiv(nsup, 8) * 8 * 64), dim3(512), 0, s,
                           SgProb{A, B, C, M, N, K, K, K, ldc}, ntm, ntn);
    } else {
        launch_sg<false, false>(SgProb{A, B, C, M, N, K, K, K, ldc}, nullptr, s);
    }
    NEAT_LAUNCH_CHECK("gemm_f32_nt");
    return NEAT_OK;
}

}  // namespace neat

using namespace neat;

extern "C" int neat_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
    return launch_gemm_f32_nt(A, B, C, M, N, K, N, (hipStream_t)stream);
}

extern "C" long neat_coarse_ws_floats(int B, int N) { return 2L * B * cdiv(N, SG_T); }

extern "C" int neat_coarse_align(const float* q, const float* pos, const float* neg, const float* G, float* S_out,
                                 int32_t* idx_out, float* coarse_out, float* ws, int B, int N, int D, void* stream) {
    NEAT_REQUIRE(q && pos && neg && G && S_out && idx_out && coarse_out && ws, NEAT_ERR_ARG, "neat_coarse_align: null pointer");
    NEAT_REQUIRE(B > 0 && N > 0 && D > 0 && D % SG_K == 0, NEAT_ERR_SHAPE, "neat_coarse_align: D=%d must be a multiple of 16", D);
    const int ntn = cdiv(N, SG_T);
    float* pval = ws;
    int* pidx = reinterpret_cast<int*>(ws + (long)B * ntn);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(coarse_tile_kernel, dim3(ntn, cdiv(B, SG_T)), dim3(256), 0, s, q, pos, neg, G, S_out, pval, pidx, B,
                       N, D, ntn);
    NEAT_LAUNCH_CHECK("neat_coarse_align(tile)");
    hipLaunchKernelGGL(coarse_final_kernel, dim3(B), dim3(64), 0, s, pval, pidx, G, idx_out, coarse_out, ntn, D);
    NEAT_LAUNCH_CHECK("neat_coarse_align(final)");
    return NEAT_OK;
}

// Image-to-text coarse alignment (reference train_TTA.py:20-28): idx = argmax_j (q Gp^T)(1 - sigmoid(q Gn^T)) (first max),
// coarse_out = G[idx].  q f32 [B,D]; G, Gp, Gn f32 [N,D]; ws as for neat_coarse_align.
extern "C" int neat_coarse_align_i2t(const float* q, const float* G, const float* Gp, const float* Gn, int32_t* idx_out,
                                     float* coarse_out, float* ws, int B, int N, int D, void* stream) {
    NEAT_REQUIRE(q && G && Gp && Gn && idx_out && coarse_out && ws, NEAT_ERR_ARG, "neat_coarse_align_i2t: null pointer");
    NEAT_REQUIRE(B > 0 && N > 0 && D > 0 && D % SG_K == 0, NEAT_ERR_SHAPE, "neat_coarse_align_i2t: D=%d must be a multiple of 16", D);
    const int ntn = cdiv(N, SG_T);
    float* pval = ws;
    int* pidx = reinterpret_cast<int*>(ws + (long)B * ntn);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(coarse_i2t_tile_kernel, dim3(ntn, cdiv(B, SG_T)), dim3(256), 0, s, q, Gp, Gn, pval, pidx, B, N, D, ntn);
    NEAT_LAUNCH_CHECK("neat_coarse_align_i2t(tile)");
    hipLaunchKernelGGL(coarse_final_kernel, dim3(B), dim3(64), 0, s, pval, pidx, G, idx_out, coarse_out, ntn, D);
    NEAT_LAUNCH_CHECK("neat_coarse_align_i2t(final)");
    return NEAT_OK;
}

// ---- neat_coarse_align without a caller workspace and without any allocation (for callers that cannot allocate:
// code under HIP graph capture, tight serving loops).  The per-tile (max, argmax) partials live in a file-scope device
// array; every GPU the library runs on holds its own copy at its own address, looked up once per device.
constexpr long COARSE_STATIC_SLOTS = 1L << 18;               // B * ceil(N / 64) <= 262 144 (e.g. B = 2048, N = 8192)
static __device__ __align__(256) float scratch[2 * COARSE_STATIC_SLOTS];   // 2 MiB per GPU, held until the process ends
constexpr int MAX_DEVICES = 64;
static float* g_scratch_ptr[MAX_DEVICES] = {};               // device address of `scratch` per device (NOT the symbol)

static int coarse_scratch(float** out) {
    int dev = -1;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    NEAT_REQUIRE(dev >= 0 && dev < MAX_DEVICES, NEAT_ERR_ARG, "neat_coarse_align_static: device %d out of range", dev);
    float* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        // once per device, with that device current; racing first calls look up the same address
        rc = check_hip(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
        if (rc) return rc;
        NEAT_REQUIRE(p != nullptr, NEAT_ERR_HIP, "neat_coarse_align_static: null scratch address on device %d", dev);
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
    *out = p;
    return NEAT_OK;
}

extern "C" long neat_coarse_static_max_slots(void) { return COARSE_STATIC_SLOTS; }

// set-up of the allocation-free entry point for the CURRENT device: looks up the device address of the scratch array
// (hipGetSymbolAddress is a synchronous lookup that may load the code object: not something to do while a stream is
// capturing).  Idempotent; neat_coarse_align_static does the same on its first call on a device.
extern "C" int neat_coarse_static_init(void) {
    float* ws = nullptr;
    return coarse_scratch(&ws);
}

extern "C" int neat_coarse_align_static(const float* q, const float* pos, const float* neg, const float* G, float* S_out,
                                        int32_t* idx_out, float* coarse_out, int B, int N, int D, void* stream) {
    NEAT_REQUIRE(q && pos && neg && G && S_out && idx_out && coarse_out, NEAT_ERR_ARG,
                 "neat_coarse_align_static: null pointer");
    NEAT_REQUIRE(B > 0 && N > 0 && D > 0 && D % SG_K == 0, NEAT_ERR_SHAPE,
                 "neat_coarse_align_static: D=%d must be a multiple of 16", D);
    const long slots = (long)B * cdiv(N, SG_T);
    NEAT_REQUIRE(slots <= COARSE_STATIC_SLOTS, NEAT_ERR_SHAPE,
                 "neat_coarse_align_static: B * ceil(N/64) = %ld exceeds the static scratch of %ld slots "
                 "(use neat_coarse_align with a caller workspace)", slots, COARSE_STATIC_SLOTS);
    float* ws = nullptr;
    int rc = coarse_scratch(&ws);
    if (rc) return rc;
    // every slot the final kernel reads is written by the tile kernel of THIS call: nothing needs clearing
    return neat_coarse_align(q, pos, neg, G, S_out, idx_out, coarse_out, ws, B, N, D, stream);
}

extern "C" long neat_loss_ws_floats(int B, int Bl) { return (long)B * B + (long)Bl * B + B + 3L * Bl; }

extern "C" int neat_tta_loss_fwd_bwd(const float* text, const float* pos, const float* inv, const float* text_all,
                                     const float* coarse_all, float T1, float T2, float* losses, float* dtext,
                                     float* dpos, float* dinv, float* ws, int B, int Bl, int row0, int D, void* stream) {
    NEAT_REQUIRE(text && pos && inv && text_all && coarse_all && losses && dtext && dpos && dinv && ws, NEAT_ERR_ARG,
                 "neat_tta_loss_fwd_bwd: null pointer");
    NEAT_REQUIRE(B > 0 && Bl > 0 && row0 >= 0 && row0 + Bl <= B && D > 0 && D % SG_K == 0 && T1 > 0 && T2 > 0,
                 NEAT_ERR_SHAPE, "neat_tta_loss_fwd_bwd: bad shape B=%d Bl=%d row0=%d D=%d", B, Bl, row0, D);
    hipStream_t s = (hipStream_t)stream;
    float* Z = ws;
    float* U = Z + (long)B * B;
    float* Hrow = U + (long)Bl * B;
    float* srm = Hrow + B;
    int* srj = reinterpret_cast<int*>(srm + Bl);
    float* trirow = srm + 2 * Bl;
    // Z[j,i] = coarse_j . text_i ; U[i,j] = inv_i . coarse_j -- both products in ONE launch
    const SgProb pz{coarse_all, text_all, Z, B, B, D, D, D}, pu{inv, coarse_all, U, Bl, B, D, D, D};
    launch_sg<false, false>(pz, &pu, s);
    hipLaunchKernelGGL(loss_rows_kernel, dim3(B + Bl), dim3(256), 0, s, Z, Hrow, U, srm, srj, B, row0, 1.0f / T1);
    // dtext <- (dZ^T coarse) for the local rows: a GEMM with both operands k-major (k = the B rows of dZ / coarse), so the
    // walk down a column of dZ is a coalesced slab load instead of one strided read per output element
    launch_sg<true, true>(SgProb{Z + row0, coarse_all, dtext, Bl, D, B, B, D}, nullptr, s);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(Bl), dim3(256), 0, s, text, pos, inv, coarse_all, srj, dtext, dpos, dinv,
                       trirow, B, row0, D, 1.0f / T1, 1.0f / T2);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, Hrow, srm, trirow, losses, B, Bl, 1.0f / T2);
    NEAT_LAUNCH_CHECK("neat_tta_loss_fwd_bwd");
    return NEAT_OK;
}

extern "C" int neat_adamw_flat(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                               double beta2, double eps, double wd, int step, void* stream) {
    NEAT_REQUIRE(p && g && m && v, NEAT_ERR_ARG, "neat_adamw_flat: null pointer");
    NEAT_REQUIRE(n > 0 && step >= 1, NEAT_ERR_SHAPE, "neat_adamw_flat: n=%ld step=%d", n, step);
    const double bc1 = 1.0 - pow(beta1, step);
    const float sbc2 = (float)sqrt(1.0 - pow(beta2, step));
    long g_ = (n + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)(g_ > 1024 ? 1024 : g_)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - lr * wd), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr / bc1), sbc2);
    NEAT_LAUNCH_CHECK("neat_adamw_flat");
    return NEAT_OK;
}
