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

constexpr int SG_T = 64;   // output tile
constexpr int SG_K = 16;   // k slab
constexpr int SG_LD = SG_T + 4;

// loads a [64 x 16] slab of a K-contiguous matrix (rows clamped) and stores it k-major into LDS
__device__ __forceinline__ void sg_load(const float* __restrict__ src, int rows, int K, int r0, int k0, float* dst) {
    const int r = threadIdx.x >> 2, kq = threadIdx.x & 3;
    int gr = r0 + r;
    gr = gr < rows ? gr : rows - 1;
    const float4 v = *reinterpret_cast<const float4*>(src + (long)gr * K + k0 + kq * 4);
    dst[(kq * 4 + 0) * SG_LD + r] = v.x;
    dst[(kq * 4 + 1) * SG_LD + r] = v.y;
    dst[(kq * 4 + 2) * SG_LD + r] = v.z;
    dst[(kq * 4 + 3) * SG_LD + r] = v.w;
}

__device__ __forceinline__ void sg_fma(const float* As, const float* Bs, float (&acc)[4][4], int ty, int tx) {
#pragma unroll
    for (int k = 0; k < SG_K; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(As + k * SG_LD + ty * 4);
        const float4 b = *reinterpret_cast<const float4*>(Bs + k * SG_LD + tx * 4);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
}

// C[M,N] = A[M,K] . B[N,K]^T
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[SG_K * SG_LD], Bs[SG_K * SG_LD];
    const int m0 = blockIdx.y * SG_T, n0 = blockIdx.x * SG_T;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += SG_K) {
        sg_load(A, M, K, m0, k0, As);
        sg_load(B, N, K, n0, k0, Bs);
        __syncthreads();
        sg_fma(As, Bs, acc, ty, tx);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) C[(long)m * N + n] = acc[i][j];
        }
    }
}

// Fused coarse alignment tile: S = q G^T (stored), A = (pos G^T)(1 - relu(neg G^T)) -> per-row
// (max, first argmax) of this 64-column tile.
__global__ __launch_bounds__(256) void coarse_tile_kernel(const float* __restrict__ q, const float* __restrict__ pos,
                                                          const float* __restrict__ neg, const float* __restrict__ G,
                                                          float* __restrict__ S, float* __restrict__ pval,
                                                          int* __restrict__ pidx, int B, int N, int D, int ntn) {
    __shared__ __attribute__((aligned(16))) float Qs[SG_K * SG_LD], Ps[SG_K * SG_LD], Ns[SG_K * SG_LD], Gs[SG_K * SG_LD];
    const int m0 = blockIdx.y * SG_T, n0 = blockIdx.x * SG_T;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float aq[4][4] = {}, ap[4][4] = {}, an[4][4] = {};
    for (int k0 = 0; k0 < D; k0 += SG_K) {
        sg_load(q, B, D, m0, k0, Qs);
        sg_load(pos, B, D, m0, k0, Ps);
        sg_load(neg, B, D, m0, k0, Ns);
        sg_load(G, N, D, n0, k0, Gs);
        __syncthreads();
        sg_fma(Qs, Gs, aq, ty, tx);
        sg_fma(Ps, Gs, ap, ty, tx);
        sg_fma(Ns, Gs, an, ty, tx);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) {
                if (m < B) S[(long)m * N + n] = aq[i][j];
                const float a = ap[i][j] * (1.0f - fmaxf(an[i][j], 0.0f));
                if (a > best) { best = a; bi = n; }
            }
        }
        // reduce over the 16 lanes (tx) that share this row; ties -> lowest index
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 16);
            const int oi = __shfl_xor(bi, o, 16);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (tx == 0 && m < B) {
            pval[(long)m * ntn + blockIdx.x] = best;
            pidx[(long)m * ntn + blockIdx.x] = bi;
        }
    }
}

__global__ void coarse_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                    const float* __restrict__ G, int32_t* __restrict__ idx_out,
                                    float* __restrict__ coarse, int ntn, int D) {
    __shared__ int sidx;
    const int m = blockIdx.x;
    if (threadIdx.x == 0) {
        float best = -INFINITY;
        int bi = 0;
        bool any = false;
        for (int t = 0; t < ntn; ++t) {
            const float v = pval[(long)m * ntn + t];
            if (!any || v > best) { best = v; bi = pidx[(long)m * ntn + t]; any = true; }
        }
        if (bi == 0x7fffffff) bi = 0;  // all-NaN row: torch returns the NaN position; never hit with finite features
        sidx = bi;
        idx_out[m] = bi;
    }
    __syncthreads();
    const int j = sidx;
    for (int d = threadIdx.x; d < D; d += blockDim.x) coarse[(long)m * D + d] = G[(long)j * D + d];
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
__global__ __launch_bounds__(256) void entropy_row_kernel(float* __restrict__ Z, float* __restrict__ Hrow, int B,
                                                          float invT1) {
    __shared__ float red[8];
    float* z = Z + (long)blockIdx.x * B;
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
    if (threadIdx.x == 0) Hrow[blockIdx.x] = h;
}

// local row i of U (= inv_i . coarse_j): hinge max and its first argmax
__global__ __launch_bounds__(256) void sr_row_kernel(const float* __restrict__ U, float* __restrict__ srm,
                                                     int* __restrict__ srj, int B, int row0) {
    __shared__ float rv[8];
    __shared__ int ri[8];
    const int i = blockIdx.x, gi = row0 + i;
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

// gradients for local row i: dtext, dpos, dinv; triplet row term
__global__ __launch_bounds__(256) void loss_grad_kernel(
    const float* __restrict__ text, const float* __restrict__ pos, const float* __restrict__ inv,
    const float* __restrict__ coarse_all, const float* __restrict__ dZ, const int* __restrict__ srj,
    float* __restrict__ dtext, float* __restrict__ dpos, float* __restrict__ dinv, float* __restrict__ trirow, int B,
    int row0, int D, float invT1, float invT2) {
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
        float acc = 0.f;
        for (int j = 0; j < B; ++j) acc = fmaf(dZ[(long)j * B + gi], coarse_all[(long)j * D + d], acc);
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

int launch_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t s) {
    NEAT_REQUIRE(A && B && C, NEAT_ERR_ARG, "gemm_f32_nt: null pointer");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && K % SG_K == 0, NEAT_ERR_SHAPE, "gemm_f32_nt: K=%d must be a multiple of 16", K);
    hipLaunchKernelGGL(gemm_f32_nt_kernel, dim3(cdiv(N, SG_T), cdiv(M, SG_T)), dim3(256), 0, s, A, B, C, M, N, K);
    NEAT_LAUNCH_CHECK("gemm_f32_nt");
    return NEAT_OK;
}

}  // namespace neat

using namespace neat;

extern "C" int neat_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
    return launch_gemm_f32_nt(A, B, C, M, N, K, (hipStream_t)stream);
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
    hipLaunchKernelGGL(coarse_final_kernel, dim3(B), dim3(128), 0, s, pval, pidx, G, idx_out, coarse_out, ntn, D);
    NEAT_LAUNCH_CHECK("neat_coarse_align(final)");
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
    // Z[j,i] = coarse_j . text_i ; U[i,j] = inv_i . coarse_j
    hipLaunchKernelGGL(gemm_f32_nt_kernel, dim3(cdiv(B, SG_T), cdiv(B, SG_T)), dim3(256), 0, s, coarse_all, text_all, Z, B,
                       B, D);
    hipLaunchKernelGGL(gemm_f32_nt_kernel, dim3(cdiv(B, SG_T), cdiv(Bl, SG_T)), dim3(256), 0, s, inv, coarse_all, U, Bl, B,
                       D);
    hipLaunchKernelGGL(entropy_row_kernel, dim3(B), dim3(256), 0, s, Z, Hrow, B, 1.0f / T1);
    hipLaunchKernelGGL(sr_row_kernel, dim3(Bl), dim3(256), 0, s, U, srm, srj, B, row0);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(Bl), dim3(256), 0, s, text, pos, inv, coarse_all, Z, srj, dtext, dpos, dinv,
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
