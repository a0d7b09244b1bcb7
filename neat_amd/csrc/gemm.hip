// bf16 MFMA GEMM with fused epilogues for the encoder linears (MFMA-bound kernels).
// Replaces nn.Linear / nn.MultiheadAttention projections (reference open_clip/transformer.py:209-213,
// 227-229) and their data-gradient GEMMs (weights are frozen: there is no wgrad on this path).
//
//   out[M,N] = epilogue( A[M,K] . W[N,K]^T )      A = activations, W = weights, both K-contiguous
//
// CDNA4 mapping
//  * 128x128 block tile, BK = 64, 4 wavefronts (2x2), each owning a 64x64 sub-tile as 4x4
//    v_mfma_f32_16x16x32_bf16 accumulators.
//  * W is the MFMA *A* operand and the activations the *B* operand, so an accumulator register
//    quad holds 4 CONSECUTIVE output columns n for one row m: bias / residual / saved-activation
//    loads and the output stores of the epilogue are 8-16 B per lane.
//  * global -> LDS by global_load_lds_dwordx4 (no VGPR round trip), double buffered; the LDS image
//    is lane-linear (1 KiB per wave-instruction), the bank-conflict XOR swizzle is applied to the
//    per-lane SOURCE address and to the ds_read_b128 address (chunk ^= (row>>1)&7 on 128-B rows).
//  * 1-D grid remapped so that the 8 XCDs each walk a contiguous range of tiles (tiles that share
//    an activation row-panel hit the same L2).
//  * rows >= M / weight rows >= N are clamped on load and masked on store: no padding contract.
#include "common.h"

namespace neat {

constexpr int G_BM = 128, G_BN = 128, G_BK = 64;
constexpr int G_THREADS = 256;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;            // 16 KiB per operand tile
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;          // A + W
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;           // double buffer = 64 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Stage one 128x64 bf16 tile: 1024 16-B chunks, 4 per thread.  LDS chunk q=(row r=q>>3, c=q&7)
// receives global chunk c ^ ((r>>1)&7) of row r.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, long ld, int row0, int nrows_total,
                                           int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = p * 256 + wave * 64 + lane;
        const int r = q >> 3;
        const int c = (q & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows_total ? gr : nrows_total - 1;
        const bf16_t* g = src + (long)gr * ld + k0 + c * 8;
        char* l = lds_tile + (p * 256 + wave * 64) * 16;  // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* lds_tile, int row, int chunk) {
    const int c = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + c * 16);
}

template <int EPI>
__global__ __launch_bounds__(G_THREADS) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const float* res, const bf16_t* __restrict__ aux, void* out0, void* out1, int M, int N, int K, int ntm,
    int ntn) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware bijective remap of the 1-D grid (blocks b and b+8 share an XCD)
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int m0 = tm * G_BM, n0 = tn * G_BN;

    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / G_BK;
    stage_tile(A, K, m0, M, 0, lds, wave, lane);
    stage_tile(W, K, n0, N, 0, lds + G_TILE_BYTES, wave, lane);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the stage

    const int fr = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        char* cur = lds + (t & 1) * G_STAGE_BYTES;
        if (t + 1 < nk) {
            char* nxt = lds + ((t + 1) & 1) * G_STAGE_BYTES;
            stage_tile(A, K, m0, M, (t + 1) * G_BK, nxt, wave, lane);
            stage_tile(W, K, n0, N, (t + 1) * G_BK, nxt + G_TILE_BYTES, wave, lane);
        }
        const char* At = cur;
        const char* Wt = cur + G_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = lds_frag(Wt, wn * 64 + i * 16 + fr, kk * 4 + fq);
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = lds_frag(At, wm * 64 + j * 16 + fr, kk * 4 + fq);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds, for row m, the 4 consecutive columns n .. n+3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            if (n >= N) continue;
            const long o = (long)m * N + n;
            f32x4 v = acc[i][j];
            if constexpr (EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_BIAS_GELU) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16) {
                *reinterpret_cast<uint2*>((bf16_t*)out0 + o) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            } else if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                const float4 r = *reinterpret_cast<const float4*>(res + o);
                *reinterpret_cast<float4*>((float*)out0 + o) = make_float4(v[0] + r.x, v[1] + r.y, v[2] + r.z, v[3] + r.w);
            } else if constexpr (EPI == NEAT_EPI_BIAS_GELU) {
                // the saved pre-activation is the bf16-rounded value; GELU is applied to the same
                // rounded value so that forward and backward see one h
                const uint2 hp = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                if (out0) *reinterpret_cast<uint2*>((bf16_t*)out0 + o) = hp;
                const float h0 = __uint_as_float(hp.x << 16), h1 = __uint_as_float(hp.x & 0xffff0000u);
                const float h2 = __uint_as_float(hp.y << 16), h3 = __uint_as_float(hp.y & 0xffff0000u);
                *reinterpret_cast<uint2*>((bf16_t*)out1 + o) =
                    make_uint2(pack2bf(quick_gelu(h0), quick_gelu(h1)), pack2bf(quick_gelu(h2), quick_gelu(h3)));
            } else if constexpr (EPI == NEAT_EPI_GELU_BWD) {
                const uint2 hp = *reinterpret_cast<const uint2*>(aux + o);
                const float h0 = __uint_as_float(hp.x << 16), h1 = __uint_as_float(hp.x & 0xffff0000u);
                const float h2 = __uint_as_float(hp.y << 16), h3 = __uint_as_float(hp.y & 0xffff0000u);
                *reinterpret_cast<uint2*>((bf16_t*)out0 + o) =
                    make_uint2(pack2bf(v[0] * quick_gelu_grad(h0), v[1] * quick_gelu_grad(h1)),
                               pack2bf(v[2] * quick_gelu_grad(h2), v[3] * quick_gelu_grad(h3)));
            } else {  // NEAT_EPI_F32
                *reinterpret_cast<float4*>((float*)out0 + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int EPI>
static int launch_gemm(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                       const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s) {
    const int ntm = cdiv(M, G_BM), ntn = cdiv(N, G_BN);
    static bool attr_set = false;  // per-template-instance: allow 64 KiB dynamic LDS
    if (!attr_set) {
        int rc = check_hip(hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES),
                           "hipFuncSetAttribute(gemm)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(ntm * ntn), dim3(G_THREADS), G_LDS_BYTES, s, A, W, bias, res,
                       aux, out0, out1, M, N, K, ntm, ntn);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16");
    return NEAT_OK;
}

}  // namespace neat

using namespace neat;

extern "C" int neat_gemm_bf16(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                              const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                              void* stream) {
    NEAT_REQUIRE(A && W, NEAT_ERR_ARG, "neat_gemm_bf16: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && K % G_BK == 0 && N % 16 == 0, NEAT_ERR_SHAPE,
                 "neat_gemm_bf16: M=%d N=%d K=%d unsupported (need K %% 64 == 0, N %% 16 == 0)", M, N, K);
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
