// Small HBM-bound kernels around the encoders: token embedding, EOT argmax, pooled head
// (final LayerNorm + projection + L2-normalise) forward/backward, ViT stem helpers, casts.
#include "common.h"

namespace neat {

// ---- nn.Embedding + positional add (reference open_clip/model.py:272-274) ----
__global__ void embed_fwd_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                 const float* __restrict__ pos_emb, float* __restrict__ x, long rows, int L, int C,
                                 int vocab) {
    const int c4 = C >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * c4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c4;
        const int c = (int)(i - row * c4);
        long id = tokens[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float4 a = reinterpret_cast<const float4*>(tok_emb + id * C)[c];
        const float4 p = reinterpret_cast<const float4*>(pos_emb + (row % L) * C)[c];
        reinterpret_cast<float4*>(x + row * C)[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// ---- text.argmax(dim=-1) (reference open_clip/transformer.py:555-558); first maximum wins ----
// one wavefront per sequence: coalesced 8-B loads, then a butterfly arg-max that keeps the lowest position on ties
__global__ __launch_bounds__(256) void eot_index_kernel(const int64_t* __restrict__ tokens, int32_t* __restrict__ eot, int nseq, int L) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseq) return;
    const int64_t* t = tokens + (long)s * L;
    int64_t best = INT64_MIN;
    int bi = 0x7fffffff;
    for (int l = lane; l < L; l += 64) {
        const int64_t v = t[l];
        if (bi == 0x7fffffff || v > best) { best = v; bi = l; }   // l ascending per lane: first max kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) eot[s] = bi;
}

// ---- packed ("varlen") text batches -------------------------------------------------------------------
// With a causal mask and EOT pooling (open_clip/transformer.py:555-558, 655-661) no position after the EOT token
// can influence the pooled feature or receive a gradient from it, so those rows are dead work.  The packed
// layout keeps rows 0..eot of every sequence back to back: cu[s] = first row of sequence s, cu[nseq] = rows.
// cu = exclusive prefix sum of (eot + 1), one block: every thread sums a contiguous run of sequences, the 256 run totals
// are scanned in LDS, every thread then writes its run.  (The EOT positions come from eot_index_kernel, one wavefront
// per sequence with coalesced loads: one thread walking a 77-token row per sequence took 96 us for 1 024 captions.)
__global__ __launch_bounds__(256) void pack_scan_kernel(const int32_t* __restrict__ eot, int32_t* __restrict__ cu, int nseq) {
    __shared__ int32_t tot[256];
    const int per = (nseq + 255) / 256;
    const int s0 = threadIdx.x * per, s1 = min(s0 + per, nseq);
    int acc = 0;
    for (int s = s0; s < s1; ++s) acc += eot[s] + 1;
    tot[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = tot[i]; tot[i] = run; run += v; }
        cu[nseq] = run;
    }
    __syncthreads();
    acc = tot[threadIdx.x];
    for (int s = s0; s < s1; ++s) { cu[s] = acc; acc += eot[s] + 1; }
}

// one block per sequence: rows cu[s] .. cu[s+1]-1 <- token embedding + positional embedding
__global__ __launch_bounds__(256) void embed_packed_kernel(const int64_t* __restrict__ tokens,
                                                           const int32_t* __restrict__ cu,
                                                           const float* __restrict__ tok_emb,
                                                           const float* __restrict__ pos_emb, float* __restrict__ x, int L,
                                                           int C, int vocab) {
    const int s = blockIdx.x;
    const int r0 = cu[s], len = cu[s + 1] - r0;
    const int c4 = C >> 2;
    for (int i = threadIdx.x; i < len * c4; i += blockDim.x) {
        const int pos = i / c4, c = i - pos * c4;
        long id = tokens[(long)s * L + pos];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float4 a = reinterpret_cast<const float4*>(tok_emb + id * C)[c];
        const float4 p = reinterpret_cast<const float4*>(pos_emb + (long)pos * C)[c];
        reinterpret_cast<float4*>(x + (long)(r0 + pos) * C)[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// ---- pooled head: pooled row -> LayerNorm (fp32) -> @ proj [C,D] -> optional F.normalize
// (reference open_clip/model.py:278-286, open_clip/transformer.py:538-542, model.py:267), forward and data
// gradient.  Three launches each way: a wave-per-sequence gather + LayerNorm, the exact-fp32 tile GEMM of
// loss.hip for the projection (a per-sequence mat-vec re-read the 1 MB projection once per sequence), and a
// wave-per-sequence normalise.
constexpr int PP_THREADS = 256;   // 4 sequences per block

__global__ __launch_bounds__(PP_THREADS) void pool_ln_fwd_kernel(const float* __restrict__ x,
                                                                 const int32_t* __restrict__ row_idx,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ z,
                                                                 float* __restrict__ xe, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int nseq, int L, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (PP_THREADS / 64) + (threadIdx.x >> 6);
    if (s >= nseq) return;
    const long row = (long)s * L + (row_idx ? row_idx[s] : 0);
    const float* xr = x + row * C;
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += xr[c];
    const float mu = wave_sum(a) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mu;
        q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    for (int c = lane; c < C; c += 64) {
        const float v = xr[c];
        if (xe) xe[(long)s * C + c] = v;
        z[(long)s * C + c] = (v - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0) {
        if (mean) mean[s] = mu;
        if (rstd) rstd[s] = rs;
    }
}

// in place: feat[s] /= max(||feat[s]||, 1e-12) (iff normalize); nrm[s] = ||feat[s]|| before the division
__global__ __launch_bounds__(PP_THREADS) void l2norm_rows_kernel(float* __restrict__ feat, float* __restrict__ nrm, int nseq,
                                                                 int D, int normalize) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (PP_THREADS / 64) + (threadIdx.x >> 6);
    if (s >= nseq) return;
    float* f = feat + (long)s * D;
    float n2 = 0.f;
    for (int d = lane; d < D; d += 64) n2 = fmaf(f[d], f[d], n2);
    const float nr = sqrtf(wave_sum(n2));
    if (normalize) {
        const float inv = 1.0f / fmaxf(nr, 1e-12f);
        for (int d = lane; d < D; d += 64) f[d] *= inv;
    }
    if (lane == 0 && nrm) nrm[s] = nr;
}

// df = (dfeat - feat <dfeat, feat>) / max(nrm, 1e-12)
__global__ __launch_bounds__(PP_THREADS) void l2norm_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat,
                                                                const float* __restrict__ nrm, float* __restrict__ df, int n,
                                                                int D) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (PP_THREADS / 64) + (threadIdx.x >> 6);
    if (s >= n) return;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot = fmaf(dfeat[(long)s * D + d], feat[(long)s * D + d], dot);
    dot = wave_sum(dot);
    const float inv = 1.0f / fmaxf(nrm[s], 1e-12f);
    for (int d = lane; d < D; d += 64) df[(long)s * D + d] = (dfeat[(long)s * D + d] - feat[(long)s * D + d] * dot) * inv;
}

// dz -> LayerNorm data gradient -> scattered to the pooled row (ln_final / ln_post are not adapted)
__global__ __launch_bounds__(PP_THREADS) void pool_ln_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ xe,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma,
                                                                 const int32_t* __restrict__ row_idx, float* __restrict__ dxf,
                                                                 bf16_t* __restrict__ dxb, int n, int L, int C) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (PP_THREADS / 64) + (threadIdx.x >> 6);
    if (s >= n) return;
    const float mu = mean[s], rs = rstd[s];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = dz[(long)s * C + c] * gamma[c];
        s1 += g;
        s2 += g * ((xe[(long)s * C + c] - mu) * rs);
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    const long row = (long)s * L + (row_idx ? row_idx[s] : 0);
    for (int c = lane; c < C; c += 64) {
        const float g = dz[(long)s * C + c] * gamma[c];
        const float xh = (xe[(long)s * C + c] - mu) * rs;
        const float o = rs * (g - s1 - xh * s2);
        if (dxf) dxf[row * C + c] = o;
        if (dxb) dxb[row * C + c] = f2bf(o);
    }
}

// ---- ViT stem ----
// images f32 NCHW -> bf16 patch matrix [n*gh*gw, Kpad], column order (c, kh, kw) = conv1.weight.view(width,-1)
// One workgroup per patch row (image b, patch ph, pw); a thread moves V consecutive kw of one (c, kh) line: a V x 4-byte
// load from the image and a V x 2-byte store (V = 4 when patch % 4 == 0 -- ViT-B/32, B/16 --, else 2 -- ViT-L/14: patch 14).
// (One element per thread with 64-bit div / mod per element ran at 0.9 TB/s: 172 us for 250 images.)
template <int V>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int H, int W,
                                                       int patch, int Kpad) {
    const int gh = H / patch, gw = W / patch;
    const int K = 3 * patch * patch;
    const long prow = blockIdx.x;
    const int pw = (int)(prow % gw), ph = (int)((prow / gw) % gh);
    const long b = prow / ((long)gw * gh);
    const float* src = img + (b * 3 * H + (long)ph * patch) * W + pw * patch;
    bf16_t* dst = out + prow * Kpad;
    const int pv = patch / V;                                      // vectors per (c, kh) line
    for (int v = threadIdx.x; v < Kpad / V; v += 256) {
        const int k = v * V;
        float x[V];
#pragma unroll
        for (int j = 0; j < V; ++j) x[j] = 0.f;
        if (k < K) {
            const int line = v / pv, kw = (v - line * pv) * V;     // line = c * patch + kh
            const int c = line / patch, kh = line - c * patch;
            const float* p = src + ((long)c * H + kh) * W + kw;
            if constexpr (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
            } else {
                const float2 t = *reinterpret_cast<const float2*>(p);
                x[0] = t.x; x[1] = t.y;
            }
        }
        if constexpr (V == 4) *reinterpret_cast<uint2*>(dst + k) = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
        else *reinterpret_cast<uint32_t*>(dst + k) = pack2bf(x[0], x[1]);
    }
}

// any other geometry (odd patch sizes, unaligned rows): one element per thread
__global__ void patchify_scalar_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int n, int H, int W, int patch,
                                       int Kpad) {
    const int gh = H / patch, gw = W / patch;
    const int K = 3 * patch * patch;
    const long total = (long)n * gh * gw * Kpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const long prow = i / Kpad;
        float v = 0.f;
        if (k < K) {
            const int kw = k % patch, kh = (k / patch) % patch, c = k / (patch * patch);
            const int pw = (int)(prow % gw), ph = (int)((prow / gw) % gh);
            const long b = prow / ((long)gw * gh);
            v = img[((b * 3 + c) * H + ph * patch + kh) * (long)W + pw * patch + kw];
        }
        out[i] = f2bf(v);
    }
}

// x[b,0] = cls + pos[0]; x[b,1+p] = patch_emb[b,p] + pos[1+p]; then ln_pre.  One wave per token row.
__global__ __launch_bounds__(256) void vision_embed_kernel(const float* __restrict__ patch_emb,
                                                           const float* __restrict__ cls,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ x,
                                                           long rows, int P, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = P + 1;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const long b = row / T;
        const int t = (int)(row - b * T);
        const float* src = t == 0 ? cls : patch_emb + (b * P + (t - 1)) * (long)C;
        const float* pr = pos + (long)t * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += src[c] + pr[c];
        const float mu = wave_sum(s) / C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = src[c] + pr[c] - mu;
            q += d * d;
        }
        const float rs = rsqrtf(wave_sum(q) / C + eps);
        for (int c = lane; c < C; c += 64) x[row * C + c] = (src[c] + pr[c] - mu) * rs * gamma[c] + beta[c];
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// VideoCLIP pooling (reference utils/video_utils/model.py:62-74,90-91): mean over frames, re-normalise
__global__ void frame_mean_norm_kernel(const float* __restrict__ feat, float* __restrict__ out, int F, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* red = sm + D;
    const int v = blockIdx.x;
    float n2 = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int f = 0; f < F; ++f) a += feat[((long)v * F + f) * D + d];
        a /= F;
        sm[d] = a;
        n2 += a * a;
    }
    const float nr = sqrtf(block_sum(n2, red));
    const float inv = 1.0f / fmaxf(nr, 1e-12f);
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(long)v * D + d] = sm[d] * inv;
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = f2bf(src[i]);
}

// src f32 [R,C] -> dst bf16 [C,R]; 32x32 LDS tile
__global__ void transpose_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) dst[(long)c * R + r] = f2bf(tile[threadIdx.x][j]);
    }
}

static inline int ew_grid(long n, int threads) {
    long g = (n + threads - 1) / threads;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace neat

using namespace neat;

extern "C" int neat_embed_fwd(const int64_t* tokens, const float* tok_emb, const float* pos_emb, float* x, int nseq,
                              int L, int C, int vocab, void* stream) {
    NEAT_REQUIRE(tokens && tok_emb && pos_emb && x, NEAT_ERR_ARG, "neat_embed_fwd: null pointer");
    NEAT_REQUIRE(nseq > 0 && L > 0 && C > 0 && C % 4 == 0 && vocab > 0, NEAT_ERR_SHAPE, "neat_embed_fwd: bad shape");
    const long rows = (long)nseq * L;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(ew_grid(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                       tok_emb, pos_emb, x, rows, L, C, vocab);
    NEAT_LAUNCH_CHECK("neat_embed_fwd");
    return NEAT_OK;
}

extern "C" int neat_eot_index(const int64_t* tokens, int32_t* eot, int nseq, int L, void* stream) {
    NEAT_REQUIRE(tokens && eot, NEAT_ERR_ARG, "neat_eot_index: null pointer");
    NEAT_REQUIRE(nseq > 0 && L > 0, NEAT_ERR_SHAPE, "neat_eot_index: bad shape");
    hipLaunchKernelGGL(eot_index_kernel, dim3(cdiv(nseq, 4)), dim3(256), 0, (hipStream_t)stream, tokens, eot, nseq, L);
    NEAT_LAUNCH_CHECK("neat_eot_index");
    return NEAT_OK;
}

extern "C" int neat_pack_plan(const int64_t* tokens, int32_t* eot, int32_t* cu_seqlens, int nseq, int L, void* stream) {
    NEAT_REQUIRE(tokens && eot && cu_seqlens, NEAT_ERR_ARG, "neat_pack_plan: null pointer");
    NEAT_REQUIRE(nseq > 0 && nseq <= 12000 && L > 0, NEAT_ERR_SHAPE, "neat_pack_plan: nseq=%d L=%d (nseq <= 12000)", nseq, L);
    hipLaunchKernelGGL(eot_index_kernel, dim3(cdiv(nseq, 4)), dim3(256), 0, (hipStream_t)stream, tokens, eot, nseq, L);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, eot, cu_seqlens, nseq);
    NEAT_LAUNCH_CHECK("neat_pack_plan");
    return NEAT_OK;
}

extern "C" int neat_embed_packed_fwd(const int64_t* tokens, const int32_t* cu_seqlens, const float* tok_emb,
                                     const float* pos_emb, float* x, int nseq, int L, int C, int vocab, void* stream) {
    NEAT_REQUIRE(tokens && cu_seqlens && tok_emb && pos_emb && x, NEAT_ERR_ARG, "neat_embed_packed_fwd: null pointer");
    NEAT_REQUIRE(nseq > 0 && L > 0 && C > 0 && C % 4 == 0 && vocab > 0, NEAT_ERR_SHAPE, "neat_embed_packed_fwd: bad shape");
    hipLaunchKernelGGL(embed_packed_kernel, dim3(nseq), dim3(256), 0, (hipStream_t)stream, tokens, cu_seqlens, tok_emb,
                       pos_emb, x, L, C, vocab);
    NEAT_LAUNCH_CHECK("neat_embed_packed_fwd");
    return NEAT_OK;
}

extern "C" long neat_pool_proj_ws_floats(int nseq, int C, int D) { return (long)nseq * (C + D); }

extern "C" int neat_pool_proj_fwd(const float* x, const int32_t* row_idx, const float* gamma, const float* beta,
                                  const float* proj_t, float* feat, float* xe, float* mean, float* rstd, float* nrm,
                                  float* ws, int nseq, int L, int C, int D, int normalize, float eps, void* stream) {
    NEAT_REQUIRE(x && gamma && beta && proj_t && feat && ws, NEAT_ERR_ARG, "neat_pool_proj_fwd: null pointer");
    NEAT_REQUIRE(nseq > 0 && (L > 0 || (L == 0 && row_idx)) && C > 0 && D > 0 && C % 16 == 0, NEAT_ERR_SHAPE,
                 "neat_pool_proj_fwd: bad shape C=%d D=%d (C %% 16 == 0)", C, D);
    hipStream_t s = (hipStream_t)stream;
    const int grid = cdiv(nseq, PP_THREADS / 64);
    float* z = ws;
    hipLaunchKernelGGL(pool_ln_fwd_kernel, dim3(grid), dim3(PP_THREADS), 0, s, x, row_idx, gamma, beta, z, xe, mean, rstd,
                       nseq, L, C, eps);
    int rc = launch_gemm_f32_nt(z, proj_t, feat, nseq, D, C, D, s);
    if (rc) return rc;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid), dim3(PP_THREADS), 0, s, feat, nrm, nseq, D, normalize);
    NEAT_LAUNCH_CHECK("neat_pool_proj_fwd");
    return NEAT_OK;
}

extern "C" int neat_pool_proj_bwd(const float* dfeat, const float* feat, const float* nrm, const float* xe,
                                  const float* mean, const float* rstd, const float* gamma, const float* proj,
                                  const int32_t* row_idx, float* dx_f32, uint16_t* dx_bf16, float* ws, int nseq, int L,
                                  int C, int D, int normalize, void* stream) {
    NEAT_REQUIRE(dfeat && xe && mean && rstd && gamma && proj && ws && (dx_f32 || dx_bf16), NEAT_ERR_ARG,
                 "neat_pool_proj_bwd: null pointer");
    NEAT_REQUIRE(!normalize || (feat && nrm), NEAT_ERR_ARG, "neat_pool_proj_bwd: normalize needs feat, nrm");
    NEAT_REQUIRE(nseq > 0 && (L > 0 || (L == 0 && row_idx)) && C > 0 && D > 0 && D % 16 == 0, NEAT_ERR_SHAPE,
                 "neat_pool_proj_bwd: bad shape (D %% 16 == 0)");
    hipStream_t s = (hipStream_t)stream;
    const int grid = cdiv(nseq, PP_THREADS / 64);
    float* df = ws;
    float* dz = ws + (long)nseq * D;
    const float* dfp = dfeat;
    if (normalize) {
        hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid), dim3(PP_THREADS), 0, s, dfeat, feat, nrm, df, nseq, D);
        dfp = df;
    }
    int rc = launch_gemm_f32_nt(dfp, proj, dz, nseq, C, D, C, s);     // dz[n,C] = df[n,D] . proj[C,D]^T
    if (rc) return rc;
    hipLaunchKernelGGL(pool_ln_bwd_kernel, dim3(grid), dim3(PP_THREADS), 0, s, dz, xe, mean, rstd, gamma, row_idx, dx_f32,
                       dx_bf16, nseq, L, C);
    NEAT_LAUNCH_CHECK("neat_pool_proj_bwd");
    return NEAT_OK;
}

extern "C" int neat_patchify(const float* images, uint16_t* patches, int n, int H, int W, int patch, int Kpad,
                             void* stream) {
    NEAT_REQUIRE(images && patches, NEAT_ERR_ARG, "neat_patchify: null pointer");
    NEAT_REQUIRE(n > 0 && patch > 0 && H % patch == 0 && W % patch == 0 && Kpad >= 3 * patch * patch, NEAT_ERR_SHAPE,
                 "neat_patchify: bad shape");
    const long total = (long)n * (H / patch) * (W / patch) * Kpad;
    const long prows = (long)n * (H / patch) * (W / patch);
    const bool aligned = (uintptr_t)images % 16 == 0 && (uintptr_t)patches % 8 == 0 && prows < (1L << 31);
    if (aligned && patch % 4 == 0 && W % 4 == 0 && Kpad % 4 == 0)
        hipLaunchKernelGGL(patchify_kernel<4>, dim3((unsigned)prows), dim3(256), 0, (hipStream_t)stream, images, patches, H, W,
                           patch, Kpad);
    else if (aligned && patch % 2 == 0 && W % 2 == 0 && Kpad % 2 == 0)
        hipLaunchKernelGGL(patchify_kernel<2>, dim3((unsigned)prows), dim3(256), 0, (hipStream_t)stream, images, patches, H, W,
                           patch, Kpad);
    else
        hipLaunchKernelGGL(patchify_scalar_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, images,
                           patches, n, H, W, patch, Kpad);
    NEAT_LAUNCH_CHECK("neat_patchify");
    return NEAT_OK;
}

extern "C" int neat_vision_embed(const float* patch_emb, const float* cls, const float* pos, const float* gamma,
                                 const float* beta, float* x, int n, int P, int C, float eps, void* stream) {
    NEAT_REQUIRE(patch_emb && cls && pos && gamma && beta && x, NEAT_ERR_ARG, "neat_vision_embed: null pointer");
    NEAT_REQUIRE(n > 0 && P > 0 && C > 0, NEAT_ERR_SHAPE, "neat_vision_embed: bad shape");
    const long rows = (long)n * (P + 1);
    hipLaunchKernelGGL(vision_embed_kernel, dim3(ew_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, patch_emb, cls,
                       pos, gamma, beta, x, rows, P, C, eps);
    NEAT_LAUNCH_CHECK("neat_vision_embed");
    return NEAT_OK;
}

extern "C" int neat_frame_mean_norm(const float* feat, float* out, int n, int F, int D, void* stream) {
    NEAT_REQUIRE(feat && out, NEAT_ERR_ARG, "neat_frame_mean_norm: null pointer");
    NEAT_REQUIRE(n > 0 && F > 0 && D > 0 && D <= 8192, NEAT_ERR_SHAPE, "neat_frame_mean_norm: bad shape");
    hipLaunchKernelGGL(frame_mean_norm_kernel, dim3(n), dim3(256), (size_t)(D + 8) * 4, (hipStream_t)stream, feat, out, F,
                       D);
    NEAT_LAUNCH_CHECK("neat_frame_mean_norm");
    return NEAT_OK;
}

extern "C" int neat_cast_bf16(const float* src, uint16_t* dst, long n, void* stream) {
    NEAT_REQUIRE(src && dst, NEAT_ERR_ARG, "neat_cast_bf16: null pointer");
    if (n <= 0) return NEAT_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    NEAT_LAUNCH_CHECK("neat_cast_bf16");
    return NEAT_OK;
}

extern "C" int neat_transpose_cast_bf16(const float* src, uint16_t* dst, int R, int C, void* stream) {
    NEAT_REQUIRE(src && dst, NEAT_ERR_ARG, "neat_transpose_cast_bf16: null pointer");
    NEAT_REQUIRE(R > 0 && C > 0, NEAT_ERR_SHAPE, "neat_transpose_cast_bf16: bad shape");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)stream, src,
                       dst, R, C);
    NEAT_LAUNCH_CHECK("neat_transpose_cast_bf16");
    return NEAT_OK;
}
