// Fused LayerNorm forward / backward for the adapted LN layers (HBM-bound kernels).
// Replaces F.layer_norm at reference open_clip/transformer.py:24-30 and its autograd backward; the
// gamma/beta gradients are the ONLY parameter gradients of the whole TTA step
// (open_clip/factory.py:395-437).
//
// Mapping: one 64-lane wavefront per row, each lane owns float4 column chunks {lane, lane+64, ...}
// (16 B per lane per access, 1 KiB per wave-instruction).  Row statistics use wave shuffles; the
// column sums for dgamma/dbeta stay in the owning lane's registers across the wave's rows, are
// combined across the block's 4 waves through an LDS tree and leave as one partial row per block;
// neat_ln_bwd_finalize adds the partial rows in fixed order (deterministic, no float atomics).
#include "common.h"

namespace neat {
// streaming accesses: the residual stream and its gradient are read once here and next touched a whole layer later, so they
// are loaded / stored non-temporally (measured on MI355X: a pure streaming read reaches 7.0 TB/s with non-temporal loads
// against 6.3 TB/s without, tools/read_probe.hip); LN outputs that the next GEMM reads at once keep the default policy
__device__ __forceinline__ float4 ld_nt(const float4* p) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ uint2 ld_nt(const uint2* p) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return make_uint2(t[0], t[1]);
}
__device__ __forceinline__ void st_nt(float4* p, float4 v) {
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
}


constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / WAVE;
// backward: 8 waves per block, so that half as many per-block partial rows reach the finalize pass
constexpr int LNB_THREADS = 512;
constexpr int LNB_WAVES = LNB_THREADS / WAVE;

template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            bf16_t* __restrict__ y, float* __restrict__ yf,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int cols,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = cols >> 2;  // float4 chunks per row
    float4 g[NCH], b[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + 64 * i;
        if (c < nch) {
            g[i] = reinterpret_cast<const float4*>(gamma)[c];
            b[i] = reinterpret_cast<const float4*>(beta)[c];
        }
    }
    const float inv = 1.0f / (float)cols;
    for (long row = (long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long)gridDim.x * LN_WAVES) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                v[i] = ld_nt(xr + c);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mu = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rs = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                float o0 = v[i].x * rs * g[i].x + b[i].x;
                float o1 = v[i].y * rs * g[i].y + b[i].y;
                float o2 = v[i].z * rs * g[i].z + b[i].z;
                float o3 = v[i].w * rs * g[i].w + b[i].w;
                if (y) {
                    uint2 pk = make_uint2(pack2bf(o0, o1), pack2bf(o2, o3));
                    reinterpret_cast<uint2*>(y + row * cols)[c] = pk;
                }
                if (yf) reinterpret_cast<float4*>(yf + row * cols)[c] = make_float4(o0, o1, o2, o3);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mu;
            if (rstd_out) rstd_out[row] = rs;
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(LNB_THREADS) void ln_bwd_kernel(
    const bf16_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* dres, float* dxf,
    bf16_t* __restrict__ dxb, float* __restrict__ partial, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = cols >> 2;
    float4 g[NCH], ag[NCH], ab[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + 64 * i;
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) g[i] = reinterpret_cast<const float4*>(gamma)[c];
    }
    const float inv = 1.0f / (float)cols;
    for (long row = (long)blockIdx.x * LNB_WAVES + wave; row < rows; row += (long)gridDim.x * LNB_WAVES) {
        const float mu = mean[row], rs = rstd[row];
        const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
        const uint2* dyr = reinterpret_cast<const uint2*>(dy + row * cols);
        float4 xh[NCH], gg[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                float4 xv = ld_nt(xr + c);
                uint2 d = ld_nt(dyr + c);
                float d0 = __uint_as_float(d.x << 16), d1 = __uint_as_float(d.x & 0xffff0000u);
                float d2 = __uint_as_float(d.y << 16), d3 = __uint_as_float(d.y & 0xffff0000u);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                ag[i].x += d0 * xh[i].x; ag[i].y += d1 * xh[i].y; ag[i].z += d2 * xh[i].z; ag[i].w += d3 * xh[i].w;
                ab[i].x += d0; ab[i].y += d1; ab[i].z += d2; ab[i].w += d3;
                gg[i] = make_float4(d0 * g[i].x, d1 * g[i].y, d2 * g[i].z, d3 * g[i].w);
                s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
                s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
            }
        }
        if (!dxf && !dxb) continue;                              // parameter gradients only (uniform for the whole grid)
        const float m1 = wave_sum(s1) * inv;
        const float m2 = wave_sum(s2) * inv;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                float4 o;
                o.x = rs * (gg[i].x - m1 - xh[i].x * m2);
                o.y = rs * (gg[i].y - m1 - xh[i].y * m2);
                o.z = rs * (gg[i].z - m1 - xh[i].z * m2);
                o.w = rs * (gg[i].w - m1 - xh[i].w * m2);
                if (dres) {
                    float4 r = ld_nt(reinterpret_cast<const float4*>(dres + row * cols) + c);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                if (dxf) st_nt(reinterpret_cast<float4*>(dxf + row * cols) + c, o);
                if (dxb) reinterpret_cast<uint2*>(dxb + row * cols)[c] = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
            }
        }
    }
    if (partial == nullptr) return;  // uniform for the whole grid
    // LDS tree across the block's waves: [wave][2][cols]
    extern __shared__ __attribute__((aligned(16))) float red[];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + 64 * i;
        if (c < nch) {
            reinterpret_cast<float4*>(red + (wave * 2 + 0) * cols)[c] = ag[i];
            reinterpret_cast<float4*>(red + (wave * 2 + 1) * cols)[c] = ab[i];
        }
    }
    __syncthreads();
    for (int stride = LNB_WAVES / 2; stride > 0; stride >>= 1) {
        if (wave < stride) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                int c = lane + 64 * i;
                if (c < nch) {
                    for (int t = 0; t < 2; ++t) {
                        float4* a = reinterpret_cast<float4*>(red + (wave * 2 + t) * cols) + c;
                        float4 o = reinterpret_cast<float4*>(red + ((wave + stride) * 2 + t) * cols)[c];
                        float4 m = *a;
                        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
                        *a = m;
                    }
                }
            }
        }
        __syncthreads();
    }
    float* dst = partial + (long)blockIdx.x * 2 * cols;
    for (int c = threadIdx.x; c < 2 * cols; c += LNB_THREADS) dst[c] = red[c];
}

// 64 columns x 16 row-slices per block: coalesced 256-B reads, fixed summation order (deterministic)
constexpr int LNF_SLICES = 16;
__global__ __launch_bounds__(64 * LNF_SLICES) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                         int cols, float* dgamma, float* dbeta,
                                                                         int accumulate) {
    __shared__ float red[LNF_SLICES][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < 2 * cols)
        for (int b = slice; b < nblk; b += LNF_SLICES) s += partial[(long)b * 2 * cols + c];
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && c < 2 * cols) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_SLICES; ++k) s += red[k][lane];
        float* dst = (c < cols) ? dgamma + c : dbeta + (c - cols);
        *dst = accumulate ? (*dst + s) : s;
    }
}

// the same reduction for `count` LayerNorms at once: blockIdx.y = which; dst[2*which] = dgamma, dst[2*which+1] = dbeta
__global__ __launch_bounds__(64 * LNF_SLICES) void ln_bwd_finalize_batched_kernel(const float* __restrict__ partial,
                                                                                 int nblk, int cols,
                                                                                 float* const* __restrict__ dst,
                                                                                 int accumulate) {
    __shared__ float red[LNF_SLICES][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const float* src = partial + (long)blockIdx.y * nblk * 2 * cols;
    float s = 0.f;
    if (c < 2 * cols)
        for (int b = slice; b < nblk; b += LNF_SLICES) s += src[(long)b * 2 * cols + c];
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && c < 2 * cols) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_SLICES; ++k) s += red[k][lane];
        float* d = (c < cols) ? dst[2 * blockIdx.y] + c : dst[2 * blockIdx.y + 1] + (c - cols);
        *d = accumulate ? (*d + s) : s;
    }
}

static int ln_grid(int rows) {
    int g = cdiv(rows, LN_WAVES);
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

}  // namespace neat

using namespace neat;

extern "C" int neat_ln_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* y_f32,
                           float* mean, float* rstd, int rows, int cols, float eps, void* stream) {
    NEAT_REQUIRE(x && gamma && beta && (y || y_f32), NEAT_ERR_ARG, "neat_ln_fwd: null pointer");
    NEAT_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, NEAT_ERR_SHAPE,
                 "neat_ln_fwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    if (rows == 0) return NEAT_OK;
    hipStream_t s = (hipStream_t)stream;
    int nch = cdiv(cols / 4, 64);
    dim3 grid(ln_grid(rows)), block(LN_THREADS);
    switch (nch) {
        case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, s, x, gamma, beta, y, y_f32, mean, rstd, rows, cols, eps); break;
        case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, s, x, gamma, beta, y, y_f32, mean, rstd, rows, cols, eps); break;
        case 3: hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, block, 0, s, x, gamma, beta, y, y_f32, mean, rstd, rows, cols, eps); break;
        default: hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, s, x, gamma, beta, y, y_f32, mean, rstd, rows, cols, eps); break;
    }
    NEAT_LAUNCH_CHECK("neat_ln_fwd");
    return NEAT_OK;
}

extern "C" int neat_ln_bwd_blocks(int rows) {
    // every block leaves one partial row pair behind; 2 blocks (16 waves) per CU keep the streaming part
    // HBM-bound, and the finalize pass reads at most 512 partial rows.  From 8 192 rows on the grid is the full 512
    // blocks (two rows per wave at least): with eight rows per wave the packed text batches (11 136 rows) ran on 174
    // blocks, every wave walking its rows one memory latency after the other (30 us for 91 MB).
    int g = cdiv(rows, LNB_WAVES * 2);
    return g < 1 ? 1 : (g > 512 ? 512 : g);
}

extern "C" int neat_ln_bwd(const uint16_t* dy, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const float* dres, float* dx_f32, uint16_t* dx_bf16,
                           float* partial, int rows, int cols, void* stream) {
    NEAT_REQUIRE(dy && x && gamma && mean && rstd && (dx_f32 || dx_bf16 || partial), NEAT_ERR_ARG,
                 "neat_ln_bwd: null pointer (at least one of dx_f32, dx_bf16, partial)");
    NEAT_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, NEAT_ERR_SHAPE,
                 "neat_ln_bwd: rows=%d cols=%d unsupported (cols %% 4 == 0, <= 1024)", rows, cols);
    hipStream_t s = (hipStream_t)stream;
    int nch = cdiv(cols / 4, 64);
    dim3 grid(neat_ln_bwd_blocks(rows)), block(LNB_THREADS);
    size_t lds = partial ? (size_t)LNB_WAVES * 2 * cols * sizeof(float) : 0;
    switch (nch) {
        case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, lds, s, dy, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, partial, rows, cols); break;
        case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, lds, s, dy, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, partial, rows, cols); break;
        case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, block, lds, s, dy, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, partial, rows, cols); break;
        default: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, lds, s, dy, x, gamma, mean, rstd, dres, dx_f32, dx_bf16, partial, rows, cols); break;
    }
    NEAT_LAUNCH_CHECK("neat_ln_bwd");
    return NEAT_OK;
}

extern "C" int neat_ln_bwd_finalize(const float* partial, int nblk, int cols, float* dgamma, float* dbeta,
                                    int accumulate, void* stream) {
    NEAT_REQUIRE(partial && dgamma && dbeta, NEAT_ERR_ARG, "neat_ln_bwd_finalize: null pointer");
    NEAT_REQUIRE(nblk > 0 && cols > 0, NEAT_ERR_SHAPE, "neat_ln_bwd_finalize: bad shape");
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(cdiv(2 * cols, 64)), dim3(64 * LNF_SLICES), 0, (hipStream_t)stream,
                       partial, nblk, cols, dgamma, dbeta, accumulate);
    NEAT_LAUNCH_CHECK("neat_ln_bwd_finalize");
    return NEAT_OK;
}

extern "C" int neat_ln_bwd_finalize_batched(const float* partial, int nblk, int cols, int count, float* const* dst,
                                            int accumulate, void* stream) {
    NEAT_REQUIRE(partial && dst, NEAT_ERR_ARG, "neat_ln_bwd_finalize_batched: null pointer");
    NEAT_REQUIRE(nblk > 0 && cols > 0 && count > 0 && count <= 65535, NEAT_ERR_SHAPE,
                 "neat_ln_bwd_finalize_batched: bad shape");
    hipLaunchKernelGGL(ln_bwd_finalize_batched_kernel, dim3(cdiv(2 * cols, 64), count), dim3(64 * LNF_SLICES), 0,
                       (hipStream_t)stream, partial, nblk, cols, dst, accumulate);
    NEAT_LAUNCH_CHECK("neat_ln_bwd_finalize_batched");
    return NEAT_OK;
}
