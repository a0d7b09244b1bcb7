// Video-frame preprocessing of the MSR-VTT gallery loader on the GPU: what the reference's video transform does on the
// host (src/data/__init__.py:22-26): ConvertImageDtype(float32) -> Resize((S,S), BICUBIC, antialias=True) on a TENSOR ->
// Normalize.  Resize on a float tensor is ATen's upsample_bicubic2d_aa (NOT Pillow's 8-bit resampler of image.hip): a
// separable float convolution, support scaled by the downscale factor, cubic a = -0.5, weights normalised to sum 1,
// horizontal pass first, no clamping of the result.
//
// The arithmetic follows oracle/video_oracle.py operation by operation (explicit __fmul_rn / __fadd_rn / __fdiv_rn: no
// FMA contraction), so kernel and restatement agree bit for bit; the restatement agrees with torch's CPU operator
// to 2e-6 on the [0, 1] scale (summation order inside ATen's vectorised loops).
//
// HBM-bound: a clip is read once (3 B/pixel), the fp32 intermediate [T, 3, H, S] is written and re-read (L2 / Infinity
// Cache resident at MSR-VTT sizes), the result leaves as fp32 planes with full-line stores.  Three launches per clip:
// coefficient tables (both axes), horizontal pass, vertical pass + Normalize.
#include "common.h"

namespace neat {

__device__ __forceinline__ float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) {
        // ((a + 2) * x - (a + 3)) * x * x + 1
        const float t = __fsub_rn(__fmul_rn(a + 2.0f, x), a + 3.0f);
        return __fadd_rn(__fmul_rn(__fmul_rn(t, x), x), 1.0f);
    }
    if (x < 2.0f) {
        // (((x - 5) * x + 8) * x - 4) * a
        const float t = __fadd_rn(__fmul_rn(__fsub_rn(x, 5.0f), x), 8.0f);
        return __fmul_rn(__fsub_rn(__fmul_rn(t, x), 4.0f), a);
    }
    return 0.0f;
}

static __host__ __device__ int aa_kmax(int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? 2.0f * scale : 2.0f;
    return (int)ceilf(support) * 2 + 1;
}

// blockIdx.y = axis (0: W -> S, 1: H -> S); bounds[i] = (first tap, tap count); wt[i*kmax + j] = normalised weight
__global__ void aa_coeffs_kernel(int W, int H, int S, int kmax_h, int kmax_v, int2* __restrict__ bounds_h,
                                 float* __restrict__ wt_h, int2* __restrict__ bounds_v, float* __restrict__ wt_v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const int axis = blockIdx.y;
    const int in_size = axis == 0 ? W : H;
    const int kmax = axis == 0 ? kmax_h : kmax_v;
    int2* bounds = axis == 0 ? bounds_h : bounds_v;
    float* wt = (axis == 0 ? wt_h : wt_v) + (long)i * kmax;
    const float scale = __fdiv_rn((float)in_size, (float)S);
    const float support = scale >= 1.0f ? __fmul_rn(2.0f, scale) : 2.0f;
    const float invscale = scale >= 1.0f ? __fdiv_rn(1.0f, scale) : 1.0f;
    const float center = (float)__dmul_rn((double)scale, (double)i + 0.5);
    long lo = (long)__dadd_rn((double)__fsub_rn(center, support), 0.5);
    if (lo < 0) lo = 0;
    long hi = (long)__dadd_rn((double)__fadd_rn(center, support), 0.5);
    if (hi > in_size) hi = in_size;
    const int n = (int)(hi - lo);
    float total = 0.0f;
    for (int j = 0; j < n; ++j) {
        const double arg = __dmul_rn(__dadd_rn((double)__fsub_rn((float)(j + lo), center), 0.5), (double)invscale);
        const float w = cubic_aa((float)arg);
        wt[j] = w;
        total = __fadd_rn(total, w);
    }
    if (total != 0.0f)
        for (int j = 0; j < n; ++j) wt[j] = __fdiv_rn(wt[j], total);
    for (int j = n; j < kmax; ++j) wt[j] = 0.0f;
    bounds[i] = make_int2((int)lo, n);
}

// src u8 [T, H, W, 3] -> tmp f32 [T, 3, H, S]   (value / 255 first, then the horizontal convolution)
__global__ void aa_h_kernel(const uint8_t* __restrict__ src, float* __restrict__ tmp, int H, int W, int S, int kmax,
                            const int2* __restrict__ bounds, const float* __restrict__ wt) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, t = blockIdx.z;
    if (xx >= S) return;
    const int2 b = bounds[xx];
    const float* k = wt + (long)xx * kmax;
    const uint8_t* row = src + (((long)t * H + y) * W + b.x) * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int x = 0; x < b.y; ++x) {
        const float w = k[x];
        const float p0 = __fmul_rn(__fdiv_rn((float)row[3 * x + 0], 255.0f), w);
        const float p1 = __fmul_rn(__fdiv_rn((float)row[3 * x + 1], 255.0f), w);
        const float p2 = __fmul_rn(__fdiv_rn((float)row[3 * x + 2], 255.0f), w);
        s0 = x == 0 ? p0 : __fadd_rn(s0, p0);
        s1 = x == 0 ? p1 : __fadd_rn(s1, p1);
        s2 = x == 0 ? p2 : __fadd_rn(s2, p2);
    }
    const long plane = (long)H * S;
    float* o = tmp + (long)t * 3 * plane + (long)y * S + xx;
    o[0] = s0;
    o[plane] = s1;
    o[2 * plane] = s2;
}

// tmp f32 [T, 3, H, S] -> dst f32 [T, 3, S, S], value = (v - mean[c]) / std[c]
__global__ void aa_v_norm_kernel(const float* __restrict__ tmp, float* __restrict__ dst, int H, int S, int kmax,
                                 const int2* __restrict__ bounds, const float* __restrict__ wt, float m0, float m1,
                                 float m2, float d0, float d1, float d2) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y, t = blockIdx.z;
    if (xx >= S) return;
    const int2 b = bounds[yy];
    const float* k = wt + (long)yy * kmax;
    const long plane = (long)H * S;
    const float* in = tmp + (long)t * 3 * plane + (long)b.x * S + xx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int y = 0; y < b.y; ++y) {
        const float w = k[y];
        const float p0 = __fmul_rn(in[(long)y * S], w);
        const float p1 = __fmul_rn(in[plane + (long)y * S], w);
        const float p2 = __fmul_rn(in[2 * plane + (long)y * S], w);
        s0 = y == 0 ? p0 : __fadd_rn(s0, p0);
        s1 = y == 0 ? p1 : __fadd_rn(s1, p1);
        s2 = y == 0 ? p2 : __fadd_rn(s2, p2);
    }
    const long oplane = (long)S * S, o = (long)t * 3 * oplane + (long)yy * S + xx;
    dst[o] = __fdiv_rn(__fsub_rn(s0, m0), d0);
    dst[oplane + o] = __fdiv_rn(__fsub_rn(s1, m1), d1);
    dst[2 * oplane + o] = __fdiv_rn(__fsub_rn(s2, m2), d2);
}

static long falign256(long v) { return (v + 255) / 256 * 256; }

}  // namespace neat

using namespace neat;

// workspace: [bounds_h int2[S]] [wt_h f32[S*kmax_h]] [bounds_v int2[S]] [wt_v f32[S*kmax_v]] [tmp f32 [T,3,H,S]]
extern "C" long neat_frames_preprocess_ws_bytes(int T, int H, int W, int S) {
    if (T <= 0 || H <= 0 || W <= 0 || S <= 0) return 0;
    return falign256((long)S * 8) + falign256((long)S * aa_kmax(W, S) * 4) + falign256((long)S * 8) +
           falign256((long)S * aa_kmax(H, S) * 4) + falign256((long)T * 3 * H * S * 4);
}

extern "C" int neat_frames_preprocess(const uint8_t* src, int T, int H, int W, float* dst, int S, float mean0, float mean1,
                                      float mean2, float std0, float std1, float std2, void* workspace, void* stream) {
    NEAT_REQUIRE(src && dst && workspace, NEAT_ERR_ARG, "neat_frames_preprocess: null pointer");
    NEAT_REQUIRE(T > 0 && T <= 65535 && H > 0 && W > 0 && S > 0 && H <= 32768 && W <= 32768 && S <= 4096, NEAT_ERR_SHAPE,
                 "neat_frames_preprocess: T=%d H=%d W=%d S=%d unsupported", T, H, W, S);
    const int kh = aa_kmax(W, S), kv = aa_kmax(H, S);
    NEAT_REQUIRE(kh <= 1024 && kv <= 1024, NEAT_ERR_SHAPE, "neat_frames_preprocess: downscale factor too large (H=%d W=%d -> %d)",
                 H, W, S);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int2* bounds_h = (int2*)ws;            ws += falign256((long)S * 8);
    float* wt_h = (float*)ws;              ws += falign256((long)S * kh * 4);
    int2* bounds_v = (int2*)ws;            ws += falign256((long)S * 8);
    float* wt_v = (float*)ws;              ws += falign256((long)S * kv * 4);
    float* tmp = (float*)ws;
    const dim3 blk(64);
    hipLaunchKernelGGL(aa_coeffs_kernel, dim3(cdiv(S, 64), 2), blk, 0, s, W, H, S, kh, kv, bounds_h, wt_h, bounds_v, wt_v);
    hipLaunchKernelGGL(aa_h_kernel, dim3(cdiv(S, 64), H, T), blk, 0, s, src, tmp, H, W, S, kh, bounds_h, wt_h);
    hipLaunchKernelGGL(aa_v_norm_kernel, dim3(cdiv(S, 64), S, T), blk, 0, s, tmp, dst, H, S, kv, bounds_v, wt_v, mean0, mean1,
                       mean2, std0, std1, std2);
    NEAT_LAUNCH_CHECK("neat_frames_preprocess");
    return NEAT_OK;
}
