// Image preprocessing of the TTA gallery loader on the GPU: what the reference's dataset transform does on the host
// (src/data/__init__.py:14-20): torchvision Resize((S,S), BICUBIC) on a PIL image -> ToTensor -> Normalize.
//
// Resize on a PIL image is Pillow's Image.resize (libImaging/Resample.c), restated here so that the result is
// bit-identical: separable two-pass convolution on 8-bit channels, HORIZONTAL pass first into an 8-bit intermediate,
// then the vertical pass; antialiasing by scaling the filter support with the downscale factor; coefficients
// computed in double precision, normalised, and rounded to 22-bit fixed point; accumulation in int32 starting from
// 1 << 21, arithmetic shift, saturation to 0..255.  ToTensor / Normalize are three fp32 operations per value
// (u / 255, - mean, / std), each correctly rounded -- the same three roundings torch does on the host.
//
// HBM-bound byte work: an image is read once (3 B/pixel), the intermediate (H x S x 3 B) is written and re-read from
// L2, the result leaves as fp32 CHW planes with full-line stores.  Three small launches per image:
//   1. resample_coeffs_kernel: one thread per output index, Pillow's precompute_coeffs in the same operation order
//      (explicit __dmul_rn / __dadd_rn / __ddiv_rn: no FMA contraction, like the generic x86-64 build of Pillow);
//   2. resample_h_kernel: thread per (row, output column), 3 channels;
//   3. resample_v_norm_kernel: thread per (output row, output column), 3 channels, fused ToTensor + Normalize.
#include "common.h"

namespace neat {

constexpr int IMG_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) {
        // ((a + 2) * x - (a + 3)) * x * x + 1
        const double t = __dsub_rn(__dmul_rn(a + 2.0, x), a + 3.0);
        return __dadd_rn(__dmul_rn(__dmul_rn(t, x), x), 1.0);
    }
    if (x < 2.0) {
        // (((x - 5) * x + 8) * x - 4) * a
        const double t = __dadd_rn(__dmul_rn(__dsub_rn(x, 5.0), x), 8.0);
        return __dmul_rn(__dsub_rn(__dmul_rn(t, x), 4.0), a);
    }
    return 0.0;
}

// bounds[xx] = (first input index, tap count); kk[xx*ksize + x] = fixed-point weight; ksize = ceil(support)*2 + 1
__global__ void resample_coeffs_kernel(int in_size, int out_size, int ksize, int2* __restrict__ bounds,
                                       int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);      // C cast: truncation toward zero
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = kk + (long)xx * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        // filter((x + xmin - center + 0.5) * ss)
        const double arg = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
        ww = __dadd_rn(ww, bicubic_filter(arg));
    }
    for (int x = 0; x < ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            const double arg = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
            double w = bicubic_filter(arg);
            if (ww != 0.0) w = __ddiv_rn(w, ww);
            const double f = __dmul_rn(w, (double)(1 << IMG_PRECISION_BITS));
            v = w < 0.0 ? (int)__dadd_rn(-0.5, f) : (int)__dadd_rn(0.5, f);
        }
        k[x] = v;
    }
    bounds[xx] = make_int2(xmin, xmax);
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> IMG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// src u8 [H, W, 3] -> tmp u8 [H, S, 3]
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int H, int W, int S,
                                  int ksize, const int2* __restrict__ bounds, const int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (xx >= S) return;
    const int2 b = bounds[xx];
    const int* k = kk + (long)xx * ksize;
    const uint8_t* row = src + ((long)y * W + b.x) * 3;
    int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < b.y; ++x) {
        const int w = k[x];
        s0 += row[3 * x + 0] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    uint8_t* o = tmp + ((long)y * S + xx) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// in u8 [Hin, S, 3] -> dst f32 [3, S, S], value = ((u / 255) - mean[c]) / std[c]; Hin == S: no vertical pass
__global__ void resample_v_norm_kernel(const uint8_t* __restrict__ in, float* __restrict__ dst, int Hin, int S, int ksize,
                                       const int2* __restrict__ bounds, const int* __restrict__ kk, float m0, float m1,
                                       float m2, float d0, float d1, float d2) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= S) return;
    int u0, u1, u2;
    if (bounds == nullptr) {
        const uint8_t* p = in + ((long)yy * S + xx) * 3;
        u0 = p[0], u1 = p[1], u2 = p[2];
    } else {
        const int2 b = bounds[yy];
        const int* k = kk + (long)yy * ksize;
        int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < b.y; ++y) {
            const uint8_t* p = in + ((long)(b.x + y) * S + xx) * 3;
            const int w = k[y];
            s0 += p[0] * w;
            s1 += p[1] * w;
            s2 += p[2] * w;
        }
        u0 = clip8(s0), u1 = clip8(s1), u2 = clip8(s2);
    }
    const long plane = (long)S * S, o = (long)yy * S + xx;
    dst[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u0, 255.0f), m0), d0);
    dst[plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u1, 255.0f), m1), d1);
    dst[2 * plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u2, 255.0f), m2), d2);
}

// ---- batched forms: blockIdx.z = image; every image has its own slice of the workspace with a uniform layout
// (tables sized for the largest image of the batch), so that one launch set serves a gallery batch of mixed sizes ----
struct ImgBatch {
    const uint8_t* const* srcs;   // [n] device pointers, u8 [H_i, W_i, 3]
    const int* hw;                // [n][2]
    char* ws;                     // n slices of ws_stride bytes
    long ws_stride;
    long off_bh, off_kh, off_bv, off_kv, off_tmp;   // byte offsets inside a slice
    int kmax_h, kmax_v, S;
};

__device__ __forceinline__ int img_ksize_dev(int in_size, int out_size) {
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(__dmul_rn(2.0, filterscale)) * 2 + 1;
}

__device__ __forceinline__ void coeffs_one(int in_size, int out_size, int xx, int kstride, int2* bounds, int* kk) {
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = kk + (long)xx * kstride;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x)
        ww = __dadd_rn(ww, bicubic_filter(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss)));
    for (int x = 0; x < xmax; ++x) {
        double w = bicubic_filter(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
        if (ww != 0.0) w = __ddiv_rn(w, ww);
        const double f = __dmul_rn(w, (double)(1 << IMG_PRECISION_BITS));
        k[x] = w < 0.0 ? (int)__dadd_rn(-0.5, f) : (int)__dadd_rn(0.5, f);
    }
    bounds[xx] = make_int2(xmin, xmax);
}

__global__ void resample_coeffs_batch_kernel(ImgBatch b) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= b.S) return;
    const int img = blockIdx.z, axis = blockIdx.y;                // axis 0: horizontal (W -> S), 1: vertical (H -> S)
    const int in_size = b.hw[2 * img + (axis == 0 ? 1 : 0)];
    if (in_size == b.S) return;                                   // that pass is skipped
    char* ws = b.ws + (long)img * b.ws_stride;
    if (axis == 0) coeffs_one(in_size, b.S, xx, b.kmax_h, (int2*)(ws + b.off_bh), (int*)(ws + b.off_kh));
    else coeffs_one(in_size, b.S, xx, b.kmax_v, (int2*)(ws + b.off_bv), (int*)(ws + b.off_kv));
}

__global__ void resample_h_batch_kernel(ImgBatch b) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, img = blockIdx.z;
    const int H = b.hw[2 * img], W = b.hw[2 * img + 1], S = b.S;
    if (xx >= S || y >= H || W == S) return;
    char* ws = b.ws + (long)img * b.ws_stride;
    const int2 bd = ((const int2*)(ws + b.off_bh))[xx];
    const int* k = (const int*)(ws + b.off_kh) + (long)xx * b.kmax_h;
    const uint8_t* row = b.srcs[img] + ((long)y * W + bd.x) * 3;
    int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < bd.y; ++x) {
        const int w = k[x];
        s0 += row[3 * x + 0] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    uint8_t* o = (uint8_t*)(ws + b.off_tmp) + ((long)y * S + xx) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

__global__ void resample_v_norm_batch_kernel(ImgBatch b, float* __restrict__ dst, float m0, float m1, float m2, float d0,
                                             float d1, float d2) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y, img = blockIdx.z;
    const int H = b.hw[2 * img], W = b.hw[2 * img + 1], S = b.S;
    if (xx >= S) return;
    char* ws = b.ws + (long)img * b.ws_stride;
    const uint8_t* in = (W == S) ? b.srcs[img] : (const uint8_t*)(ws + b.off_tmp);   // [H, S, 3] either way
    int u0, u1, u2;
    if (H == S) {
        const uint8_t* p = in + ((long)yy * S + xx) * 3;
        u0 = p[0], u1 = p[1], u2 = p[2];
    } else {
        const int2 bd = ((const int2*)(ws + b.off_bv))[yy];
        const int* k = (const int*)(ws + b.off_kv) + (long)yy * b.kmax_v;
        int s0 = 1 << (IMG_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < bd.y; ++y) {
            const uint8_t* p = in + ((long)(bd.x + y) * S + xx) * 3;
            const int w = k[y];
            s0 += p[0] * w;
            s1 += p[1] * w;
            s2 += p[2] * w;
        }
        u0 = clip8(s0), u1 = clip8(s1), u2 = clip8(s2);
    }
    const long plane = (long)S * S, o = (long)img * 3 * plane + (long)yy * S + xx;
    dst[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u0, 255.0f), m0), d0);
    dst[plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u1, 255.0f), m1), d1);
    dst[2 * plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u2, 255.0f), m2), d2);
}

static int img_ksize(int in_size, int out_size) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}
static long align256(long v) { return (v + 255) / 256 * 256; }

}  // namespace neat

using namespace neat;

// workspace layout: [bounds_h int2[S]] [kk_h int[S*ksize_h]] [bounds_v int2[S]] [kk_v int[S*ksize_v]] [tmp u8 [H,S,3]]
extern "C" long neat_image_preprocess_ws_bytes(int H, int W, int S) {
    if (H <= 0 || W <= 0 || S <= 0) return 0;
    return align256((long)S * 8) + align256((long)S * img_ksize(W, S) * 4) + align256((long)S * 8) +
           align256((long)S * img_ksize(H, S) * 4) + align256((long)H * S * 3);
}

extern "C" int neat_image_preprocess(const uint8_t* src, int H, int W, float* dst, int S, float mean0, float mean1,
                                     float mean2, float std0, float std1, float std2, void* workspace, void* stream) {
    NEAT_REQUIRE(src && dst && workspace, NEAT_ERR_ARG, "neat_image_preprocess: null pointer");
    NEAT_REQUIRE(H > 0 && W > 0 && S > 0 && H <= 32768 && W <= 32768 && S <= 4096, NEAT_ERR_SHAPE,
                 "neat_image_preprocess: H=%d W=%d S=%d unsupported", H, W, S);
    // Pillow's own limit: a tap count that would overflow the fixed-point accumulator never occurs below ~2^9 taps
    NEAT_REQUIRE(img_ksize(W, S) <= 1024 && img_ksize(H, S) <= 1024, NEAT_ERR_SHAPE,
                 "neat_image_preprocess: downscale factor too large (H=%d W=%d -> %d)", H, W, S);
    hipStream_t s = (hipStream_t)stream;
    const int kh = img_ksize(W, S), kv = img_ksize(H, S);
    char* ws = (char*)workspace;
    int2* bounds_h = (int2*)ws;            ws += align256((long)S * 8);
    int* kk_h = (int*)ws;                  ws += align256((long)S * kh * 4);
    int2* bounds_v = (int2*)ws;            ws += align256((long)S * 8);
    int* kk_v = (int*)ws;                  ws += align256((long)S * kv * 4);
    uint8_t* tmp = (uint8_t*)ws;
    const dim3 blk(64);
    const uint8_t* hsrc = src;             // input of the vertical pass
    if (W != S) {                          // Pillow skips a pass whose size does not change
        hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(S, 64)), blk, 0, s, W, S, kh, bounds_h, kk_h);
        hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(S, 64), H), blk, 0, s, src, tmp, H, W, S, kh, bounds_h, kk_h);
        hsrc = tmp;
    }
    if (H != S) {
        hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(S, 64)), blk, 0, s, H, S, kv, bounds_v, kk_v);
        hipLaunchKernelGGL(resample_v_norm_kernel, dim3(cdiv(S, 64), S), blk, 0, s, hsrc, dst, H, S, kv, bounds_v, kk_v,
                           mean0, mean1, mean2, std0, std1, std2);
    } else {
        hipLaunchKernelGGL(resample_v_norm_kernel, dim3(cdiv(S, 64), S), blk, 0, s, hsrc, dst, H, S, kv,
                           (const int2*)nullptr, (const int*)nullptr, mean0, mean1, mean2, std0, std1, std2);
    }
    NEAT_LAUNCH_CHECK("neat_image_preprocess");
    return NEAT_OK;
}

// Batched form: n images of mixed sizes in one launch set.  srcs = DEVICE array of n device pointers (u8 [H_i, W_i, 3]),
// hw = DEVICE int32 [n][2]; max_h / max_w = the largest H_i / W_i (host values, they size the grid and the tables);
// workspace = n * neat_image_preprocess_ws_bytes(max_h, max_w, S) bytes.  dst f32 [n, 3, S, S].  Bit-identical to n
// single calls.
extern "C" int neat_image_preprocess_batch(const uint8_t* const* srcs, const int32_t* hw, int n, int max_h, int max_w,
                                           float* dst, int S, float mean0, float mean1, float mean2, float std0,
                                           float std1, float std2, void* workspace, void* stream) {
    NEAT_REQUIRE(srcs && hw && dst && workspace, NEAT_ERR_ARG, "neat_image_preprocess_batch: null pointer");
    NEAT_REQUIRE(n > 0 && n <= 65535 && max_h > 0 && max_w > 0 && S > 0 && max_h <= 32768 && max_w <= 32768 && S <= 4096,
                 NEAT_ERR_SHAPE, "neat_image_preprocess_batch: n=%d max_h=%d max_w=%d S=%d unsupported", n, max_h, max_w, S);
    // images smaller than S on an axis are up-scaled: their tap count (5) never exceeds the table of the largest image
    const int kh = img_ksize(max_w > S ? max_w : S, S), kv = img_ksize(max_h > S ? max_h : S, S);
    NEAT_REQUIRE(kh <= 1024 && kv <= 1024, NEAT_ERR_SHAPE, "neat_image_preprocess_batch: downscale factor too large");
    ImgBatch b;
    b.srcs = srcs, b.hw = hw, b.ws = (char*)workspace, b.S = S, b.kmax_h = kh, b.kmax_v = kv;
    b.off_bh = 0;
    b.off_kh = b.off_bh + align256((long)S * 8);
    b.off_bv = b.off_kh + align256((long)S * kh * 4);
    b.off_kv = b.off_bv + align256((long)S * 8);
    b.off_tmp = b.off_kv + align256((long)S * kv * 4);
    b.ws_stride = neat_image_preprocess_ws_bytes(max_h > S ? max_h : S, max_w > S ? max_w : S, S);
    hipStream_t s = (hipStream_t)stream;
    const dim3 blk(64);
    hipLaunchKernelGGL(resample_coeffs_batch_kernel, dim3(cdiv(S, 64), 2, n), blk, 0, s, b);
    hipLaunchKernelGGL(resample_h_batch_kernel, dim3(cdiv(S, 64), max_h, n), blk, 0, s, b);
    hipLaunchKernelGGL(resample_v_norm_batch_kernel, dim3(cdiv(S, 64), S, n), blk, 0, s, b, dst, mean0, mean1, mean2, std0,
                       std1, std2);
    NEAT_LAUNCH_CHECK("neat_image_preprocess_batch");
    return NEAT_OK;
}
