// Shared device/host helpers for libneat_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neat_hip.h"

namespace neat {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16; NaN stays NaN (a plain cast lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// one v_cvt_pk_bf16_f32: lo -> bits 0..15, hi -> bits 16..31
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// sigmoid(1.702 x) in three instructions (v_mul, v_exp, v_add, v_rcp: hardware exp2 / rcp, ~1 ulp each --
// far inside the bf16 rounding that follows every use).  exp2 -> +inf gives rcp -> 0, the right limit.
__device__ __forceinline__ float sigmoid1702(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896341f)));
}
// QuickGELU, reference open_clip/transformer.py:33-36
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid1702(x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = sigmoid1702(x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// ---- error plumbing (capi.hip) ----
int set_error(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define NEAT_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) return ::neat::set_error((code), __VA_ARGS__); \
    } while (0)

#define NEAT_LAUNCH_CHECK(what) \
    do {                         \
        int _rc = ::neat::check_hip(hipGetLastError(), (what)); \
        if (_rc) return _rc;     \
    } while (0)

// loss.hip: C[M,N] = A[M,K] . B[N,K]^T, exact fp32 FMA chains in k order; K % 16 == 0
int launch_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, long ldc, hipStream_t s);   // ldc: row pitch of C (floats)

// gemm_rowln.hip: the 64 x 512-unit GEMM for N = 512 launches of at most one 128 x 256 unit per CU (fp32 residual and
// plain bf16 epilogues; same packed weights and bit-identical results as gemm_panel_kernel)
bool gemm_wide512_applies(int M, int N, int K, int epilogue);
// gemm.hip: 0 when NEAT_GEMM_WARM_W=0 switched the weight warm-up of the GEMM kernels off (A/B measurements), else 1
int gemm_warm_w_enabled();
int launch_gemm_wide512(int epilogue, const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res, void* out0,
                        int M, int K, hipStream_t s);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: remember it per device (bit d of `done`),
// not per process.  Call before launching `fn` with more than 64 KiB of dynamic LDS.
int ensure_dynamic_lds(const void* fn, int bytes, unsigned long long* done, const char* what);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace neat
