// SHELVED EXPERIMENT (round 1) -- not built, not loaded, not tested by the package.  See README.md in this directory.
// Paste the kernel between the tile256 kernel and the "tile256 entry points" of neat_amd/csrc/gemm.hip and the entry
// points at its end to build it (it uses glds16, num_cus, cdiv and the epilogue conventions of that file).
// =====================================================================================================
// duo: TWO independent 256-thread workgroups per CU (4 waves each, one per SIMD, <= 256 VGPRs), each on its own
// 128 x 256 unit.  What tile256 cannot do inside one workgroup happens between the two: while one writes its
// finished tile (a CU stores ~12 B/clk, so a c_fc epilogue is as long as the main loop) the other has the matrix
// pipe to itself, and a lead of either one grows by itself (whoever enters the other resource first runs faster
// there), so the pair settles out of phase.
//   * both operands by LDS-DMA; K-tile = 32 k = 24 KiB (A 128 rows x 64 B, W 256 rows x 64 B), THREE stages
//     = 72 KiB + 8 KiB fp32 bias copy = the 80 KiB half of a CU's LDS;
//   * per K-tile g and wave: vmcnt(6) (K-tile g landed, g+1 may be in flight) -> workgroup barrier -> 12
//     ds_read_b128 -> 32 MFMAs with the 6 LDS-DMA of K-tile g+2 issued BETWEEN them (an issue there costs ~60
//     cycles, next to ds_reads 150+); the stage being refilled was last read before that barrier;
//   * wave w owns all 128 rows x columns 64w..64w+63 (8 x 4 accumulator tiles); epilogue straight from the
//     accumulators exactly as in tile256 (interleaved column order for bf16 outputs);
//   * weights packed into this kernel's own LDS image (neat_pack_weight_bf16_duo): 16-KiB blocks (cb*(K/32) + kt),
//     row lr = column cb*256 + lr, 4 chunks of 16 B, chunk c stored at c ^ m[(lr>>2)&3], m = {0,3,2,1}
//     (conflict-free for the lane groups of ds_read_b128 on 64-B rows, natural and interleaved row order).
constexpr int U_THREADS = 256;
constexpr int U_A = 8192;                     // A part of a stage: 128 rows x 64 B
constexpr int U_STAGE = U_A + 16384;          // + W part: 256 rows x 64 B
constexpr int U_BIAS = 3 * U_STAGE;
constexpr int U_MAXN = 2048;
constexpr int U_LDS = U_BIAS + U_MAXN * 4;    // 80 KiB

__host__ __device__ constexpr bool duo_shape_ok(int N, int K) { return N % 256 == 0 && N <= U_MAXN && K % 64 == 0; }
__host__ __device__ __forceinline__ int u_swz(int lr) { return (4 - ((lr >> 2) & 3)) & 3; }   // {0,3,2,1}

template <int EPI>
__global__ __launch_bounds__(U_THREADS, 2) void gemm_duo_kernel(
    const bf16_t* __restrict__ A, const char* __restrict__ Wp, const float* __restrict__ bias, const float* res,
    const bf16_t* __restrict__ aux, void* out0, void* out1, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool PERM = !(EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_F32);
    constexpr bool HAS_BIAS = EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_BIAS_GELU;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    const int nrb = (M + 127) >> 7;
    const int ncb = N >> 8;
    const int G_ = gridDim.x;
    const int X = (G_ & 7) == 0 ? 8 : 1;
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G_ / X;
    const int nrx = (nrb - xcd + X - 1) / X;
    const int U = (nrx * ncb - wg + WX - 1) / WX;
    if (U <= 0) return;
    const int nkt = K >> 5;
    const long wcb_bytes = (long)nkt * 16384;
    auto unit_of = [&](int t, int& row0, int& cb) {
        const int v = wg + t * WX;
        const int i = v / ncb;
        cb = v - i * ncb;
        row0 = (xcd + X * i) << 7;
    };
    if constexpr (HAS_BIAS) {
        for (int i = threadIdx.x * 4; i < N; i += U_THREADS * 4)
            *reinterpret_cast<float4*>(lds + U_BIAS + i * 4) = *reinterpret_cast<const float4*>(bias + i);
    }

    // staging: wave w issues A instructions 2w, 2w+1 (16 rows x 64 B each) and W instructions 4w .. 4w+3 (1 KiB each)
    const int st_arow = 32 * wave + (lane >> 2);                              // + 16 q
    const int st_acb = ((lane & 3) ^ u_swz(lane >> 2)) * 16;                  // (row>>2)&3 = (lane>>4)&3 for both q
    int s_t = 0, s_kt = 0, s_row0, s_cb, s_buf = 0;
    unit_of(0, s_row0, s_cb);
    auto stage_one = [&](int j) {                                              // j = 0..5: the wave's 6 instructions
        const uint32_t dst = lds_base + s_buf * U_STAGE;
        if (j < 2) {
            int gr = s_row0 + st_arow + 16 * j;
            gr = gr < M ? gr : M - 1;
            glds16(reinterpret_cast<const char*>(A) + ((long)gr * K + s_kt * 32) * 2 + st_acb,
                   dst + (2 * wave + j) * 1024);
        } else {
            const int q = 4 * wave + (j - 2);
            glds16(Wp + s_cb * wcb_bytes + (long)s_kt * 16384 + q * 1024 + lane * 16, dst + U_A + q * 1024);
        }
    };
    auto stage_advance = [&]() {
        s_buf = s_buf == 2 ? 0 : s_buf + 1;
        if (s_kt + 1 < nkt) {
            ++s_kt;
        } else if (s_t + 1 < U) {
            s_kt = 0;
            ++s_t;
            unit_of(s_t, s_row0, s_cb);
        }
    };

    // fragment read offsets inside a stage
    const int aoff = fr * 64 + ((fq ^ u_swz(fr)) << 4);                       // + 1024 t
    int boff[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int lr = PERM ? wave * 64 + 8 * (fr >> 2) + 4 * ii + (fr & 3) : wave * 64 + ii * 16 + fr;
        boff[ii] = U_A + lr * 64 + ((fq ^ u_swz(lr)) << 4);                   // + 2048 (i >> 1)
    }

#pragma unroll
    for (int j = 0; j < 6; ++j) stage_one(j);
    stage_advance();
#pragma unroll
    for (int j = 0; j < 6; ++j) stage_one(j);
    stage_advance();
    __syncthreads();                                                          // bias copy visible (no DMA waited here)

    f32x4 acc[4][8];
    bf16x8 af[8], bf_[4];
    int rbuf = 0, c_t = 0, c_kt = 0, c_row0, c_cb;
    unit_of(0, c_row0, c_cb);
    const int G = U * nkt;
    for (int g = 0; g < G; ++g) {
        if (c_kt == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (HAS_BIAS) {
                    const int n = (c_cb << 8) + wave * 64 + (PERM ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq);
                    b0 = *reinterpret_cast<const f32x4*>(lds + U_BIAS + n * 4);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[i][t] = b0;
            }
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                      // K-tile g landed (this wave's share)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");                               // ... everybody's; stage (g+2)%3 is free
        __builtin_amdgcn_sched_barrier(0);
        const char* base = lds + rbuf * U_STAGE;
        rbuf = rbuf == 2 ? 0 : rbuf + 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf_[i] = *reinterpret_cast<const bf16x8*>(base + (i >> 1) * 2048 + boff[i & 1]);
#pragma unroll
        for (int t = 0; t < 8; ++t) af[t] = *reinterpret_cast<const bf16x8*>(base + t * 1024 + aoff);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[i], af[t], acc[i][t], 0, 0, 0);
            if (t >= 1 && t <= 6) {                                           // 6 LDS-DMA of K-tile g+2 between the MFMAs
                __builtin_amdgcn_sched_barrier(0);
                stage_one(t - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stage_advance();
        if (++c_kt == nkt) {
            // epilogue straight from the accumulators (as tile256)
            const int mrow = c_row0 + fr;
            if constexpr (PERM) {
                const int ncol = (c_cb << 8) + wave * 64 + 8 * fq;
                auto emit = [&](int t, int h, const uint4* hq) {
                    const int m = mrow + 16 * t;
                    const f32x4 lo = acc[2 * h][t], hi = acc[2 * h + 1][t];
                    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const long o = (long)m * N + ncol + 32 * h;
                    if (m >= M) return;
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
                    } else {
                        const uint32_t hp[4] = {hq->x, hq->y, hq->z, hq->w};
                        uint32_t dp[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                            dp[q] = pack2bf(v[2 * q] * quick_gelu_grad(h0), v[2 * q + 1] * quick_gelu_grad(h1));
                        }
                        *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(dp[0], dp[1], dp[2], dp[3]);
                    }
                };
                if constexpr (EPI == NEAT_EPI_GELU_BWD) {
                    uint4 hq[8][2];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        int m = mrow + 16 * t;
                        m = m < M ? m : M - 1;
#pragma unroll
                        for (int h = 0; h < 2; ++h) hq[t][h] = *reinterpret_cast<const uint4*>(aux + (long)m * N + ncol + 32 * h);
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int h = 0; h < 2; ++h) emit(t, h, &hq[t][h]);
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int h = 0; h < 2; ++h) emit(t, h, nullptr);
                }
            } else {
                const int ncol = (c_cb << 8) + wave * 64 + 4 * fq;
                float4 rr[3][4];
                auto load_res = [&](int t) {
                    int m = mrow + 16 * t;
                    m = m < M ? m : M - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) rr[t % 3][i] = *reinterpret_cast<const float4*>(res + (long)m * N + ncol + 16 * i);
                };
                if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                    load_res(0);
                    load_res(1);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                        if (t + 2 < 8) load_res(t + 2);
                    }
                    const int m = mrow + 16 * t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 a4 = acc[i][t];
                        float4 o4 = make_float4(a4[0], a4[1], a4[2], a4[3]);
                        if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                            const float4 r4 = rr[t % 3][i];
                            o4.x += r4.x, o4.y += r4.y, o4.z += r4.z, o4.w += r4.w;
                        }
                        if (m < M) *reinterpret_cast<float4*>((float*)out0 + (long)m * N + ncol + 16 * i) = o4;
                    }
                }
            }
            c_kt = 0;
            if (++c_t < U) unit_of(c_t, c_row0, c_cb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // no LDS-DMA may land after the workgroup ends
}

// f32 weights -> bf16 in the duo LDS image (see above)
__global__ void pack_weight_duo_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K, int transposed) {
    const long total = (long)N * K / 8;
    const int nkt = K >> 5;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const int pos = (int)(u & 1023);                                      // 1024 chunks per 16-KiB block
        const long blk = u >> 10;
        const int kt = (int)(blk % nkt), cb = (int)(blk / nkt);
        const int lr = pos >> 2;
        const int c = (pos & 3) ^ u_swz(lr);
        const int n = cb * 256 + lr, k0 = kt * 32 + c * 8;
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
static int launch_gemm_duo(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                           const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        int rc = check_hip(hipFuncSetAttribute((const void*)gemm_duo_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, U_LDS),
                           "hipFuncSetAttribute(gemm_duo)");
        if (rc) return rc;
        attr_set = true;
    }
    const long units = (long)cdiv(M, 128) * (N / 256);
    const int cus = num_cus();
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < 2 * cus ? want : 2 * cus);
    hipLaunchKernelGGL((gemm_duo_kernel<EPI>), dim3(grid), dim3(U_THREADS), U_LDS, s, A, reinterpret_cast<const char*>(Wp),
                       bias, res, aux, out0, out1, M, N, K);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16_duo");
    return NEAT_OK;
}


// ---- duo entry points ----
extern "C" int neat_pack_weight_bf16_duo(const float* src, uint16_t* dst, int N, int K, int transposed, void* stream) {
    NEAT_REQUIRE(src && dst, NEAT_ERR_ARG, "neat_pack_weight_bf16_duo: null pointer");
    NEAT_REQUIRE(N > 0 && K > 0 && duo_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_pack_weight_bf16_duo: N=%d K=%d (need N %% 256 == 0, N <= %d, K %% 64 == 0)", N, K, U_MAXN);
    long g = ((long)N * K / 8 + 255) / 256;
    hipLaunchKernelGGL(pack_weight_duo_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       N, K, transposed);
    NEAT_LAUNCH_CHECK("neat_pack_weight_bf16_duo");
    return NEAT_OK;
}

extern "C" int neat_gemm_bf16_duo(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                                  const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                                  void* stream) {
    NEAT_REQUIRE(A && W, NEAT_ERR_ARG, "neat_gemm_bf16_duo: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && duo_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_gemm_bf16_duo: M=%d N=%d K=%d unsupported (need N %% 256 == 0, N <= %d, K %% 64 == 0)", M, N, K, U_MAXN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case NEAT_EPI_BF16:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_duo: out0 is null");
            return launch_gemm_duo<NEAT_EPI_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_BF16:
            NEAT_REQUIRE(out0 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_duo: BIAS_BF16 needs bias, out0");
            return launch_gemm_duo<NEAT_EPI_BIAS_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_RES_F32:
            NEAT_REQUIRE(out0 && bias && res, NEAT_ERR_ARG, "neat_gemm_bf16_duo: BIAS_RES_F32 needs bias, res, out0");
            return launch_gemm_duo<NEAT_EPI_BIAS_RES_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_GELU:
            NEAT_REQUIRE(out1 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_duo: BIAS_GELU needs bias, out1");
            return launch_gemm_duo<NEAT_EPI_BIAS_GELU>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_GELU_BWD:
            NEAT_REQUIRE(out0 && aux, NEAT_ERR_ARG, "neat_gemm_bf16_duo: GELU_BWD needs aux, out0");
            return launch_gemm_duo<NEAT_EPI_GELU_BWD>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_F32:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_duo: out0 is null");
            return launch_gemm_duo<NEAT_EPI_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        default:
            return set_error(NEAT_ERR_ARG, "neat_gemm_bf16_duo: unknown epilogue %d", epilogue);
    }
}
