// SHELVED EXPERIMENT (round 1) -- not built, not loaded, not tested by the package.  See README.md in this directory.
// Paste between the tile256 kernel and the "tile256 entry points" of neat_amd/csrc/gemm.hip to build it (it uses
// glds16, t_swz, T_HALF, T_THREADS, T_BARRIER, num_cus and the tile256 packed-weight image of that file).
// =====================================================================================================
// drip: third structure, for the K = C linears (instantiated for K = 512: 8 K-tiles per unit) whose epilogue is as long
// as their main loop.  A CU stores at ~12 B/clk whatever the other CUs do, so an epilogue that runs after the
// main loop costs its full length; here it runs UNDER the next unit's MFMAs:
//
//  * unit = 128 rows x 256 columns, one 512-thread workgroup per CU, waves 2 (M) x 4 (N), wave tile 64 x 64
//    = 4 x 4 accumulator tiles = 64 VGPRs -- and TWO accumulator sets: unit u accumulates into set u & 1 while
//    the finished set of unit u-1 leaves in 8 steps (row tile t, column pair h), one step per K-tile phase;
//  * K-tile = 64 k = three 16-KiB half-tiles (A: 128 rows; B0 / B1 as in tile256, same packed weights), three
//    K-tile buffers (144 KiB): phase g reads K-tile g (16 ds_read_b128), stages K-tile g+2 (6 LDS-DMA per wave),
//    waits until K-tile g+1 has landed, then barrier / 32 MFMAs / barrier, the two wave rows one barrier apart
//    exactly as in tile256;
//  * the residual (fp32) or saved-activation (bf16) operand of a step is loaded TWO phases ahead by inline-asm
//    global loads into 3 rotating register slots; nothing in the loop is visible to the compiler's vmcnt
//    bookkeeping, so the one wait per phase is computed at run time from the exact number of vector-memory
//    operations issued since the previous phase's LDS-DMA (wave-uniform counter -> switch over immediates).
//    A too-small count only over-waits; every path that might issue fewer operations than planned (ragged
//    units, null out0) does not drip at all but writes the whole tile at once and resets the counter to 0;
//  * every phase of a unit is straight-line code (NKT is a template parameter), so accumulator and slot indices
//    are static.
constexpr int D_KT = 3 * T_HALF;              // bytes of a K-tile: A, B0, B1
constexpr int D_BIAS = 3 * D_KT;              // fp32 bias copy behind the three K-tile buffers
constexpr int D_MAXN = 4096;
constexpr int D_LDS = D_BIAS + D_MAXN * 4;    // 160 KiB

__host__ __device__ constexpr bool drip_shape_ok(int N, int K) {
    return N % 256 == 0 && N <= D_MAXN && K == 512;
}

template <int N>
__device__ __forceinline__ void vmcnt_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n (wave-uniform, >= 6) vector-memory operations are outstanding; larger n are rounded DOWN
__device__ __forceinline__ void vmcnt_dyn(int n) {
    switch (n) {
        case 6: vmcnt_imm<6>(); break;
        case 7: vmcnt_imm<7>(); break;
        case 8: vmcnt_imm<8>(); break;
        case 9: vmcnt_imm<9>(); break;
        case 10: vmcnt_imm<10>(); break;
        case 11: vmcnt_imm<11>(); break;
        case 12: vmcnt_imm<12>(); break;
        case 13: vmcnt_imm<13>(); break;
        case 14: vmcnt_imm<14>(); break;
        case 15: vmcnt_imm<15>(); break;
        case 16: vmcnt_imm<16>(); break;
        default:
            if (n > 16) vmcnt_imm<16>();
            else vmcnt_imm<0>();
            break;
    }
}

template <int I>
using ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

template <int EPI, int NKT>
__global__ __launch_bounds__(T_THREADS, 2) void gemm_drip_kernel(
    const bf16_t* __restrict__ A, const char* __restrict__ Wp, const float* __restrict__ bias, const float* res,
    const bf16_t* aux, void* out0, void* out1, int M, int N) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int K = NKT * 64;
    constexpr bool PERM = !(EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_F32);   // bf16 outputs
    constexpr bool HAS_BIAS = EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_BIAS_RES_F32 || EPI == NEAT_EPI_BIAS_GELU;
    constexpr int NL = EPI == NEAT_EPI_BIAS_RES_F32 ? 2 : (EPI == NEAT_EPI_GELU_BWD ? 1 : 0);  // asm loads per step
    constexpr int NS = (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16 || EPI == NEAT_EPI_GELU_BWD) ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // unit -> workgroup deal: all column blocks of a 128-row block on workgroups of one XCD
    const int nrb = (M + 127) >> 7;
    const int ncb = N >> 8;
    const int G_ = gridDim.x;
    const int X = (G_ & 7) == 0 ? 8 : 1;
    const int xcd = blockIdx.x % X, wg = blockIdx.x / X, WX = G_ / X;
    const int nrx = (nrb - xcd + X - 1) / X;
    const int U = (nrx * ncb - wg + WX - 1) / WX;
    if (U <= 0) return;                                           // uniform, before any barrier
    constexpr long wcb_bytes = (long)NKT * 2 * T_HALF;            // packed bytes of one column block (tile256 image)
    auto unit_of = [&](int t, int& row0, int& cb) {
        const int v = wg + t * WX;
        const int i = v / ncb;
        cb = v - i * ncb;
        row0 = (xcd + X * i) << 7;
    };

    if constexpr (HAS_BIAS) {
        for (int i = threadIdx.x * 4; i < N; i += T_THREADS * 4)
            *reinterpret_cast<float4*>(lds + D_BIAS + i * 4) = *reinterpret_cast<const float4*>(bias + i);
    }

    // ---- staging: 6 LDS-DMA per wave and K-tile (instructions 2*wave, 2*wave+1 of each half-tile) ----
    int st_row[2], st_cb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lr = (2 * wave + q) * 8 + (lane >> 3);
        st_row[q] = lr;
        st_cb[q] = ((lane & 7) ^ t_swz(lr)) * 16;
    }
    const uint32_t st_lds = lds_base + wave * 2048;
    int s_t = 0, s_kt = 0, s_row0, s_cb;                          // staging cursor: K-tile g+2
    unit_of(0, s_row0, s_cb);
    int s_buf = 0;                                                // buffer the cursor's K-tile goes to: (g+2) % 3
    auto stage = [&]() {
        const uint32_t dst = st_lds + s_buf * D_KT;
        const char* wb = Wp + s_cb * wcb_bytes + (long)(s_kt * 2) * T_HALF + (2 * wave) * 1024 + lane * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int gr = s_row0 + st_row[q];
            gr = gr < M ? gr : M - 1;
            glds16(reinterpret_cast<const char*>(A) + ((long)gr * K + s_kt * 64) * 2 + st_cb[q], dst + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(wb + q * 1024, dst + T_HALF + q * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(wb + T_HALF + q * 1024, dst + 2 * T_HALF + q * 1024);
        // advance to the next K-tile; past the last one the cursor holds (harmless re-reads into free buffers)
        s_buf = s_buf == 2 ? 0 : s_buf + 1;
        if (s_kt + 1 < NKT) {
            ++s_kt;
        } else if (s_t + 1 < U) {
            s_kt = 0;
            ++s_t;
            unit_of(s_t, s_row0, s_cb);
        }
    };

    // ---- fragment read offsets inside a K-tile buffer ----
    int aoff[2][2], boff[2][2];                                   // [row-tile parity][kstep], [column tile of a pair][kstep]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int lr = wr * 64 + pp * 16 + fr;
            aoff[pp][ks] = lr * 128 + (((ks * 4 + fq) ^ t_swz(lr)) << 4);
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int lr = PERM ? wc * 32 + 8 * (fr >> 2) + 4 * ii + (fr & 3) : wc * 32 + ii * 16 + fr;
            boff[ii][ks] = T_HALF + lr * 128 + (((ks * 4 + fq) ^ t_swz(lr)) << 4);
        }
    }

    stage();                                                       // K-tile 0 -> buffer 0
    stage();                                                       // K-tile 1 -> buffer 1; cursor now at K-tile 2 / buffer 2
    vmcnt_imm<6>();
    __syncthreads();                                               // K-tile 0 landed everywhere; bias copy visible
    if (wr == 1) T_BARRIER();                                      // wave row 1 runs one barrier behind row 0

    f32x4 acc[2][4][4];                                            // [set][column tile][row tile]
    bf16x8 afr[4][2], bfr[4][2];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ld[3][NL > 0 ? NL : 1];                                  // asm-loaded operands of drip steps s % 3
    int since = 0;                                                 // vector-memory ops issued after the last stage()
    int rbuf = 0;                                                  // buffer of the K-tile read next: g % 3

    // geometry of a unit's output as seen by this lane
    auto row_of = [&](int row0, int t) { return row0 + wr * 64 + 16 * t + fr; };
    auto col_of = [&](int cb, int h) { return (cb << 8) + wc * 64 + (PERM ? 32 * h + 8 * fq : 32 * h + 4 * fq); };

    // asm loads of step s of the unit (row0, cb) into slot s % 4; rows of a dripped unit are all < M
    auto issue_loads = [&](auto sc, int row0, int cb) {
        constexpr int s = decltype(sc)::value;
        if constexpr (NL > 0) {
            const long o = (long)row_of(row0, s >> 1) * N + col_of(cb, s & 1);
            if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                const float* p0 = res + o;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[s % 3][0]) : "v"(p0) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(ld[s % 3][1]) : "v"(p0) : "memory");
            } else {
                const bf16_t* p0 = aux + o;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[s % 3][0]) : "v"(p0) : "memory");
            }
            since += NL;
        }
    };

    // step s of a finished accumulator set: IMM = plain loads (whole-tile path), else the asm-loaded slot
    auto emit_step = [&](auto setc, auto sc, auto immc, int row0, int cb) {
        constexpr int SET = decltype(setc)::value;
        constexpr int s = decltype(sc)::value;
        constexpr bool IMM = decltype(immc)::value != 0;
        constexpr int t = s >> 1, h = s & 1;
        const int m = row_of(row0, t);
        const int mc = m < M ? m : M - 1;
        const long o = (long)m * N + col_of(cb, h);
        const long oc = (long)mc * N + col_of(cb, h);
        if constexpr (PERM) {
            const f32x4 lo = acc[SET][2 * h][t], hi = acc[SET][2 * h + 1][t];
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if constexpr (EPI == NEAT_EPI_BF16 || EPI == NEAT_EPI_BIAS_BF16) {
                const uint4 r = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                if (m < M) *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = r;
            } else if constexpr (EPI == NEAT_EPI_BIAS_GELU) {
                uint32_t hp[4], up[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hp[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                    const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                    up[q] = pack2bf(quick_gelu(h0), quick_gelu(h1));
                }
                if (m < M) {
                    if (out0) *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                    *reinterpret_cast<uint4*>((bf16_t*)out1 + o) = make_uint4(up[0], up[1], up[2], up[3]);
                }
            } else {  // NEAT_EPI_GELU_BWD
                uint4 hq;
                if constexpr (IMM) hq = *reinterpret_cast<const uint4*>(aux + oc);
                else hq = __builtin_bit_cast(uint4, ld[s % 3][0]);
                const uint32_t hp[4] = {hq.x, hq.y, hq.z, hq.w};
                uint32_t dp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float h0 = __uint_as_float(hp[q] << 16), h1 = __uint_as_float(hp[q] & 0xffff0000u);
                    dp[q] = pack2bf(v[2 * q] * quick_gelu_grad(h0), v[2 * q + 1] * quick_gelu_grad(h1));
                }
                if (m < M) *reinterpret_cast<uint4*>((bf16_t*)out0 + o) = make_uint4(dp[0], dp[1], dp[2], dp[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {                          // column tiles 2h, 2h+1: 16 fp32 columns apart
                const f32x4 a4 = acc[SET][2 * h + j][t];
                float4 o4 = make_float4(a4[0], a4[1], a4[2], a4[3]);
                if constexpr (EPI == NEAT_EPI_BIAS_RES_F32) {
                    float4 r4;
                    if constexpr (IMM) r4 = *reinterpret_cast<const float4*>(res + oc + 16 * j);
                    else r4 = __builtin_bit_cast(float4, ld[s % 3][j]);
                    o4.x += r4.x, o4.y += r4.y, o4.z += r4.z, o4.w += r4.w;
                }
                if (m < M) *reinterpret_cast<float4*>((float*)out0 + o + 16 * j) = o4;
            }
        }
    };
    auto flush = [&](auto setc, int row0, int cb) {               // whole tile at once, compiler-counted loads
        static_for<0, 8>([&](auto sc) { emit_step(setc, sc, ic<1>{}, row0, cb); });
        since = 0;                                                 // conservative: the next wait covers these stores
    };

    auto init_acc = [&](auto setc, int cb) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS) {
                const int n = (cb << 8) + wc * 64 + (PERM ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq);
                b0 = *reinterpret_cast<const f32x4*>(lds + D_BIAS + n * 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[SET][i][t] = b0;
        }
    };

    // pending = the previous unit (finished set, not yet written); cur = the unit being accumulated
    bool pend = false;
    int p_row0 = 0, p_cb = 0;

    auto unit = [&](auto setc, int row0, int cb, bool cur_drips) {
        constexpr int SET = decltype(setc)::value;
        init_acc(setc, cb);
        static_for<0, NKT>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            const char* base = lds + rbuf * D_KT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[i][ks] = *reinterpret_cast<const bf16x8*>(base + (i >> 1) * T_HALF + boff[i & 1][ks]);
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[t][ks] = *reinterpret_cast<const bf16x8*>(base + (t >> 1) * 4096 + aoff[t & 1][ks]);
            }
            rbuf = rbuf == 2 ? 0 : rbuf + 1;
            stage();
            const int allowed = since + 6;                         // everything issued after the previous stage()
            since = 0;
            vmcnt_dyn(allowed);                                    // K-tile g+1 landed; loads of step p (two phases old) too
            if constexpr (NL > 0 && p < 8) {
                // nothing may read the slot before this point
                u32x4 t0 = ld[p % 3][0];
                if constexpr (NL == 2) {
                    u32x4 t1 = ld[p % 3][1];
                    asm volatile("" : "+v"(t0), "+v"(t1));
                    ld[p % 3][1] = t1;
                } else {
                    asm volatile("" : "+v"(t0));
                }
                ld[p % 3][0] = t0;
            }
            if constexpr (p < 8) {
                if (pend) {
                    emit_step(ic<SET ^ 1>{}, pc, ic<0>{}, p_row0, p_cb);
                    since += NS;
                }
            }
            // operands of the step two phases ahead, into the slot the step above has just released
            if constexpr (NL > 0) {
                if constexpr (p + 2 < 8) {
                    if (pend) issue_loads(ic<p + 2>{}, p_row0, p_cb);
                }
                if constexpr (p >= NKT - 2) {
                    if (cur_drips) issue_loads(ic<p - (NKT - 2)>{}, row0, cb);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            T_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[SET][i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[i][ks], afr[t][ks], acc[SET][i][t], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            T_BARRIER();
        });
        pend = false;
        if (cur_drips) {
            pend = true;
            p_row0 = row0;
            p_cb = cb;
        } else {
            flush(setc, row0, cb);
        }
    };

    const bool can_drip = !(EPI == NEAT_EPI_BIAS_GELU && out0 == nullptr);
    for (int t = 0; t < U; t += 2) {
        int row0, cb;
        unit_of(t, row0, cb);
        unit(ic<0>{}, row0, cb, can_drip && t + 1 < U && row0 + 128 <= M);
        if (t + 1 < U) {
            unit_of(t + 1, row0, cb);
            unit(ic<1>{}, row0, cb, can_drip && t + 2 < U && row0 + 128 <= M);
        }
    }
    if (wr == 0) T_BARRIER();                                      // pairs with wave row 1's last barrier
    vmcnt_imm<0>();                                                // no LDS-DMA may land after the workgroup ends
}

template <int EPI, int NKT>
static int launch_gemm_drip_k(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                              const uint16_t* aux, void* out0, void* out1, int M, int N, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        int rc = check_hip(hipFuncSetAttribute((const void*)gemm_drip_kernel<EPI, NKT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS),
                           "hipFuncSetAttribute(gemm_drip)");
        if (rc) return rc;
        attr_set = true;
    }
    const long units = (long)cdiv(M, 128) * (N / 256);
    const int cus = num_cus();
    const long want = (units + 7) / 8 * 8;
    const int grid = (int)(want < cus ? want : cus);
    hipLaunchKernelGGL((gemm_drip_kernel<EPI, NKT>), dim3(grid), dim3(T_THREADS), D_LDS, s, A,
                       reinterpret_cast<const char*>(Wp), bias, res, aux, out0, out1, M, N);
    NEAT_LAUNCH_CHECK("neat_gemm_bf16_t256(drip)");
    return NEAT_OK;
}

template <int EPI>
static int launch_gemm_drip(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                            const uint16_t* aux, void* out0, void* out1, int M, int N, int K, hipStream_t s) {
    switch (K) {
        case 512: return launch_gemm_drip_k<EPI, 8>(A, Wp, bias, res, aux, out0, out1, M, N, s);
        default: return set_error(NEAT_ERR_SHAPE, "gemm drip: K=%d", K);
    }
}


// ---- entry point ----
extern "C" int neat_gemm_bf16_drip(const uint16_t* A, const uint16_t* W, const float* bias, const float* res,
                                   const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                                   void* stream) {
    NEAT_REQUIRE(A && W, NEAT_ERR_ARG, "neat_gemm_bf16_drip: null operand");
    NEAT_REQUIRE(M > 0 && N > 0 && K > 0 && drip_shape_ok(N, K), NEAT_ERR_SHAPE,
                 "neat_gemm_bf16_drip: M=%d N=%d K=%d unsupported (need N %% 256 == 0, N <= %d, K == 512)", M,
                 N, K, D_MAXN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case NEAT_EPI_BF16:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_drip: out0 is null");
            return launch_gemm_drip<NEAT_EPI_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_BF16:
            NEAT_REQUIRE(out0 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_drip: BIAS_BF16 needs bias, out0");
            return launch_gemm_drip<NEAT_EPI_BIAS_BF16>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_RES_F32:
            NEAT_REQUIRE(out0 && bias && res, NEAT_ERR_ARG, "neat_gemm_bf16_drip: BIAS_RES_F32 needs bias, res, out0");
            return launch_gemm_drip<NEAT_EPI_BIAS_RES_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_BIAS_GELU:
            NEAT_REQUIRE(out1 && bias, NEAT_ERR_ARG, "neat_gemm_bf16_drip: BIAS_GELU needs bias, out1");
            return launch_gemm_drip<NEAT_EPI_BIAS_GELU>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_GELU_BWD:
            NEAT_REQUIRE(out0 && aux, NEAT_ERR_ARG, "neat_gemm_bf16_drip: GELU_BWD needs aux, out0");
            return launch_gemm_drip<NEAT_EPI_GELU_BWD>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        case NEAT_EPI_F32:
            NEAT_REQUIRE(out0, NEAT_ERR_ARG, "neat_gemm_bf16_drip: out0 is null");
            return launch_gemm_drip<NEAT_EPI_F32>(A, W, bias, res, aux, out0, out1, M, N, K, s);
        default:
            return set_error(NEAT_ERR_ARG, "neat_gemm_bf16_drip: unknown epilogue %d", epilogue);
    }
}

