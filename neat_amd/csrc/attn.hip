// Fused attention forward / backward for short sequences (text L = 77 causal; vision L = 50..257).
// Replaces the SDPA core of nn.MultiheadAttention (reference open_clip/transformer.py:202,227-229,
// causal mask :655-661) and its autograd backward.  head_dim is 64 for every CLIP ViT config.
//
// The whole score matrix of a head lives in MFMA accumulators (L <= 272), so there is no online-softmax
// rescaling and no HBM round trip for S/P.  One workgroup per (sequence, head) over LDS-staged row-major tiles: three
// wavefronts for short sequences (L <= 80: text, ViT-B/32 vision; forward and backward), nine for the longer vision
// sequences (L <= 272, forward only).
//
// Orientation trick (guide §3 "accumulator tile as the next MFMA's operand"): scores are computed
// TRANSPOSED, S^T = K.Q^T, so the accumulator of v_mfma_f32_16x16x32_bf16 holds for one query
// (= lane&15) four consecutive keys per register quad.  Softmax statistics are then lane-local plus two
// xor-shuffles, and the bf16-packed accumulator is directly the B operand of the next product
// (O^T = V^T.P^T, dQ^T = K^T.dS^T) with a permuted k order; the other operand is read with that same
// k order from a [d][key] LDS image.  Products that sum over queries (dV, dK) use the second
// orientation S = Q.K^T, recomputed from the saved log-sum-exp.
#include "common.h"

namespace neat {

constexpr int HD = 64;

__device__ __forceinline__ bf16x8 pack_acc2(const f32x4& a, const f32x4& b) {
    uint4 r = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float xor_max_16_32(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum_16_32(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---- backward: one 192-thread workgroup (3 wavefronts) per (sequence, head) ----
// Q, K, V, dO tiles are staged ONCE, row-major, into LDS with coalesced 16-B loads.  Every MFMA operand
// then comes from LDS: k-contiguous fragments by ds_read_b128, and the fragments of products that sum
// over the sequence index (dQ = dS.K, dK = dS^T.Q, dV = P^T.dO) by the gfx950 transposing read
// ds_read_b64_tr_b16 on the same row-major image -- no transposed copies, no scalar LDS stores.
// The causal work (tile pairs) is dealt to the three waves in snake order, which balances it exactly for
// L = 77 (5 tiles: 5+5+5 tile pairs).
constexpr int AB_WAVES = 3;
constexpr int AB_THREADS = AB_WAVES * 64;
constexpr int AB_PITCH = 72;   // bf16 elements per LDS row (64 + 8 pad: 144 B, 16-B aligned, conflict-free rows)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 lds_rowfrag(const bf16_t* T, int row, int ks, int fq) {
    return *reinterpret_cast<const bf16x8*>(T + row * AB_PITCH + ks * 32 + fq * 8);
}

// fragment with the SUMMED index on the tile rows: element j<4 = T[ta*16+4fq+j][d], j>=4 = T[tb*16+4fq+j-4][d],
// d = dt*16 + fr.  Lane 4q+p of each 16-lane group supplies the address of block row q, columns 4p..4p+3.
__device__ __forceinline__ bf16x8 lds_trfrag(const bf16_t* T, int dt, int ta, int tb, int fr, int fq) {
    const bf16_t* pa = T + (ta * 16 + 4 * fq + (fr >> 2)) * AB_PITCH + dt * 16 + 4 * (fr & 3);
    const bf16_t* pb = T + (tb * 16 + 4 * fq + (fr >> 2)) * AB_PITCH + dt * 16 + 4 * (fr & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// snake deal of NT work items (sorted by decreasing weight) to 3 waves
__device__ __forceinline__ int snake_owner(int rank) {
    const int m = rank % (2 * AB_WAVES);
    return m < AB_WAVES ? m : 2 * AB_WAVES - 1 - m;
}

// Stage NW row-major [LP x 64] bf16 tiles (Q|K|V at column offsets 0, C, 2C of qkv; tile 3 = dO) into LDS with
// pitch AB_PITCH.  U independent 16-B loads per thread are issued before the first LDS store, so that a
// workgroup has its whole tile set in flight at once (rows >= L are zero-filled).
template <int NW, int U, int LP, int NTHR = AB_THREADS>
__device__ __forceinline__ void stage_tiles(bf16_t* dst, const bf16_t* qb, long ld, int part, const bf16_t* dob, int dld, int L) {
    constexpr int total = LP * 8 * NW;
    for (int base = 0; base < total; base += U * NTHR) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NTHR + threadIdx.x;
            const int which = idx / (LP * 8);
            const int rem = idx - which * (LP * 8);
            const int row = rem >> 3, ch = rem & 7;
            v[u] = make_uint4(0, 0, 0, 0);
            if (idx < total && row < L)
                v[u] = which < 3 ? *reinterpret_cast<const uint4*>(qb + (long)row * ld + which * part + ch * 8)
                                 : *reinterpret_cast<const uint4*>(dob + (long)row * dld + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NTHR + threadIdx.x;
            const int which = idx / (LP * 8);
            const int rem = idx - which * (LP * 8);
            const int row = rem >> 3, ch = rem & 7;
            if (idx < total) *reinterpret_cast<uint4*>(dst + which * (LP * AB_PITCH) + row * AB_PITCH + ch * 8) = v[u];
        }
    }
}

// ---- forward: same LDS-staged structure as the backward.  NT <= 5 (text L = 77, ViT-B/32 vision L = 50): 3 waves per
// (sequence, head).  NT = 13 / 17 (ViT-B/16 L = 197, ViT-L/14 L = 257: the gallery encode): 9 waves, two query tiles each, the
// three 39-KB images of one (sequence, head) in LDS -- this replaced a one-wave-per-head kernel that read its K fragments
// from global memory 17 times over and staged V^T with scalar LDS stores (867 us per layer at ViT-L/14, 250 images).  Q, K, V of one (sequence, head) are staged once, row-major, with coalesced 16-B
// loads; V^T fragments come from ds_read_b64_tr_b16; the O tile goes back through LDS (the wave's own Q rows)
// so that the output leaves as whole 128-B rows, 16 B per lane.
template <int NT, int NWV>
__global__ __launch_bounds__(NWV * 64) void attn_fwd3_kernel(const bf16_t* __restrict__ qkv,
                                                               bf16_t* __restrict__ out, float* __restrict__ lse,
                                                               const int32_t* __restrict__ cu, int nseq, int Lmax,
                                                               int heads, int flags) {
    const int causal = flags & 1;
    constexpr int LP = NT * 16;
    constexpr int NP = (NT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ks = Qs + LP * AB_PITCH;
    bf16_t* Vs = Ks + LP * AB_PITCH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int C = heads * HD;
    const long ld = 3L * C;
    const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
    // packed batches: rows cu[s] .. cu[s+1]-1; fixed-length batches: rows s*Lmax .. s*Lmax+Lmax-1
    const long r0 = cu ? (long)cu[s] : (long)s * Lmax;
    const int L = cu ? cu[s + 1] - cu[s] : Lmax;
    const int part = (flags & 2) ? HD : C;                      // head-major qkv: [row][head][q|k|v][64]
    const bf16_t* qb = qkv + r0 * ld + ((flags & 2) ? 3 * h * HD : h * HD);

    constexpr int NTHR = NWV * 64;
    constexpr int SU = NT <= 5 ? (LP * 8 * 3 + NTHR - 1) / NTHR : 6;   // 16-B loads in flight per thread and pass
    stage_tiles<3, SU, LP, NTHR>(Qs, qb, ld, part, nullptr, 0, L);
    __syncthreads();
    const float scale = 0.125f;  // 1/sqrt(64)

    for (int rank = 0; rank < NT; ++rank) {
        if ((NWV == AB_WAVES ? snake_owner(rank) : rank % NWV) != wave) continue;
        const int qt = causal ? NT - 1 - rank : rank;   // causal: heaviest tile first
        const int q0 = qt * 16;
        if (q0 >= L) continue;
        const int query = q0 + fr;
        const int nkt = causal ? qt + 1 : NT;
        const bf16x8 qf0 = lds_rowfrag(Qs, query, 0, fq), qf1 = lds_rowfrag(Qs, query, 1, fq);
        f32x4 st[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                const int krow = kt * 16 + fr;
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Ks, krow, 0, fq), qf0, st[kt], 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Ks, krow, 1, fq), qf1, st[kt], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fq * 4 + r;
                const bool ok = kt < nkt && key < L && (!causal || key <= query);
                st[kt][r] = ok ? st[kt][r] * scale : -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        }
        mx = xor_max_16_32(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = __expf(st[kt][r] - mx);  // exp(-inf) = 0 for masked keys
                sum += st[kt][r];
            }
        sum = xor_sum_16_32(sum);
        if (lse && fq == 0 && query < L) lse[((long)s * heads + h) * Lmax + query] = mx + __logf(sum);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NP; ++ks) {
            if (2 * ks < nkt) {
                const int ta = 2 * ks, tb = (2 * ks + 1 < NT) ? 2 * ks + 1 : 2 * ks;
                const f32x4 hi = (2 * ks + 1 < NT) ? st[tb] : f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16x8 pf = pack_acc2(st[ta], hi);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_trfrag(Vs, dt, ta, tb, fr, fq), pf, o[dt], 0, 0, 0);
            }
        }
        // O tile -> the wave's own (already consumed) Q rows -> global, 16 B per lane, 8 lanes per 128-B row
        const float inv = 1.0f / sum;
        bf16_t* qrow = Qs + query * AB_PITCH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(qrow + dt * 16 + fq * 4) =
                make_uint2(pack2bf(o[dt][0] * inv, o[dt][1] * inv), pack2bf(o[dt][2] * inv, o[dt][3] * inv));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = half * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(Qs + (q0 + r) * AB_PITCH + ch * 8);
            if (q0 + r < L) *reinterpret_cast<uint4*>(out + (r0 + q0 + r) * C + h * HD + ch * 8) = v;
        }
    }
}

template <int NT>
__global__ __launch_bounds__(AB_THREADS) void attn_bwd_kernel(const bf16_t* __restrict__ qkv,
                                                              const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse,
                                                              bf16_t* __restrict__ dqkv,
                                                              const int32_t* __restrict__ cu, int nseq, int Lmax,
                                                              int heads, int flags) {
    const int causal = flags & 1;
    constexpr int LP = NT * 16;
    constexpr int NP = (NT + 1) / 2;  // pairs of 16-wide tiles = MFMA k-steps over a summed sequence index
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ks = Qs + LP * AB_PITCH;
    bf16_t* Vs = Ks + LP * AB_PITCH;
    bf16_t* Os = Vs + LP * AB_PITCH;     // dO
    float* lse_s = reinterpret_cast<float*>(Os + LP * AB_PITCH);
    float* delta = lse_s + LP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per-wave [16 x 64] bf16 tile through which result tiles leave as whole 128-B rows (16 B per lane)
    bf16_t* xs = reinterpret_cast<bf16_t*>(delta + LP) + wave * (16 * AB_PITCH);
    const int fr = lane & 15, fq = lane >> 4;
    const int C = heads * HD;
    const long ld = 3L * C;
    const int s = blockIdx.x / heads, h = blockIdx.x - s * heads;
    const long r0 = cu ? (long)cu[s] : (long)s * Lmax;
    const int L = cu ? cu[s + 1] - cu[s] : Lmax;
    const int part = (flags & 2) ? HD : C;                      // head-major qkv / dqkv: [row][head][q|k|v][64]
    const int hoff = (flags & 2) ? 3 * h * HD : h * HD;
    const bf16_t* qb = qkv + r0 * ld + hoff;
    const bf16_t* dob = dout + r0 * C + h * HD;
    const float* lser = lse + ((long)s * heads + h) * Lmax;
    bf16_t* dqb = dqkv + r0 * ld + hoff;
    bf16_t* dkb = dqb + part;
    bf16_t* dvb = dqb + 2 * part;

    // acc[dt] holds T^T[d = dt*16 + 4fq + r][row = fr] of a 16-row result tile starting at sequence row r0
    auto store_tile = [&](const f32x4 (&a)[4], bf16_t* gbase, int r0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(xs + fr * AB_PITCH + dt * 16 + fq * 4) =
                make_uint2(pack2bf(a[dt][0], a[dt][1]), pack2bf(a[dt][2], a[dt][3]));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = half * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(xs + r * AB_PITCH + ch * 8);
            if (r0 + r < L) *reinterpret_cast<uint4*>(gbase + (long)(r0 + r) * ld + ch * 8) = v;
        }
    };

    // ---- stage Q, K, V, dO (rows >= L zero-filled) and lse ----
    stage_tiles<4, (LP * 8 * 4 + 2 * AB_THREADS - 1) / (2 * AB_THREADS), LP>(Qs, qb, ld, part, dob, C, L);
    for (int q = threadIdx.x; q < LP; q += AB_THREADS) lse_s[q] = q < L ? lser[q] : 0.f;
    __syncthreads();
    const float scale = 0.125f;

    // ---- pass 1: per query tile in the S^T orientation (query on the lane) -> delta, dQ ----
    for (int rank = 0; rank < NT; ++rank) {
        if (snake_owner(rank) != wave) continue;
        const int qt = NT - 1 - rank;              // heaviest (most visible keys) first
        const int q0 = qt * 16;
        if (q0 >= L) continue;                     // (short sequence of a packed batch)
        const int query = q0 + fr;
        const int nkt = causal ? qt + 1 : NT;
        const bf16x8 qf0 = lds_rowfrag(Qs, query, 0, fq), qf1 = lds_rowfrag(Qs, query, 1, fq);
        const bf16x8 df0 = lds_rowfrag(Os, query, 0, fq), df1 = lds_rowfrag(Os, query, 1, fq);
        const float lq = lse_s[query];
        f32x4 pv[NT], ds[NT];
        float dl = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                const int krow = kt * 16 + fr;
                sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Ks, krow, 0, fq), qf0, sv, 0, 0, 0);
                sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Ks, krow, 1, fq), qf1, sv, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Vs, krow, 0, fq), df0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Vs, krow, 1, fq), df1, dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fq * 4 + r;
                const bool ok = kt < nkt && key < L && query < L && (!causal || key <= query);
                const float p = ok ? __expf(sv[r] * scale - lq) : 0.f;
                pv[kt][r] = p;
                ds[kt][r] = dp[r];
                dl += p * dp[r];
            }
        }
        dl = xor_sum_16_32(dl);  // delta[query] = sum_key P * dP  (= rowsum(dO * O))
        if (fq == 0) delta[q0 + fr] = dl;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[kt][r] = pv[kt][r] * (ds[kt][r] - dl) * scale;
        // dQ^T[d, query] = sum_key K^T[d,key] dS^T[key,query]
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NP; ++ks) {
            if (2 * ks < nkt) {
                const int ta = 2 * ks, tb = (2 * ks + 1 < NT) ? 2 * ks + 1 : 2 * ks;
                const f32x4 hi = (2 * ks + 1 < NT) ? ds[tb] : f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16x8 sf = pack_acc2(ds[ta], hi);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_trfrag(Ks, dt, ta, tb, fr, fq), sf, acc[dt], 0, 0, 0);
            }
        }
        store_tile(acc, dqb, q0);
    }
    __syncthreads();  // delta[] complete

    // ---- pass 2: per key tile in the S orientation (key on the lane) -> dV, dK ----
    for (int rank = 0; rank < NT; ++rank) {
        if (snake_owner(rank) != wave) continue;
        const int kt = rank;                       // heaviest (seen by most queries) first
        const int k0 = kt * 16;
        if (k0 >= L) continue;
        const int key = k0 + fr;
        const bf16x8 kf0 = lds_rowfrag(Ks, key, 0, fq), kf1 = lds_rowfrag(Ks, key, 1, fq);
        const bf16x8 vf0 = lds_rowfrag(Vs, key, 0, fq), vf1 = lds_rowfrag(Vs, key, 1, fq);
        f32x4 av[4], ak[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            av[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            ak[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int qp0 = causal ? (kt >> 1) : 0;  // first pair of query tiles that can see this key tile
#pragma unroll
        for (int qp = 0; qp < NP; ++qp) {
            if (qp < qp0) continue;
            f32x4 pt[2], dt2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int qt = 2 * qp + e;
                pt[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                dt2[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool tile_ok = qt < NT && qt * 16 < L && (!causal || qt >= kt);
                if (tile_ok) {
                    const int qrow = qt * 16 + fr;
                    f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                    // D[query][key]: A = Q rows (queries), B = K rows as columns (keys)
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Qs, qrow, 0, fq), kf0, sv, 0, 0, 0);
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Qs, qrow, 1, fq), kf1, sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Os, qrow, 0, fq), vf0, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_rowfrag(Os, qrow, 1, fq), vf1, dp, 0, 0, 0);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + fq * 4);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta + qt * 16 + fq * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int query = qt * 16 + fq * 4 + r;
                        const bool ok = key < L && query < L && (!causal || key <= query);
                        const float p = ok ? __expf(sv[r] * scale - l4[r]) : 0.f;
                        pt[e][r] = p;
                        dt2[e][r] = p * (dp[r] - d4[r]) * scale;
                    }
                }
            }
            const int ta = 2 * qp, tb = (2 * qp + 1 < NT) ? 2 * qp + 1 : 2 * qp;
            const bf16x8 pf = pack_acc2(pt[0], pt[1]);
            const bf16x8 sf = pack_acc2(dt2[0], dt2[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_trfrag(Os, dt, ta, tb, fr, fq), pf, av[dt], 0, 0, 0);  // dV^T[d,key]
                ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_trfrag(Qs, dt, ta, tb, fr, fq), sf, ak[dt], 0, 0, 0);  // dK^T[d,key]
            }
        }
        store_tile(av, dvb, k0);
        store_tile(ak, dkb, k0);
    }
}

static inline int attn_nt(int L) {
    if (L <= 16) return 1;
    if (L <= 32) return 2;
    if (L <= 48) return 3;
    if (L <= 64) return 4;
    if (L <= 80) return 5;
    if (L <= 208) return 13;
    if (L <= 272) return 17;
    return 0;
}

template <int NT>
static int launch_fwd3(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu, int nseq, int L, int heads,
                       int causal, hipStream_t s) {
    constexpr int NWV = NT <= 5 ? AB_WAVES : 9;
    constexpr size_t lds = (size_t)3 * NT * 16 * AB_PITCH * 2;
    if (lds > 64 * 1024) {
        static unsigned long long attr_done = 0;                // per template instance, one bit per device
        if (int rc = ensure_dynamic_lds((const void*)attn_fwd3_kernel<NT, NWV>, (int)lds, &attr_done, "attn_fwd attr")) return rc;
    }
    hipLaunchKernelGGL((attn_fwd3_kernel<NT, NWV>), dim3(nseq * heads), dim3(NWV * 64), lds, s, qkv, out, lse, cu, nseq, L,
                       heads, causal);
    NEAT_LAUNCH_CHECK("neat_attn_fwd");
    return NEAT_OK;
}

template <int NT>
static int launch_bwd(const uint16_t* qkv, const uint16_t* dout, const float* lse, uint16_t* dqkv, const int32_t* cu,
                      int nseq, int L, int heads, int causal, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * NT * 16 * AB_PITCH * 2 + 2 * NT * 16 * 4 + (size_t)AB_WAVES * 16 * AB_PITCH * 2;
    static unsigned long long attr_done = 0;
    if (int rc = ensure_dynamic_lds((const void*)attn_bwd_kernel<NT>, (int)lds, &attr_done, "attn_bwd attr")) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel<NT>, dim3(nseq * heads), dim3(AB_THREADS), lds, s, qkv, dout, lse, dqkv, cu, nseq,
                       L, heads, causal);
    NEAT_LAUNCH_CHECK("neat_attn_bwd");
    return NEAT_OK;
}

}  // namespace neat

using namespace neat;

extern "C" int neat_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens, int nseq, int L,
                             int heads, int causal, void* stream) {
    NEAT_REQUIRE(qkv && out, NEAT_ERR_ARG, "neat_attn_fwd: null pointer");
    NEAT_REQUIRE(nseq > 0 && L > 0 && heads > 0, NEAT_ERR_SHAPE, "neat_attn_fwd: bad shape");
    NEAT_REQUIRE(causal >= 0 && causal <= 3, NEAT_ERR_ARG, "neat_attn_fwd: flags %d (bit 0 causal, bit 1 head-major)", causal);
    NEAT_REQUIRE(!cu_seqlens || L <= 80, NEAT_ERR_SHAPE, "neat_attn_fwd: packed batches need L <= 80 (got %d)", L);
    hipStream_t s = (hipStream_t)stream;
    switch (attn_nt(L)) {
        case 1: return launch_fwd3<1>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 2: return launch_fwd3<2>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 3: return launch_fwd3<3>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 4: return launch_fwd3<4>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 5: return launch_fwd3<5>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 13: return launch_fwd3<13>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        case 17: return launch_fwd3<17>(qkv, out, lse, cu_seqlens, nseq, L, heads, causal, s);
        default: return set_error(NEAT_ERR_SHAPE, "neat_attn_fwd: sequence length %d > 272 unsupported", L);
    }
}

extern "C" int neat_attn_bwd(const uint16_t* qkv, const uint16_t* dout, const float* lse, uint16_t* dqkv,
                             const int32_t* cu_seqlens, int nseq, int L, int heads, int causal, void* stream) {
    NEAT_REQUIRE(qkv && dout && lse && dqkv, NEAT_ERR_ARG, "neat_attn_bwd: null pointer");
    NEAT_REQUIRE(nseq > 0 && L > 0 && heads > 0, NEAT_ERR_SHAPE, "neat_attn_bwd: bad shape");
    NEAT_REQUIRE(causal >= 0 && causal <= 3, NEAT_ERR_ARG, "neat_attn_bwd: flags %d (bit 0 causal, bit 1 head-major)", causal);
    hipStream_t s = (hipStream_t)stream;
    switch (attn_nt(L)) {
        case 1: return launch_bwd<1>(qkv, dout, lse, dqkv, cu_seqlens, nseq, L, heads, causal, s);
        case 2: return launch_bwd<2>(qkv, dout, lse, dqkv, cu_seqlens, nseq, L, heads, causal, s);
        case 3: return launch_bwd<3>(qkv, dout, lse, dqkv, cu_seqlens, nseq, L, heads, causal, s);
        case 4: return launch_bwd<4>(qkv, dout, lse, dqkv, cu_seqlens, nseq, L, heads, causal, s);
        case 5: return launch_bwd<5>(qkv, dout, lse, dqkv, cu_seqlens, nseq, L, heads, causal, s);
        default: return set_error(NEAT_ERR_SHAPE, "neat_attn_bwd: sequence length %d > 80 unsupported", L);
    }
}
