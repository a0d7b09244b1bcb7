// Wavefront-level top-k selection and recall@k scoring (HBM-bound: one pass over the score matrix).
// Replaces torch.topk + one_hot + sum at reference evaluation/retrieval.py:294-300: the reference
// materialises a [b, k, N] int64 one-hot (128 MB per 64-row batch at N = 25 014); here the k winners
// are selected in registers and only gathered against the positives.
//
// Row-contiguous matrices: one 64-lane wavefront per row; every lane keeps a sorted top-k list of its
// strided elements in registers (static indices, insertion by compare-swap), then k rounds of a
// wave-wide arg-max (xor shuffles) merge the 64 lists.  Ties resolve to the lowest column index.
// Transposed views (scores.T, as the reference passes for text retrieval, retrieval.py:248-250) are
// read without a transpose copy: lanes map to logical rows so the loads stay coalesced; the summed
// axis is split across blocks and a second kernel merges the partial lists.
#include "common.h"

namespace neat {

constexpr int IMAX = 0x7fffffff;

template <int KP>
struct TopList {
    float v[KP];
    int i[KP];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KP; ++j) { v[j] = -INFINITY; i[j] = IMAX; }
    }
    // insert keeping (value desc, index asc); callers feed each list in increasing index order, so a
    // strict '>' keeps the earlier (lower) index ahead on ties
    __device__ __forceinline__ void insert(float x, int idx) {
        if (!(x > v[KP - 1])) return;
        bool shifting = false;   // once the slot is found everything below moves down by one
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            if (shifting || x > v[j]) {
                const float tv = v[j]; const int ti = i[j];
                v[j] = x; i[j] = idx;
                x = tv; idx = ti;
                shifting = true;
            }
        }
    }
    // insert() that also says what LEFT: x itself when it does not make the list, else the entry pushed off the end
    // (-inf while the list is not full).  Callers keep the maximum of what left to prove a short list was enough.
    __device__ __forceinline__ float insert_drop(float x, int idx) {
        if (!(x > v[KP - 1])) return x;
        bool shifting = false;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            if (shifting || x > v[j]) {
                const float tv = v[j]; const int ti = i[j];
                v[j] = x; i[j] = idx;
                x = tv; idx = ti;
                shifting = true;
            }
        }
        return x;
    }
    // general insert for merging lists whose index order is arbitrary
    __device__ __forceinline__ void insert_tie(float x, int idx) {
        bool shifting = false;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            if (shifting || x > v[j] || (x == v[j] && idx < i[j])) {
                const float tv = v[j]; const int ti = i[j];
                v[j] = x; i[j] = idx;
                x = tv; idx = ti;
                shifting = true;
            }
        }
    }
    __device__ __forceinline__ void pop() {
#pragma unroll
        for (int j = 0; j + 1 < KP; ++j) { v[j] = v[j + 1]; i[j] = i[j + 1]; }
        v[KP - 1] = -INFINITY; i[KP - 1] = IMAX;
    }
};

// ---- selection in two tiers --------------------------------------------------------------------------------------------
// A full k-entry list per lane makes the pass VALU-bound: a lane sees only Ncol/64 values, so most of them enter its list
// (382 us for 25 014 x 5 000 = 1.3 TB/s).  The fast kernels keep SHORT_M = 4 entries per lane and the maximum of every
// value that left a list ("lost").  The k winners of the row are all inside the merged short lists unless some lane
// held more than 4 of them -- in which case one of them left that lane's list, and lost >= the k-th merged value says so.
// Such rows (about 1 in 10^5 on continuous scores; every row of a matrix full of ties) are flagged in need[] and redone
// by the exact kernels below, which skip every other row.  Results are identical to the exact kernels' on every input.
constexpr int SHORT_M = 4;

// VEC: every lane reads 16 B (4 consecutive columns) per step, two steps in flight -- needs 16-B aligned rows
// (scores % 16 == 0 and ld_row % 4 == 0, checked by the launcher); otherwise 4-B loads.
template <int M, bool VEC>
__global__ __launch_bounds__(256) void topk_rows_fast_kernel(const float* __restrict__ scores, long ld_row, int R,
                                                             int Ncol, int k, int32_t* __restrict__ idx,
                                                             float* __restrict__ val, int* __restrict__ need) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* row = scores + (long)r * ld_row;
        TopList<M> tl;
        tl.init();
        float lost = -INFINITY;
#define TK_INS(x, c) lost = fmaxf(lost, tl.insert_drop((x), (c)))
        if constexpr (VEC) {
            const int nvec = Ncol >> 2;
            const float4* row4 = reinterpret_cast<const float4*>(row);
            int g = lane;
            for (; g + 192 < nvec; g += 256) {                        // four 16-B loads in flight per lane
                const float4 a = row4[g], b = row4[g + 64], c4 = row4[g + 128], d = row4[g + 192];
                const int ca = g * 4, cb = ca + 256, cc = ca + 512, cd = ca + 768;
                TK_INS(a.x, ca); TK_INS(a.y, ca + 1); TK_INS(a.z, ca + 2); TK_INS(a.w, ca + 3);
                TK_INS(b.x, cb); TK_INS(b.y, cb + 1); TK_INS(b.z, cb + 2); TK_INS(b.w, cb + 3);
                TK_INS(c4.x, cc); TK_INS(c4.y, cc + 1); TK_INS(c4.z, cc + 2); TK_INS(c4.w, cc + 3);
                TK_INS(d.x, cd); TK_INS(d.y, cd + 1); TK_INS(d.z, cd + 2); TK_INS(d.w, cd + 3);
            }
            for (; g < nvec; g += 64) {
                const float4 a = row4[g];
                const int ca = g * 4;
                TK_INS(a.x, ca); TK_INS(a.y, ca + 1); TK_INS(a.z, ca + 2); TK_INS(a.w, ca + 3);
            }
            const int c = (nvec << 2) + lane;                         // the last Ncol % 4 columns: largest indices
            if (c < Ncol) TK_INS(row[c], c);
        } else {
            for (int c = lane; c < Ncol; c += 64) TK_INS(row[c], c);
        }
#undef TK_INS
        float bv = -INFINITY;
        int bi = IMAX;
        for (int round = 0; round < k; ++round) {
            bv = tl.v[0];
            bi = tl.i[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (tl.i[0] == bi && bi != IMAX) tl.pop();
            if (lane == 0) {
                idx[(long)r * k + round] = bi;
                if (val) val[(long)r * k + round] = bv;
            }
        }
        lost = wave_max(lost);
        // bv = the k-th selected value.  A value that left some lane's list and is >= bv may belong to the top k
        // (> bv: certainly; == bv: by the lowest-index tie rule) -> exact pass for this row.
        if (lane == 0) need[r] = (M < k && lost >= bv) || bi == IMAX;
    }
}

// Register-resident fast tier for rows of up to 64 * 4 * NV columns (NV <= 20: 5 120 columns, 16-B aligned rows).  The
// insertion lists above are issue-bound: some lane of the wave passes its list threshold at almost every step and then
// the whole wave walks the insertion code, and every merge round is a chain of dependent cross-lane shuffles (136 us for
// 25 014 x 5 000 = 3.7 TB/s, where a plain streaming read reaches 6.3 TB/s on the same GPU).  Here a wave loads its WHOLE
// row into registers first (NV non-temporal 16-B loads per lane, all in flight at once) and selects without a single
// shuffle:
//   1. every lane takes the maximum of its own elements (one v_max per element);
//   2. T = the k-th largest of the 64 lane maxima, found bit by bit on an order-preserving integer key (32 ballots +
//      scalar pop-counts): k lanes hold an element >= T, so the k-th largest element of the row is >= T;
//   3. the elements >= T (typically 10..25) are compacted into an LDS list (ballot + prefix count; a 16-B vector without
//      a candidate costs one compare and one uniform branch);
//   4. with at most 64 candidates, one per lane, every lane counts the candidates that beat its own (value descending,
//      lowest column first on ties) by v_readlane broadcasts; that count is the candidate's place in the output.
// The result is exact.  A row with more than 64 candidates (heavy ties) or fewer than k finite values is flagged for the
// exact tier.
constexpr int REG_CAP = 64;

__device__ __forceinline__ uint32_t order_key(float x) {                  // a < b  <=>  key(a) < key(b) (no NaN)
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int NV>
__global__ __launch_bounds__(256) void topk_rows_reg_kernel(const float* __restrict__ scores, long ld_row, int R, int Ncol,
                                                            int S, int seg_len, int k, int32_t* __restrict__ idx,
                                                            float* __restrict__ val, int* __restrict__ need) {
    // S > 1: rows wider than a wave's registers are cut into S segments of seg_len columns (a multiple of 4); a wave then
    // selects the top k of ONE segment (virtual row r * S + segment; idx / val / need are indexed by virtual row and
    // topk_seg_merge_kernel picks the row's k winners among its S * k candidates).  S == 1: seg_len = Ncol, as before.
    __shared__ float cv[4][REG_CAP];
    __shared__ int ci[4][REG_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wcv = cv[wave];
    int* wci = ci[wave];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int nvirt = R * S;
    for (int vr = blockIdx.x * 4 + wave; vr < nvirt; vr += gridDim.x * 4) {
        const int r = S == 1 ? vr : vr / S;
        const int c_off = (vr - r * S) * seg_len;
        const int ncol = min(seg_len, Ncol - c_off);
        const int nvec = ncol >> 2, rem = ncol & 3;
        const float* row = scores + (long)r * ld_row + c_off;
        const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
        f32x4 x[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int g = lane + 64 * j;
            x[j] = g < nvec ? __builtin_nontemporal_load(row4 + g) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        const float xt = lane < rem ? row[(nvec << 2) + lane] : -INFINITY;   // the last Ncol % 4 columns
        float m = xt;
#pragma unroll
        for (int j = 0; j < NV; ++j) m = fmaxf(m, fmaxf(fmaxf(x[j][0], x[j][1]), fmaxf(x[j][2], x[j][3])));
        // k-th largest lane maximum, MSB first: t grows by a bit whenever at least k lanes stay at or above it
        const uint32_t mk = order_key(m);
        uint32_t t = 0;
#pragma unroll
        for (int b = 31; b >= 0; --b) {
            const uint32_t c = t | (1u << b);
            if (__popcll(__ballot(mk >= c)) >= k) t = c;
        }
        const float T = __uint_as_float((t & 0x80000000u) ? (t & 0x7fffffffu) : ~t);
        int base = 0;                                                     // wave-uniform candidate count
        auto push = [&](float v, int col) {
            const bool p = v >= T;
            const unsigned long long b = __ballot(p);
            if (b) {
                const int pos = base + __popcll(b & lt_mask);
                if (p && pos < REG_CAP) { wcv[pos] = v; wci[pos] = col; }
                base += __popcll(b);
            }
        };
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float m4 = fmaxf(fmaxf(x[j][0], x[j][1]), fmaxf(x[j][2], x[j][3]));
            if (__ballot(m4 >= T)) {
                const int c0 = c_off + ((lane + 64 * j) << 2);
                push(x[j][0], c0); push(x[j][1], c0 + 1); push(x[j][2], c0 + 2); push(x[j][3], c0 + 3);
            }
        }
        if (rem) push(xt, c_off + (nvec << 2) + lane);
        const bool ok = base <= REG_CAP && base >= k && T > -INFINITY;    // wave-uniform
        if (ok) {
            // one candidate per lane; place = number of candidates that come before it
            const bool have = lane < base;
            const float v = have ? wcv[lane] : -INFINITY;
            const int c = have ? wci[lane] : IMAX;
            const uint32_t kv = order_key(v);
            int place = 0;
            for (int i = 0; i < base; ++i) {
                const uint32_t ov = __builtin_amdgcn_readlane(kv, i);
                const int oc = __builtin_amdgcn_readlane(c, i);
                place += (ov > kv || (ov == kv && oc < c)) ? 1 : 0;
            }
            if (have && place < k) {
                idx[(long)vr * k + place] = c;
                if (val) val[(long)vr * k + place] = v;
            }
        }
        if (lane == 0) need[vr] = ok ? 0 : 1;
    }
}

// The k winners of a row among the S * k (<= 64) candidates its segments produced, one wave per row, by the same rank
// counting (value descending, lowest column first).  A row with a flagged segment is flagged as a whole for the exact tier.
__global__ __launch_bounds__(256) void topk_seg_merge_kernel(const int32_t* __restrict__ sgi, const float* __restrict__ sgv,
                                                             const int* __restrict__ sneed, int R, int S, int k,
                                                             int32_t* __restrict__ idx, float* __restrict__ val,
                                                             int* __restrict__ need) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const bool bad = lane < S && sneed[(long)r * S + lane] != 0;
    if (__ballot(bad)) {
        if (lane == 0) need[r] = 1;
        return;
    }
    const int n = S * k;
    const bool have = lane < n;
    const float v = have ? sgv[(long)r * n + lane] : -INFINITY;
    const int c = have ? sgi[(long)r * n + lane] : IMAX;
    const uint32_t kv = order_key(v);
    int place = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t ov = __builtin_amdgcn_readlane(kv, i);
        const int oc = __builtin_amdgcn_readlane(c, i);
        place += (ov > kv || (ov == kv && oc < c)) ? 1 : 0;
    }
    if (have && place < k) {
        idx[(long)r * k + place] = c;
        if (val) val[(long)r * k + place] = v;
    }
    if (lane == 0) need[r] = 0;
}

// The exact kernels (a k-entry list per lane).  need == NULL: every row; else only rows with need[r] != 0.
// One 256-thread block per row whatever the layout (with ld_col != 1, a transposed view, a lane's loads are strided -- slow
// per element, but this tier only sees the rows the fast tier flagged): a k-entry list per lane, then k rounds of a
// block-wide arg-max (wave shuffles, then the four wave winners through LDS).
template <int KP, bool VEC>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, long ld_row, long ld_col, int R,
                                                        int Ncol, int k, int32_t* __restrict__ idx,
                                                        float* __restrict__ val, const int* __restrict__ need) {
    __shared__ float wv[2][4];
    __shared__ int wi[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = threadIdx.x;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        if (need && !need[r]) continue;                               // block-uniform
        const float* row = scores + (long)r * ld_row;
        TopList<KP> tl;
        tl.init();
        if constexpr (VEC) {
            // a thread's columns come in increasing order (4t + j, then + 1024), which insert() relies on
            const int nvec = Ncol >> 2;
            const float4* row4 = reinterpret_cast<const float4*>(row);
            int g = t;
            for (; g + 256 < nvec; g += 512) {
                const float4 a = row4[g], b = row4[g + 256];
                const int ca = g * 4, cb = (g + 256) * 4;
                tl.insert(a.x, ca); tl.insert(a.y, ca + 1); tl.insert(a.z, ca + 2); tl.insert(a.w, ca + 3);
                tl.insert(b.x, cb); tl.insert(b.y, cb + 1); tl.insert(b.z, cb + 2); tl.insert(b.w, cb + 3);
            }
            if (g < nvec) {
                const float4 a = row4[g];
                const int ca = g * 4;
                tl.insert(a.x, ca); tl.insert(a.y, ca + 1); tl.insert(a.z, ca + 2); tl.insert(a.w, ca + 3);
            }
            const int c = (nvec << 2) + t;                            // the last Ncol % 4 columns: largest indices
            if (c < Ncol) tl.insert(row[c], c);
        } else {
            int c = t;
            for (; c + 7 * 256 < Ncol; c += 8 * 256) {                // eight independent (possibly strided) loads in flight
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = row[(long)(c + j * 256) * ld_col];
#pragma unroll
                for (int j = 0; j < 8; ++j) tl.insert(x[j], c + j * 256);
            }
            for (; c < Ncol; c += 256) tl.insert(row[(long)c * ld_col], c);
        }
        for (int round = 0; round < k; ++round) {
            float bv = tl.v[0];
            int bi = tl.i[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { wv[round & 1][wave] = bv; wi[round & 1][wave] = bi; }
            __syncthreads();                                          // one barrier per round: the slots alternate
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float ov = wv[round & 1][w];
                const int oi = wi[round & 1][w];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (tl.i[0] == bi && bi != IMAX) tl.pop();
            if (t == 0) {
                idx[(long)r * k + round] = bi;
                if (val) val[(long)r * k + round] = bv;
            }
        }
        __syncthreads();                                              // before the next row reuses the slots
    }
}

// fast tier, transposed view: a lane owns RPL consecutive logical rows (RPL = 4: one 16-B load covers them; a wave reads
// 1 KiB of every column), blockIdx.y owns a column range; per row a SHORT_M list + the maximum of what left it.
// partials, row-major so that ONE wavefront can finish a row: pv / pi [R][nsplit][SHORT_M], plost [R][nsplit].
template <int M, int RPL>
__global__ __launch_bounds__(64) void topk_cols_fast_kernel(const float* __restrict__ scores, long ld_col, int R, int Ncol,
                                                            int nsplit, float* __restrict__ pv, int* __restrict__ pi,
                                                            float* __restrict__ plost) {
    const int r0 = (blockIdx.x * 64 + threadIdx.x) * RPL;
    if (r0 >= R) return;
    const int per = (Ncol + nsplit - 1) / nsplit;
    const int c0 = blockIdx.y * per;
    const int c1 = min(Ncol, c0 + per);
    TopList<M> tl[RPL];
    float lost[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) { tl[q].init(); lost[q] = -INFINITY; }
    auto take = [&](const float* x, int c) {
#pragma unroll
        for (int q = 0; q < RPL; ++q) lost[q] = fmaxf(lost[q], tl[q].insert_drop(x[q], c));
    };
    auto load = [&](int c, float* x) {
        const float* p = scores + (long)c * ld_col + r0;
        if constexpr (RPL == 4) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
            x[0] = *p;
        }
    };
    int c = c0;
    for (; c + 4 <= c1; c += 4) {                        // four independent loads in flight per lane
        float x[4][RPL];
#pragma unroll
        for (int j = 0; j < 4; ++j) load(c + j, x[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) take(x[j], c + j);
    }
    for (; c < c1; ++c) {
        float x[RPL];
        load(c, x);
        take(x, c);
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = r0 + q;
        if (r < R) {
            const long o = ((long)r * nsplit + blockIdx.y) * M;
#pragma unroll
            for (int j = 0; j < M; ++j) { pv[o + j] = tl[q].v[j]; pi[o + j] = tl[q].i[j]; }
            plost[(long)r * nsplit + blockIdx.y] = lost[q];
        }
    }
}

// finish of the fast tier: one wavefront per row over that row's nsplit * M candidates (value, column), exactly the row
// kernel's scheme -- a short list per lane plus what left it, k rounds of a wave-wide arg-max, and the sufficiency proof.
template <int M>
__global__ __launch_bounds__(256) void topk_pairs_finish_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                                const float* __restrict__ plost, int R, int nsplit, int k,
                                                                int32_t* __restrict__ idx, float* __restrict__ val,
                                                                int* __restrict__ need) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    const int ncand = nsplit * M;
    const float* cv = pv + (long)r * ncand;
    const int* ci = pi + (long)r * ncand;
    TopList<M> tl;
    tl.init();
    float lost = -INFINITY;
    for (int s = lane; s < nsplit; s += 64) lost = fmaxf(lost, plost[(long)r * nsplit + s]);
    for (int e = lane; e < ncand; e += 64) {
        // walk the candidates rank-major (all first entries of the splits, then all second entries ...): a row's winners
        // are mostly first entries, and this spreads those over all 64 lanes instead of every fourth one
        const int c = (e % nsplit) * M + e / nsplit;
        float x = cv[c];
        int xi = ci[c];
        if (xi == IMAX) continue;
        // insert keeping (value desc, column asc); what falls off the end is remembered
        if (!(x > tl.v[M - 1] || (x == tl.v[M - 1] && xi < tl.i[M - 1]))) { lost = fmaxf(lost, x); continue; }
        bool shifting = false;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (shifting || x > tl.v[j] || (x == tl.v[j] && xi < tl.i[j])) {
                const float tv = tl.v[j]; const int ti = tl.i[j];
                tl.v[j] = x; tl.i[j] = xi;
                x = tv; xi = ti;
                shifting = true;
            }
        }
        if (xi != IMAX) lost = fmaxf(lost, x);
    }
    float bv = -INFINITY;
    int bi = IMAX;
    for (int round = 0; round < k; ++round) {
        bv = tl.v[0];
        bi = tl.i[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (tl.i[0] == bi && bi != IMAX) tl.pop();
        if (lane == 0) {
            idx[(long)r * k + round] = bi;
            if (val) val[(long)r * k + round] = bv;
        }
    }
    lost = wave_max(lost);
    if (lane == 0) need[r] = (M < k && lost >= bv) || bi == IMAX;
}

// ---- transposed views by a tournament over chunks ------------------------------------------------------------------------
// Logical row r of a transposed view is a strided column of the matrix in memory, so lanes must map to logical rows to keep
// the loads coalesced, and a wave-instruction then serves 256 selections at once: per-lane insertion lists diverge at almost
// every step (145 us + a 28-us merge for the 5 000 x 25 014 view = 2.8 TB/s).  Instead:
//   pass 1 (topk_cols_chunk_kernel, one read of the matrix, branch-free, 5 VALU per element): the summed index is cut into
//     chunks of CH elements (>= 128: measured 64 / 128 / 256 -> 123 / 112 / 115 us on the 5 000 x 25 014 view); per (chunk,
//     row) the maximum, where it is (first occurrence) and the SECOND largest value;
//   pass 2 (topk_cols_select_kernel, reads 12 B per (chunk, row)): T = k-th largest of the row's chunk maxima (by lanes,
//     ballot search as in the row kernel): at least k elements are >= T, so every winner is either a chunk maximum >= T or
//     lives in a chunk whose second value is >= T too; only those chunks (a few per row) are read again, the candidates go
//     through LDS and are placed by rank counting.  Exact; rows with too many candidates go to the exact tier.
constexpr int COLS_MAXCHUNKS = 512;     // chunk planes of a select workgroup's rows in LDS: 512 x 9 x 12 B = 54 KiB
constexpr int COLS_RESCAN_CAP = 32;
constexpr int COLS_RB = 8;              // logical rows per select workgroup (one per wave): 32 contiguous bytes per chunk plane
constexpr int COLS_P = COLS_RB + 1;     // LDS pitch of a chunk's row group

// lanes per workgroup of the passes whose lanes run along the logical rows of a transposed view (4 rows per lane)
static inline int cols_group_width(int R) {
    const int lanes = cdiv(R, 4);
    return lanes >= 1024 ? 256 : lanes >= 256 ? 128 : 64;
}

static inline int cols_chunk_len(int Ncol) {
    int ch = cdiv(Ncol, COLS_MAXCHUNKS);
    ch = (ch + 7) / 8 * 8;
    return ch < 128 ? 128 : ch;
}

// Workgroups of up to 256 lanes: the four waves of one walk the same matrix rows in step and read 4 KiB contiguous per row
// between them (64-lane workgroups, 1 KiB each, landed on different XCDs: 112 us against 104 us for the 5 000 x 25 014 view;
// 128 lanes 107 us, 512 lanes 112 us -- 1 250 lanes fill 512-wide groups badly).
__global__ __launch_bounds__(256) void topk_cols_chunk_kernel(const float* __restrict__ scores, long ld_col, int R, int Ncol,
                                                             int CH, float* __restrict__ pmx, float* __restrict__ psc,
                                                             int* __restrict__ pag) {
    const int r0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r0 >= R) return;
    const int c0 = blockIdx.y * CH;
    const int c1 = min(Ncol, c0 + CH);
    float mx[4], sc[4];
    int ag[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mx[q] = -INFINITY; sc[q] = -INFINITY; ag[q] = c0; }
    auto take = [&](const f32x4& x, int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = fmaxf(sc[q], fminf(x[q], mx[q]));
            ag[q] = x[q] > mx[q] ? c : ag[q];
            mx[q] = fmaxf(mx[q], x[q]);
        }
    };
    const float* p = scores + r0;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {                      // sixteen 16-B loads in flight per lane
        f32x4 x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (long)(c + j) * ld_col));
#pragma unroll
        for (int j = 0; j < 16; ++j) take(x[j], c + j);
    }
    for (; c < c1; ++c) take(*reinterpret_cast<const f32x4*>(p + (long)c * ld_col), c);
    const long o = (long)blockIdx.y * R + r0;
    *reinterpret_cast<float4*>(pmx + o) = make_float4(mx[0], mx[1], mx[2], mx[3]);
    *reinterpret_cast<float4*>(psc + o) = make_float4(sc[0], sc[1], sc[2], sc[3]);
    *reinterpret_cast<int4*>(pag + o) = make_int4(ag[0], ag[1], ag[2], ag[3]);
}

__global__ __launch_bounds__(COLS_RB * 64) void topk_cols_select_kernel(const float* __restrict__ scores, long ld_col, int R, int Ncol,
                                                               int CH, int nchunk, const float* __restrict__ pmx,
                                                               const float* __restrict__ psc, const int* __restrict__ pag,
                                                               int k, int32_t* __restrict__ idx, float* __restrict__ val,
                                                               int* __restrict__ need) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lmx = reinterpret_cast<float*>(smem);                   // [nchunk][COLS_RB + 1] (odd pitch: conflict-free down a column)
    float* lsc = lmx + nchunk * COLS_P;
    int* lag = reinterpret_cast<int*>(lsc + nchunk * COLS_P);
    float* cv = reinterpret_cast<float*>(lag + nchunk * COLS_P);       // per wave: REG_CAP candidates + COLS_RESCAN_CAP chunk ids
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wcv = cv + wave * (2 * REG_CAP + COLS_RESCAN_CAP);
    int* wci = reinterpret_cast<int*>(wcv + REG_CAP);
    int* wrs = wci + REG_CAP;
    const int r0 = blockIdx.x * COLS_RB;
    for (int e = threadIdx.x; e < nchunk * COLS_RB; e += COLS_RB * 64) {
        const int c = e / COLS_RB, i = e % COLS_RB;
        const bool in = r0 + i < R;
        const long o = (long)c * R + r0 + i;
        lmx[c * COLS_P + i] = in ? pmx[o] : -INFINITY;
        lsc[c * COLS_P + i] = in ? psc[o] : -INFINITY;
        lag[c * COLS_P + i] = in ? pag[o] : 0;
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    {                                                              // one logical row per wave: 24 waves per CU hide the
        const int i = wave;                                        // latency of the (rare) second reads of a chunk
        const int r = r0 + i;
        if (r >= R) return;                                        // wave-uniform, after the only barrier
        float m = -INFINITY;
        for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, lmx[c * COLS_P + i]);
        const uint32_t mk = order_key(m);
        uint32_t t = 0;
#pragma unroll
        for (int b = 31; b >= 0; --b) {
            const uint32_t cnd = t | (1u << b);
            if (__popcll(__ballot(mk >= cnd)) >= k) t = cnd;
        }
        const float T = __uint_as_float((t & 0x80000000u) ? (t & 0x7fffffffu) : ~t);
        int base = 0, nres = 0;                                    // wave-uniform counts
        for (int cb = 0; cb < nchunk; cb += 64) {
            const int c = cb + lane;
            const bool in = c < nchunk;
            const float v = in ? lmx[c * COLS_P + i] : -INFINITY;
            const bool p = in && v >= T;
            const unsigned long long b = __ballot(p);
            if (b) {
                const int pos = base + __popcll(b & lt_mask);
                if (p && pos < REG_CAP) { wcv[pos] = v; wci[pos] = lag[c * COLS_P + i]; }
                base += __popcll(b);
            }
            const bool p2 = in && lsc[c * COLS_P + i] >= T;
            const unsigned long long b2 = __ballot(p2);
            if (b2) {
                const int pos = nres + __popcll(b2 & lt_mask);
                if (p2 && pos < COLS_RESCAN_CAP) wrs[pos] = c;
                nres += __popcll(b2);
            }
        }
        bool ok = T > -INFINITY && nres <= COLS_RESCAN_CAP;
        if (ok) {
            for (int j = 0; j < nres; ++j) {                       // chunks with a second element at or above T: read again
                const int c = wrs[j];
                const int skip = lag[c * COLS_P + i];                  // the chunk maximum is already a candidate
                const int c0 = c * CH, c1 = min(Ncol, c0 + CH);
                for (int cc = c0 + lane; cc - lane < c1; cc += 64) {
                    const bool in = cc < c1;
                    const float v = in ? scores[(long)cc * ld_col + r] : -INFINITY;
                    const bool p = in && v >= T && cc != skip;
                    const unsigned long long b = __ballot(p);
                    if (b) {
                        const int pos = base + __popcll(b & lt_mask);
                        if (p && pos < REG_CAP) { wcv[pos] = v; wci[pos] = cc; }
                        base += __popcll(b);
                    }
                }
            }
            ok = base <= REG_CAP && base >= k;
        }
        if (ok) {
            const bool have = lane < base;
            const float v = have ? wcv[lane] : -INFINITY;
            const int c = have ? wci[lane] : IMAX;
            const uint32_t kv = order_key(v);
            int place = 0;
            for (int j = 0; j < base; ++j) {
                const uint32_t ov = __builtin_amdgcn_readlane(kv, j);
                const int oc = __builtin_amdgcn_readlane(c, j);
                place += (ov > kv || (ov == kv && oc < c)) ? 1 : 0;
            }
            if (have && place < k) {
                idx[(long)r * k + place] = c;
                if (val) val[(long)r * k + place] = v;
            }
        }
        if (lane == 0) need[r] = ok ? 0 : 1;
    }
}

// recall[r] = (# of top-k indices that are positives) / (# positives in row r)
__global__ __launch_bounds__(256) void recall_rows_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ pos,
                                                          long ld_row, int R, int Ncol, int k, float* __restrict__ recall) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const uint8_t* row = pos + (long)r * ld_row;
        float n = 0.f;
        int c = 0;
        if ((((uintptr_t)row) & 15) == 0) {               // 16 flags per load
            const int nv = Ncol >> 4;
            for (int g = lane; g < nv; g += 64) {
                const uint4 w = reinterpret_cast<const uint4*>(row)[g];
                const uint32_t ws_[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t x = ws_[j];
                    x = (x | (x >> 1) | (x >> 2) | (x >> 3) | (x >> 4) | (x >> 5) | (x >> 6) | (x >> 7)) & 0x01010101u;   // byte != 0
                    n += (float)__popc(x);
                }
            }
            c = nv << 4;
        }
        for (c += lane; c < Ncol; c += 64) n += row[c] ? 1.f : 0.f;
        n = wave_sum(n);
        float tp = 0.f;
        for (int j = lane; j < k; j += 64) {
            const int c = idx[(long)r * k + j];
            if (c >= 0 && c < Ncol) tp += row[c] ? 1.f : 0.f;
        }
        tp = wave_sum(tp);
        if (lane == 0) recall[r] = tp / n;  // 0/0 -> NaN like torch
    }
}

__global__ __launch_bounds__(64) void recall_cols_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ pos,
                                                         long ld_col, int R, int Ncol, int k, float* __restrict__ recall) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    float n = 0.f;
    for (int c = 0; c < Ncol; ++c) n += pos[(long)c * ld_col + r] ? 1.f : 0.f;
    float tp = 0.f;
    for (int j = 0; j < k; ++j) {
        const int c = idx[(long)r * k + j];
        if (c >= 0 && c < Ncol) tp += pos[(long)c * ld_col + r] ? 1.f : 0.f;
    }
    recall[r] = tp / n;
}

// The same in two launches for tall views (the 5 000 x 25 014 one: 79 workgroups walking 25 014 flags each took 6 ms): the
// positives of a logical row are counted by (row block, column chunk) workgroups into the recall buffer itself, as
// integers (atomic adds: exact and order-independent), and the finish kernel turns the count into tp / n.
__global__ __launch_bounds__(64) void recall_cols_count_kernel(const uint8_t* __restrict__ pos, long ld_col, int R, int Ncol,
                                                               int CH, int* __restrict__ count) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const int c0 = blockIdx.y * CH, c1 = min(Ncol, c0 + CH);
    int n = 0;
    for (int c = c0; c < c1; ++c) n += pos[(long)c * ld_col + r] ? 1 : 0;
    if (n) atomicAdd(count + r, n);
}
__global__ __launch_bounds__(64) void recall_cols_finish_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ pos,
                                                                long ld_col, int R, int Ncol, int k, float* __restrict__ recall) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const float n = (float)reinterpret_cast<const int*>(recall)[r];
    float tp = 0.f;
    for (int j = 0; j < k; ++j) {
        const int c = idx[(long)r * k + j];
        if (c >= 0 && c < Ncol) tp += pos[(long)c * ld_col + r] ? 1.f : 0.f;
    }
    recall[r] = tp / n;
}


// Recall from ONE top-K list per row for several cut-offs at once, with the positives given as owner vectors instead
// of a dense [R, Ncol] bool matrix (NegBench: caption t belongs to image txt2img[t]; reference evaluation/retrieval.py:
// 229-240 builds the bool matrix from exactly this list).  Column c is a positive of row r iff
//   row_owner != NULL and row_owner[r] == c      (text -> image direction: one positive per caption), or
//   col_owner != NULL and col_owner[c] == r      (image -> text direction: the captions owned by image r).
// npos[r] = number of positives of row r (NULL: 1).  recall[q*R + r] = |top-ks[q](r) ∩ positives(r)| / npos[r].
__global__ __launch_bounds__(256) void recall_owner_kernel(const int32_t* __restrict__ idx, int R, int K,
                                                           const int32_t* __restrict__ row_owner,
                                                           const int32_t* __restrict__ col_owner,
                                                           const int32_t* __restrict__ npos, const int32_t* __restrict__ ks,
                                                           int nk, float* __restrict__ recall) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int own = row_owner ? row_owner[r] : -1;
    const float n = npos ? (float)npos[r] : 1.0f;
    int q = 0, tp = 0;
    for (int j = 0; j < K && q < nk; ++j) {
        const int c = idx[(long)r * K + j];
        // c == 0x7fffffff: the top-k kernels' filler for a row with fewer than K selectable entries (-inf / NaN scores):
        // no column, so never a positive -- and never an index into col_owner
        if (c >= 0 && c != 0x7fffffff) tp += ((row_owner && c == own) || (col_owner && col_owner[c] == r)) ? 1 : 0;
        while (q < nk && ks[q] == j + 1) recall[(long)q++ * R + r] = (float)tp / n;   // 0/0 -> NaN like torch
    }
    for (; q < nk; ++q) recall[(long)q * R + r] = (float)tp / n;                     // cut-offs beyond K: all of the list
}

// ---- median-rank support, evaluation/retrieval.py:186-214: rank (1-based, descending scores) of the best-ranked
// positive of every row; equal scores are ordered by index, like the top-k kernels.  0 = row has no positive.
__global__ __launch_bounds__(256) void best_positive_rank_kernel(const float* __restrict__ scores, long ld_row,
                                                                 long ld_col, const uint8_t* __restrict__ pos,
                                                                 long pld_row, long pld_col, int R, int Ncol,
                                                                 int32_t* __restrict__ rank) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int sc[4];
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* srow = scores + (long)r * ld_row;
    const uint8_t* prow = pos + (long)r * pld_row;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < Ncol; c += 256) {
        if (prow[(long)c * pld_col]) {
            const float v = srow[(long)c * ld_col];
            if (bi == 0x7fffffff || v > best) { best = v; bi = c; }   // c ascending per thread: first max kept
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (si[w] != 0x7fffffff && (bi == 0x7fffffff || sv[w] > best || (sv[w] == best && si[w] < bi))) { best = sv[w]; bi = si[w]; }
    if (bi == 0x7fffffff) {
        if (threadIdx.x == 0) rank[r] = 0;
        return;
    }
    int cnt = 0;
    for (int c = threadIdx.x; c < Ncol; c += 256) {
        const float v = srow[(long)c * ld_col];
        cnt += (v > best || (v == best && c < bi)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) sc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[r] = 1 + sc[0] + sc[1] + sc[2] + sc[3];
}

// The same rank with the positives given as owner vectors (see recall_owner_kernel): no dense bool matrix is read.
// row_owner: the single positive of row r is column row_owner[r]; col_owner: the positives of row r are the columns c
// with col_owner[c] == r (a 4-B-per-column scan that stays in L2).
__global__ __launch_bounds__(256) void best_positive_rank_owner_kernel(const float* __restrict__ scores, long ld_row,
                                                                       long ld_col, const int32_t* __restrict__ row_owner,
                                                                       const int32_t* __restrict__ col_owner, int R,
                                                                       int Ncol, int32_t* __restrict__ rank) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int sc[4];
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* srow = scores + (long)r * ld_row;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (row_owner) {
        const int c = row_owner[r];
        if (c >= 0 && c < Ncol) { best = srow[(long)c * ld_col]; bi = c; }
    } else {
        for (int c = threadIdx.x; c < Ncol; c += 256) {
            if (col_owner[c] == r) {
                const float v = srow[(long)c * ld_col];
                if (bi == 0x7fffffff || v > best) { best = v; bi = c; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        if (lane == 0) { sv[wave] = best; si[wave] = bi; }
        __syncthreads();
        best = sv[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (si[w] != 0x7fffffff && (bi == 0x7fffffff || sv[w] > best || (sv[w] == best && si[w] < bi))) { best = sv[w]; bi = si[w]; }
    }
    if (bi == 0x7fffffff) {
        if (threadIdx.x == 0) rank[r] = 0;
        return;
    }
    int cnt = 0;
    for (int c = threadIdx.x; c < Ncol; c += 256) {
        const float v = srow[(long)c * ld_col];
        cnt += (v > best || (v == best && c < bi)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) sc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[r] = 1 + sc[0] + sc[1] + sc[2] + sc[3];
}

// The owner form on a TRANSPOSED view (ld_row == 1: logical row r is matrix column r, image -> text): a block per logical row
// would read a strided column (1 390 us for the 5 000 x 25 014 view).  Instead lanes map to logical rows, as in the chunk
// tournament: (1) the best positive of every logical row by a 64-bit atomic max on (order key of the score, ~column) -- max
// is order-independent, so the result is deterministic; (2) one coalesced pass over the matrix that counts, per logical row,
// the elements ranked before that positive (value descending, lowest column first) with integer atomics; (3) rank = count
// + 1, or 0 for a row without a positive.  Scores are compared as floats (-0.0 == 0.0); the key is only used for the max.
__device__ __forceinline__ float order_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void bpr_cols_best_kernel(const float* __restrict__ scores, long ld_col,
                                                            const int32_t* __restrict__ row_owner,
                                                            const int32_t* __restrict__ col_owner, int R, int Ncol,
                                                            unsigned long long* __restrict__ best) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (row_owner) {                                              // one positive per logical row: column row_owner[r]
        if (t >= R) return;
        const int c = row_owner[t];
        if (c < 0 || c >= Ncol) return;
        const float v = scores[(long)c * ld_col + t] + 0.0f;      // -0.0 -> +0.0
        best[t] = ((unsigned long long)order_key(v) << 32) | (0xffffffffu - (uint32_t)c);
    } else {
        if (t >= Ncol) return;
        const int r = col_owner[t];
        if (r < 0 || r >= R) return;
        const float v = scores[(long)t * ld_col + r] + 0.0f;
        atomicMax(best + r, ((unsigned long long)order_key(v) << 32) | (0xffffffffu - (uint32_t)t));
    }
}

__global__ __launch_bounds__(256) void bpr_cols_count_kernel(const float* __restrict__ scores, long ld_col, int R, int Ncol,
                                                            int CH, const unsigned long long* __restrict__ best,
                                                            int32_t* __restrict__ rank) {
    const int r0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r0 >= R) return;
    const int c0 = blockIdx.y * CH, c1 = min(Ncol, c0 + CH);
    float bv[4];
    int bi[4], cnt[4];
    bool has[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long k = best[r0 + q];
        has[q] = k != 0ull;
        bv[q] = has[q] ? order_unkey((uint32_t)(k >> 32)) : INFINITY;   // nothing beats +inf at a lower column than -1
        bi[q] = has[q] ? (int)(0xffffffffu - (uint32_t)k) : -1;
        cnt[q] = 0;
    }
    auto take = [&](const f32x4& x, int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt[q] += (x[q] > bv[q] || (x[q] == bv[q] && c < bi[q])) ? 1 : 0;
    };
    const float* p = scores + r0;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {
        f32x4 x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (long)(c + j) * ld_col));
#pragma unroll
        for (int j = 0; j < 16; ++j) take(x[j], c + j);
    }
    for (; c < c1; ++c) take(*reinterpret_cast<const f32x4*>(p + (long)c * ld_col), c);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (has[q] && cnt[q]) atomicAdd(rank + r0 + q, cnt[q]);
}

__global__ __launch_bounds__(256) void bpr_cols_finish_kernel(const unsigned long long* __restrict__ best, int R,
                                                              int32_t* __restrict__ rank) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) rank[r] = best[r] ? rank[r] + 1 : 0;
}

// ---- multiple-choice scoring, evaluation/mcq.py:92-97 and :199-205: logits[b,n] = <img[b], txt[n,b]>,
// pred[b] = argmax_n (first max).  One wavefront per image.
__global__ __launch_bounds__(256) void mcq_scores_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                         float* __restrict__ logits, int32_t* __restrict__ pred, int B,
                                                         int nopt, int D) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* ib = img + (long)b * D;
    float bestv = -INFINITY;
    int besti = 0;
    for (int n = 0; n < nopt; ++n) {
        const float* tb = txt + ((long)n * B + b) * D;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc = fmaf(ib[d], tb[d], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && logits) logits[(long)b * nopt + n] = acc;
        if (acc > bestv) { bestv = acc; besti = n; }
    }
    if (lane == 0) pred[b] = besti;
}

static inline int topk_nsplit(int Ncol) {
    int s = Ncol / 512;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}
static inline int topk_kp(int k) { return k <= 1 ? 1 : (k <= 8 ? 8 : (k <= 16 ? 16 : 32)); }

}  // namespace neat

using namespace neat;

// column splits of the fast transposed-view kernel: enough 64-lane blocks (a block covers 256 logical rows) for ~16
// waves per CU, at least 32 columns each
static inline int topk_fast_nsplit(int R, int Ncol) {
    const int rowblocks = cdiv(R, 256);
    int s = cdiv(4096, rowblocks);
    const int cap = Ncol / 32 > 1 ? Ncol / 32 : 1;
    s = s > cap ? cap : s;
    return s < 1 ? 1 : s;
}

struct TopkWs {                // carve-up of the caller's workspace (all 256-B aligned)
    int* need;                 // [R]
    float* fv; int* fi;        // fast tier: [R][nsf][SHORT_M]
    float* flost;              // [R][nsf]
    float* pmx; float* psc; int* pag;   // transposed views: [nchunk][R] maximum, second value, position of the maximum
    int* sgi; float* sgv; int* sneed;   // segmented wide rows: [R][S][k] candidates, [R][S] flags
    long bytes;
};
// Rows wider than the register kernel holds (5 120 columns) are cut into S segments: seg_len a multiple of 4, at most 5 120.
constexpr int REG_MAXCOLS = 20 * 256;
static inline int topk_segments(int Ncol, int* seg_len) {
    const int S = cdiv(Ncol, REG_MAXCOLS);
    *seg_len = S <= 1 ? Ncol : (cdiv(Ncol, S) + 3) / 4 * 4;
    return S;
}
static TopkWs topk_carve(void* ws, int R, int Ncol, int k) {
    auto up = [](long b) { return (b + 255) / 256 * 256; };
    const long nsf = topk_fast_nsplit(R, Ncol);
    char* p = reinterpret_cast<char*>(ws);
    long off = 0;
    TopkWs w;
    w.need = reinterpret_cast<int*>(p + off); off += up((long)R * 4);
    w.fv = reinterpret_cast<float*>(p + off); off += up(nsf * SHORT_M * (long)R * 4);
    w.fi = reinterpret_cast<int*>(p + off); off += up(nsf * SHORT_M * (long)R * 4);
    w.flost = reinterpret_cast<float*>(p + off); off += up(nsf * (long)R * 4);
    const long nch = cdiv(Ncol, cols_chunk_len(Ncol));             // tournament planes of the transposed-view path
    w.pmx = reinterpret_cast<float*>(p + off); off += up(nch * (long)R * 4);
    w.psc = reinterpret_cast<float*>(p + off); off += up(nch * (long)R * 4);
    w.pag = reinterpret_cast<int*>(p + off); off += up(nch * (long)R * 4);
    int seg_len;
    const long S = topk_segments(Ncol, &seg_len);
    const long ncand = S > 1 && S * k <= 64 ? S * k : 0;
    w.sgi = reinterpret_cast<int*>(p + off); off += up(ncand * R * 4);
    w.sgv = reinterpret_cast<float*>(p + off); off += up(ncand * R * 4);
    w.sneed = reinterpret_cast<int*>(p + off); off += up((ncand ? S : 0) * R * 4);
    w.bytes = off;
    return w;
}

extern "C" long neat_topk_ws_bytes(int R, int Ncol, int k) {
    if (R <= 0 || Ncol <= 0 || k < 1) return 256;
    return topk_carve(nullptr, R, Ncol, k).bytes + 256;
}

#define TOPK_DISPATCH(KPV, ...)                 \
    switch (KPV) {                              \
        case 1: { constexpr int KP = 1; __VA_ARGS__; } break;   \
        case 8: { constexpr int KP = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int KP = 16; __VA_ARGS__; } break; \
        default: { constexpr int KP = 32; __VA_ARGS__; } break; \
    }

extern "C" int neat_topk_rows(const float* scores, long ld_row, long ld_col, int R, int Ncol, int k, int32_t* idx,
                              float* val, void* ws, void* stream) {
    NEAT_REQUIRE(scores && idx, NEAT_ERR_ARG, "neat_topk_rows: null pointer");
    NEAT_REQUIRE(R > 0 && Ncol > 0 && k >= 1 && k <= 32 && k <= Ncol, NEAT_ERR_SHAPE,
                 "neat_topk_rows: need 1 <= k <= min(32, Ncol); got k=%d Ncol=%d", k, Ncol);
    NEAT_REQUIRE(ld_col == 1 || ld_row == 1, NEAT_ERR_SHAPE, "neat_topk_rows: one stride must be 1 (got %ld, %ld)", ld_row, ld_col);
    NEAT_REQUIRE(ws && (uintptr_t)ws % 256 == 0, NEAT_ERR_ARG, "neat_topk_rows: workspace missing or not 256-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int kp = topk_kp(k);
    const TopkWs w = topk_carve(ws, R, Ncol, k);
    // the fast tier, then the exact tier on the rows it flagged (none at all when k <= SHORT_M)
    int grid = cdiv(R, 4);
    grid = grid > 8192 ? 8192 : grid;
    const int xgrid = R > 2048 ? 2048 : R;                            // exact tier: blocks stride over the rows and take the flagged ones
    if (ld_col == 1) {
        const bool vec = ((uintptr_t)scores % 16 == 0) && (ld_row % 4 == 0);
        const int nv = cdiv(Ncol >> 2, 64);                               // 16-B loads per lane that hold a whole row
        int seg_len;
        const int S = topk_segments(Ncol, &seg_len);
        // wider rows: S segments per row through the same kernel, then a merge of the S * k candidates (every segment must
        // hold k columns, and a wave ranks at most 64 candidates)
        const bool seg = vec && S > 1 && S * k <= 64 && Ncol - (S - 1) * seg_len >= k;
        const bool reg = (vec && nv <= 20) || seg;                  // NV = 24 would need all 256 VGPRs
        if (reg) {
            const int snv = seg ? cdiv(seg_len >> 2, 64) : nv;
            int rgrid = cdiv(seg ? R * S : R, 4);
            rgrid = rgrid > 16384 ? 16384 : rgrid;
#define TOPK_REG(NVV)                                                                                                      \
    hipLaunchKernelGGL((topk_rows_reg_kernel<NVV>), dim3(rgrid), dim3(256), 0, s, scores, ld_row, R, Ncol, seg ? S : 1,     \
                       seg ? seg_len : Ncol, k, seg ? w.sgi : idx, seg ? w.sgv : val, seg ? w.sneed : w.need)
            if (snv <= 4) TOPK_REG(4);
            else if (snv <= 8) TOPK_REG(8);
            else if (snv <= 12) TOPK_REG(12);
            else if (snv <= 16) TOPK_REG(16);
            else TOPK_REG(20);
#undef TOPK_REG
            if (seg) hipLaunchKernelGGL(topk_seg_merge_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, w.sgi, w.sgv, w.sneed, R, S, k, idx, val, w.need);
        } else if (vec) hipLaunchKernelGGL((topk_rows_fast_kernel<SHORT_M, true>), dim3(grid), dim3(256), 0, s, scores, ld_row, R, Ncol, k, idx, val, w.need);
        else hipLaunchKernelGGL((topk_rows_fast_kernel<SHORT_M, false>), dim3(grid), dim3(256), 0, s, scores, ld_row, R, Ncol, k, idx, val, w.need);
        if (k > SHORT_M || reg) {
            if (vec) {
                TOPK_DISPATCH(kp, hipLaunchKernelGGL((topk_rows_kernel<KP, true>), dim3(xgrid), dim3(256), 0, s, scores, ld_row, 1L, R, Ncol, k, idx, val, w.need));
            } else {
                TOPK_DISPATCH(kp, hipLaunchKernelGGL((topk_rows_kernel<KP, false>), dim3(xgrid), dim3(256), 0, s, scores, ld_row, 1L, R, Ncol, k, idx, val, w.need));
            }
        }
    } else {
        const int nsf = topk_fast_nsplit(R, Ncol);
        const bool vec = ((uintptr_t)scores % 16 == 0) && (ld_col % 4 == 0) && (R % 4 == 0);
        const int CH = cols_chunk_len(Ncol);
        const int nchunk = cdiv(Ncol, CH);
        if (vec && nchunk >= 64) {                                       // tournament over chunks (wide transposed views)
            const int cw = cols_group_width(R);
            hipLaunchKernelGGL(topk_cols_chunk_kernel, dim3(cdiv(R, 4 * cw), nchunk), dim3(cw), 0, s, scores, ld_col, R, Ncol, CH,
                               w.pmx, w.psc, w.pag);
            const size_t lds = (size_t)nchunk * COLS_P * 12 + COLS_RB * (2 * REG_CAP + COLS_RESCAN_CAP) * 4;
            static unsigned long long attr_done = 0;
            constexpr int lds_max = COLS_MAXCHUNKS * COLS_P * 12 + COLS_RB * (2 * REG_CAP + COLS_RESCAN_CAP) * 4;   // set once, for any nchunk
            if (int rc = ensure_dynamic_lds((const void*)topk_cols_select_kernel, lds_max, &attr_done, "topk_cols_select attr")) return rc;
            hipLaunchKernelGGL(topk_cols_select_kernel, dim3(cdiv(R, COLS_RB)), dim3(COLS_RB * 64), lds, s, scores, ld_col, R, Ncol, CH, nchunk,
                               w.pmx, w.psc, w.pag, k, idx, val, w.need);
            TOPK_DISPATCH(kp, hipLaunchKernelGGL((topk_rows_kernel<KP, false>), dim3(xgrid), dim3(256), 0, s, scores, ld_row, ld_col, R, Ncol, k, idx, val, w.need));
            NEAT_LAUNCH_CHECK("neat_topk_rows");
            return NEAT_OK;
        }
        if (vec) hipLaunchKernelGGL((topk_cols_fast_kernel<SHORT_M, 4>), dim3(cdiv(R, 256), nsf), dim3(64), 0, s, scores, ld_col, R, Ncol, nsf, w.fv, w.fi, w.flost);
        else hipLaunchKernelGGL((topk_cols_fast_kernel<SHORT_M, 1>), dim3(cdiv(R, 64), nsf), dim3(64), 0, s, scores, ld_col, R, Ncol, nsf, w.fv, w.fi, w.flost);
        hipLaunchKernelGGL((topk_pairs_finish_kernel<SHORT_M>), dim3(cdiv(R, 4)), dim3(256), 0, s, w.fv, w.fi, w.flost, R, nsf, k, idx, val, w.need);
        if (k > SHORT_M) {
            TOPK_DISPATCH(kp, hipLaunchKernelGGL((topk_rows_kernel<KP, false>), dim3(xgrid), dim3(256), 0, s, scores, ld_row, ld_col, R, Ncol, k, idx, val, w.need));
        }
    }
    NEAT_LAUNCH_CHECK("neat_topk_rows");
    return NEAT_OK;
}

extern "C" int neat_recall_from_topk(const int32_t* idx, const uint8_t* pos, long ld_row, long ld_col, int R, int Ncol,
                                     int k, float* recall, void* stream) {
    NEAT_REQUIRE(idx && pos && recall, NEAT_ERR_ARG, "neat_recall_from_topk: null pointer");
    NEAT_REQUIRE(R > 0 && Ncol > 0 && k >= 1, NEAT_ERR_SHAPE, "neat_recall_from_topk: bad shape");
    NEAT_REQUIRE(ld_col == 1 || ld_row == 1, NEAT_ERR_SHAPE, "neat_recall_from_topk: one stride must be 1");
    hipStream_t s = (hipStream_t)stream;
    if (ld_col == 1) {
        int grid = cdiv(R, 4);
        grid = grid > 8192 ? 8192 : grid;
        hipLaunchKernelGGL(recall_rows_kernel, dim3(grid), dim3(256), 0, s, idx, pos, ld_row, R, Ncol, k, recall);
    } else {
        if (Ncol >= 2048) {
            constexpr int CH = 256;
            if (int rc = check_hip(hipMemsetAsync(recall, 0, (size_t)R * 4, s), "neat_recall_from_topk memset")) return rc;
            hipLaunchKernelGGL(recall_cols_count_kernel, dim3(cdiv(R, 64), cdiv(Ncol, CH)), dim3(64), 0, s, pos, ld_col, R, Ncol, CH,
                               reinterpret_cast<int*>(recall));
            hipLaunchKernelGGL(recall_cols_finish_kernel, dim3(cdiv(R, 64)), dim3(64), 0, s, idx, pos, ld_col, R, Ncol, k, recall);
        } else {
            hipLaunchKernelGGL(recall_cols_kernel, dim3(cdiv(R, 64)), dim3(64), 0, s, idx, pos, ld_col, R, Ncol, k, recall);
        }
    }
    NEAT_LAUNCH_CHECK("neat_recall_from_topk");
    return NEAT_OK;
}

extern "C" int neat_recall_owner_topk(const int32_t* idx, int R, int K, const int32_t* row_owner, const int32_t* col_owner,
                                      const int32_t* npos, const int32_t* ks, int nk, float* recall, void* stream) {
    NEAT_REQUIRE(idx && ks && recall && (row_owner || col_owner), NEAT_ERR_ARG, "neat_recall_owner_topk: null pointer");
    NEAT_REQUIRE(R > 0 && K >= 1 && nk >= 1, NEAT_ERR_SHAPE, "neat_recall_owner_topk: bad shape R=%d K=%d nk=%d", R, K, nk);
    hipLaunchKernelGGL(recall_owner_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, idx, R, K, row_owner,
                       col_owner, npos, ks, nk, recall);
    NEAT_LAUNCH_CHECK("neat_recall_owner_topk");
    return NEAT_OK;
}

extern "C" int neat_best_positive_rank(const float* scores, long ld_row, long ld_col, const uint8_t* pos, long pld_row,
                                       long pld_col, int R, int Ncol, int32_t* rank, void* stream) {
    NEAT_REQUIRE(scores && pos && rank, NEAT_ERR_ARG, "neat_best_positive_rank: null pointer");
    NEAT_REQUIRE(R > 0 && Ncol > 0, NEAT_ERR_SHAPE, "neat_best_positive_rank: empty matrix");
    hipLaunchKernelGGL(best_positive_rank_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, scores, ld_row, ld_col, pos,
                       pld_row, pld_col, R, Ncol, rank);
    NEAT_LAUNCH_CHECK("neat_best_positive_rank");
    return NEAT_OK;
}

extern "C" int neat_best_positive_rank_owner(const float* scores, long ld_row, long ld_col, const int32_t* row_owner,
                                             const int32_t* col_owner, int R, int Ncol, int32_t* rank, void* ws,
                                             void* stream) {
    NEAT_REQUIRE(scores && rank && (row_owner || col_owner) && !(row_owner && col_owner), NEAT_ERR_ARG,
                 "neat_best_positive_rank_owner: pass scores, rank and exactly one owner vector");
    NEAT_REQUIRE(R > 0 && Ncol > 0, NEAT_ERR_SHAPE, "neat_best_positive_rank_owner: empty matrix");
    NEAT_REQUIRE(ld_col == 1 || ld_row == 1, NEAT_ERR_SHAPE, "neat_best_positive_rank_owner: one stride must be 1 (got %ld, %ld)",
                 ld_row, ld_col);
    hipStream_t s = (hipStream_t)stream;
    const bool cols_fast = ws && ld_row == 1 && ld_col != 1 && (uintptr_t)ws % 256 == 0 && (uintptr_t)scores % 16 == 0 &&
                           ld_col % 4 == 0 && R % 4 == 0;
    if (cols_fast) {                                              // transposed view: lanes along the logical rows
        unsigned long long* best = reinterpret_cast<unsigned long long*>(ws);
        if (int rc = check_hip(hipMemsetAsync(best, 0, (size_t)R * 8, s), "neat_best_positive_rank_owner memset")) return rc;
        if (int rc = check_hip(hipMemsetAsync(rank, 0, (size_t)R * 4, s), "neat_best_positive_rank_owner memset")) return rc;
        hipLaunchKernelGGL(bpr_cols_best_kernel, dim3(cdiv(row_owner ? R : Ncol, 256)), dim3(256), 0, s, scores, ld_col,
                           row_owner, col_owner, R, Ncol, best);
        const int CH = 128;
        const int cw = cols_group_width(R);
        hipLaunchKernelGGL(bpr_cols_count_kernel, dim3(cdiv(R, 4 * cw), cdiv(Ncol, CH)), dim3(cw), 0, s, scores, ld_col, R, Ncol,
                           CH, best, rank);
        hipLaunchKernelGGL(bpr_cols_finish_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, best, R, rank);
    } else {
        hipLaunchKernelGGL(best_positive_rank_owner_kernel, dim3(R), dim3(256), 0, s, scores, ld_row, ld_col, row_owner,
                           col_owner, R, Ncol, rank);
    }
    NEAT_LAUNCH_CHECK("neat_best_positive_rank_owner");
    return NEAT_OK;
}

extern "C" int neat_mcq_scores(const float* img, const float* txt, float* logits, int32_t* pred, int B, int nopt, int D,
                               void* stream) {
    NEAT_REQUIRE(img && txt && pred, NEAT_ERR_ARG, "neat_mcq_scores: null pointer");
    NEAT_REQUIRE(B > 0 && nopt > 0 && D > 0, NEAT_ERR_SHAPE, "neat_mcq_scores: bad shape");
    hipLaunchKernelGGL(mcq_scores_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, img, txt, logits, pred, B,
                       nopt, D);
    NEAT_LAUNCH_CHECK("neat_mcq_scores");
    return NEAT_OK;
}

// ---- fused similarity + top-k: the offline path's `texts_emb @ images_emb.t()` (evaluation/retrieval.py:70) followed by
// torch.topk (:294-300), without the [Nq, Ng] score matrix in HBM.  The scores are produced SIM_CHUNK-bounded row chunks at
// a time by the exact-fp32 MFMA GEMM into the caller's workspace and reduced by the row kernels above before the next
// chunk overwrites them: a chunk (<= 128 MiB: two rounds of 128 x 128 GEMM tiles; 64 MiB, one round, measured 12 % slower
// end to end) stays in the 256 MiB Infinity Cache between its producer and its consumer,
// so HBM sees Q and G once per chunk and the k winners, not 2 x Nq x Ng x 4 bytes.  Bit-identical to
// neat_gemm_f32_nt + neat_topk_rows (same kernels, same order of operations).
static inline long sim_pitch(int Ng) { return ((long)Ng + 3) / 4 * 4; }   // chunk rows start 16-B aligned for the row kernels

static inline int sim_chunk_rows(int Nq, int Ng) {
    long rows = (128L << 20) / (4L * sim_pitch(Ng));
    rows = rows / 128 * 128;
    if (rows < 128) rows = 128;
    return (int)(rows < Nq ? rows : Nq);
}

extern "C" long neat_sim_topk_ws_bytes(int Nq, int Ng, int k) {
    if (Nq <= 0 || Ng <= 0 || k < 1) return 256;
    const int ch = sim_chunk_rows(Nq, Ng);
    const long sc = ((long)ch * sim_pitch(Ng) * 4 + 255) / 256 * 256;
    return sc + neat_topk_ws_bytes(ch, Ng, k);
}

extern "C" int neat_sim_topk(const float* Q, const float* G, int k, int32_t* idx_out, float* val_out, int Nq, int Ng, int D,
                             void* ws, void* stream) {
    NEAT_REQUIRE(Q && G && idx_out && ws, NEAT_ERR_ARG, "neat_sim_topk: null pointer");
    NEAT_REQUIRE((uintptr_t)ws % 256 == 0, NEAT_ERR_ARG, "neat_sim_topk: workspace not 256-B aligned");
    NEAT_REQUIRE(Nq > 0 && Ng > 0 && D > 0 && D % 16 == 0 && k >= 1 && k <= 32 && k <= Ng, NEAT_ERR_SHAPE,
                 "neat_sim_topk: Nq=%d Ng=%d D=%d k=%d (need D %% 16 == 0, 1 <= k <= min(32, Ng))", Nq, Ng, D, k);
    const int ch = sim_chunk_rows(Nq, Ng);
    float* sc = reinterpret_cast<float*>(ws);
    const long pitch = sim_pitch(Ng);
    void* tws = reinterpret_cast<char*>(ws) + ((long)ch * pitch * 4 + 255) / 256 * 256;
    for (int q0 = 0; q0 < Nq; q0 += ch) {
        const int rows = Nq - q0 < ch ? Nq - q0 : ch;
        if (int rc = launch_gemm_f32_nt(Q + (long)q0 * D, G, sc, rows, Ng, D, pitch, (hipStream_t)stream)) return rc;
        if (int rc = neat_topk_rows(sc, pitch, 1, rows, Ng, k, idx_out + (long)q0 * k, val_out ? val_out + (long)q0 * k : nullptr,
                                    tws, stream))
            return rc;
    }
    return NEAT_OK;
}
