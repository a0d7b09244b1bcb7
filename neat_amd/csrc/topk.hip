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

template <int KP>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, long ld_row, int R, int Ncol,
                                                        int k, int32_t* __restrict__ idx, float* __restrict__ val) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* row = scores + (long)r * ld_row;
        TopList<KP> tl;
        tl.init();
        for (int c = lane; c < Ncol; c += 64) tl.insert(row[c], c);
        for (int round = 0; round < k; ++round) {
            float bv = tl.v[0];
            int bi = tl.i[0];
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
    }
}

// transposed view: element (r, c) at scores[r + c*ld_col]; thread <-> logical row r, blockIdx.y <-> c-range
template <int KP>
__global__ __launch_bounds__(64) void topk_cols_partial_kernel(const float* __restrict__ scores, long ld_col, int R,
                                                               int Ncol, int nsplit, float* __restrict__ pv,
                                                               int* __restrict__ pi) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const int per = (Ncol + nsplit - 1) / nsplit;
    const int c0 = blockIdx.y * per;
    const int c1 = min(Ncol, c0 + per);
    TopList<KP> tl;
    tl.init();
    for (int c = c0; c < c1; ++c) tl.insert(scores[(long)c * ld_col + r], c);
    float* ov = pv + ((long)r * nsplit + blockIdx.y) * KP;
    int* oi = pi + ((long)r * nsplit + blockIdx.y) * KP;
#pragma unroll
    for (int j = 0; j < KP; ++j) { ov[j] = tl.v[j]; oi[j] = tl.i[j]; }
}

template <int KP>
__global__ __launch_bounds__(64) void topk_cols_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                             int R, int nsplit, int k, int32_t* __restrict__ idx,
                                                             float* __restrict__ val) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    TopList<KP> tl;
    tl.init();
    for (int s = 0; s < nsplit; ++s)
        for (int j = 0; j < KP; ++j) {
            const long o = ((long)r * nsplit + s) * KP + j;
            const int ci = pi[o];
            if (ci != IMAX) tl.insert_tie(pv[o], ci);
        }
    // static-index copy out
#pragma unroll
    for (int j = 0; j < KP; ++j)
        if (j < k) {
            idx[(long)r * k + j] = tl.i[j];
            if (val) val[(long)r * k + j] = tl.v[j];
        }
}

// recall[r] = (# of top-k indices that are positives) / (# positives in row r)
__global__ __launch_bounds__(256) void recall_rows_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ pos,
                                                          long ld_row, int R, int Ncol, int k, float* __restrict__ recall) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const uint8_t* row = pos + (long)r * ld_row;
        float n = 0.f;
        for (int c = lane; c < Ncol; c += 64) n += row[c] ? 1.f : 0.f;
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

extern "C" long neat_topk_ws_bytes(int R, int Ncol, int k) { return (long)R * topk_nsplit(Ncol) * topk_kp(k) * 8 + 256; }

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
    hipStream_t s = (hipStream_t)stream;
    const int kp = topk_kp(k);
    if (ld_col == 1) {
        int grid = cdiv(R, 4);
        grid = grid > 8192 ? 8192 : grid;
        TOPK_DISPATCH(kp, hipLaunchKernelGGL(topk_rows_kernel<KP>, dim3(grid), dim3(256), 0, s, scores, ld_row, R, Ncol, k, idx, val));
    } else {
        NEAT_REQUIRE(ws, NEAT_ERR_ARG, "neat_topk_rows: transposed view needs a workspace");
        const int ns = topk_nsplit(Ncol);
        float* pv = reinterpret_cast<float*>(ws);
        int* pi = reinterpret_cast<int*>(pv + (long)R * ns * kp);
        TOPK_DISPATCH(kp, hipLaunchKernelGGL(topk_cols_partial_kernel<KP>, dim3(cdiv(R, 64), ns), dim3(64), 0, s, scores, ld_col, R, Ncol, ns, pv, pi);
                      hipLaunchKernelGGL(topk_cols_merge_kernel<KP>, dim3(cdiv(R, 64)), dim3(64), 0, s, pv, pi, R, ns, k, idx, val));
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
        hipLaunchKernelGGL(recall_cols_kernel, dim3(cdiv(R, 64)), dim3(64), 0, s, idx, pos, ld_col, R, Ncol, k, recall);
    }
    NEAT_LAUNCH_CHECK("neat_recall_from_topk");
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

extern "C" int neat_mcq_scores(const float* img, const float* txt, float* logits, int32_t* pred, int B, int nopt, int D,
                               void* stream) {
    NEAT_REQUIRE(img && txt && pred, NEAT_ERR_ARG, "neat_mcq_scores: null pointer");
    NEAT_REQUIRE(B > 0 && nopt > 0 && D > 0, NEAT_ERR_SHAPE, "neat_mcq_scores: bad shape");
    hipLaunchKernelGGL(mcq_scores_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, img, txt, logits, pred, B,
                       nopt, D);
    NEAT_LAUNCH_CHECK("neat_mcq_scores");
    return NEAT_OK;
}
