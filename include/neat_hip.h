/* libneat_hip.so -- C ABI of the MI355X (gfx950) kernels behind NEAT's test-time-adaptation path.
 *
 * The reference (hhc1997/NEAT) has no native layer: its "kernels" are ATen ops reached from Python
 * (SURVEY.md §2c).  Each entry point below replaces the ATen call site(s) named in its comment
 * (paths relative to the reference's src/).  INTEGRATION.md shows the ctypes stubs the reference
 * would add to bind them.
 *
 * Conventions
 *  - Every pointer is a raw DEVICE address (tensor.data_ptr()); buffers are contiguous, row-major and
 *    owned by the caller, including workspaces.  The library never allocates or frees device memory.
 *  - bf16 values are passed as uint16_t bit patterns.
 *  - Every call enqueues on `stream` (a hipStream_t passed as void*) and returns without synchronising.
 *  - Return value: 0 = OK, negative = error (enum below); neat_last_error() returns the message of
 *    the calling thread's last failure.  No C++ exception crosses the boundary.
 */
#ifndef NEAT_HIP_H
#define NEAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    NEAT_OK = 0,
    NEAT_ERR_SHAPE = -1,       /* shape / alignment not supported by the kernel */
    NEAT_ERR_ARG = -2,         /* null pointer or invalid enum */
    NEAT_ERR_HIP = -3,         /* HIP runtime error at launch */
    NEAT_ERR_COMM = -4         /* collective error: RCCL missing, no communicator, or an RCCL call failed (neat_comm_*) */
};

/* GEMM epilogues (neat_gemm_bf16) */
enum {
    NEAT_EPI_BF16 = 0,           /* out0(bf16) = acc                      (dgrad GEMMs)                 */
    NEAT_EPI_BIAS_BF16 = 1,      /* out0(bf16) = acc + bias[n]            (attn in_proj)                */
    NEAT_EPI_BIAS_RES_F32 = 2,   /* out0(f32)  = acc + bias[n] + res[m,n] (out_proj / c_proj + residual)*/
    NEAT_EPI_BIAS_GELU = 3,      /* out0(bf16) = h = acc + bias[n]; out1(bf16) = QuickGELU(h)  (c_fc)   */
    NEAT_EPI_GELU_BWD = 4,       /* out0(bf16) = acc * QuickGELU'(aux[m,n]), aux bf16 = saved h         */
    NEAT_EPI_F32 = 5             /* out0(f32)  = acc                      (patch embedding)             */
};

const char* neat_version(void);
const char* neat_last_error(void);

/* ---- LayerNorm: F.layer_norm at open_clip/transformer.py:24-30 (eps inside sqrt, biased var) ---- */
/* x f32 [rows,cols] -> y bf16 [rows,cols], mean/rstd f32 [rows] (either may be NULL).
 * y_f32 (optional, may be NULL) receives the fp32 result instead of/in addition to y. */
int neat_ln_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* y_f32,
                float* mean, float* rstd, int rows, int cols, float eps, void* stream);
/* Backward.  dy bf16 [rows,cols]; dres f32 (may be NULL) is the residual-branch gradient that is added
 * to the LN data gradient: dx = dres + LN'(dy).  dx_f32 / dx_bf16 outputs (either may be NULL; dx_f32
 * may alias dres).  partial f32 [nblk, 2, cols] receives per-block sums of dy*xhat and dy
 * (nblk = neat_ln_bwd_blocks(rows)); pass NULL to skip parameter gradients (ln_final).  With BOTH dx outputs NULL (and
 * partial given) only the parameter sums are formed: the first block's ln_1 of a tower whose embeddings are frozen. */
int neat_ln_bwd_blocks(int rows);
int neat_ln_bwd(const uint16_t* dy, const float* x, const float* gamma, const float* mean,
                const float* rstd, const float* dres, float* dx_f32, uint16_t* dx_bf16, float* partial,
                int rows, int cols, void* stream);
/* dgamma[c] (+)= sum_b partial[b,0,c]; dbeta[c] (+)= sum_b partial[b,1,c]  (fixed order -> deterministic) */
int neat_ln_bwd_finalize(const float* partial, int nblk, int cols, float* dgamma, float* dbeta,
                         int accumulate, void* stream);
/* the same for `count` LayerNorms in one launch (a whole backward pass): partial [count, nblk, 2, cols],
 * dst = DEVICE array of 2*count device pointers {dgamma_0, dbeta_0, dgamma_1, ...}; same fixed summation order */
int neat_ln_bwd_finalize_batched(const float* partial, int nblk, int cols, int count, float* const* dst,
                                 int accumulate, void* stream);

/* ---- bf16 MFMA GEMM: nn.Linear / MHA projections, open_clip/transformer.py:209-213,227-229 ---- */
/* Frozen weights are packed ONCE into MFMA-fragment order (bf16).  Logical matrix Wl[N,K]:
 *   transposed == 0: Wl = src, row-major f32 [N,K]           (forward: nn.Linear.weight [out,in])
 *   transposed != 0: Wl[n][k] = src[k*N + n], src f32 [K,N]  (data gradient: dX = dY . W needs W^T)
 * dst holds N*K bf16: dst[((nt*(K/32)+ks)*64+lane)*8+j] = Wl[nt*16+(lane&15)][ks*32+(lane>>4)*8+j].
 * N % 16 == 0, K % 32 == 0. */
int neat_pack_weight_bf16(const float* src, uint16_t* dst, int N, int K, int transposed, void* stream);
/* out[M,N] = epilogue(A[M,K] * Wl[N,K]^T).  A bf16 row-major (K-contiguous); Wp = packed Wl (above);
 * K % 128 == 0, N % 16 == 0.  bias f32 [N]; res f32 [M,N] (may alias out0); aux bf16 [>=M,N].
 * Unused pointers may be NULL. */
int neat_gemm_bf16(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                   const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                   void* stream);
/* The same GEMM in its second form ("tile256": 256 x 256 units, one workgroup per CU, both operands by LDS-DMA, see
 * csrc/gemm.hip), for N % 256 == 0, K % 128 == 0, N <= 4096 (wider layers take neat_gemm_bf16).  Its weights are packed into the kernel's LDS image:
 * 16-KiB half-tiles ((cb*(K/64) + kt)*2 + h), each 128 rows x 8 chunks of 16 B; row lr holds column
 * cb*256 + (lr>>5)*64 + h*32 + (lr&31), chunk p holds k = kt*64 + (p ^ swz(lr))*8 .. +7,
 * swz(lr) = ((lr>>1)&7) ^ (((lr>>4)&1)<<1).  Same arguments, epilogues and results as neat_gemm_bf16 (bf16 operands,
 * fp32 accumulation in k order per 32-k step).  neat_gemm_prefers_t256 returns 1 where this form is the faster one
 * on MI355X (store-only epilogues on grids of >= 1.5 rounds of units, every epilogue from 768 x 3072 weights on); a
 * caller keeps the layout(s) it will use. */
int neat_pack_weight_bf16_t256(const float* src, uint16_t* dst, int N, int K, int transposed, void* stream);
int neat_gemm_bf16_t256(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res,
                        const uint16_t* aux, void* out0, void* out1, int M, int N, int K, int epilogue,
                        void* stream);
int neat_gemm_prefers_t256(int M, int N, int K, int epilogue);
/* The residual GEMM with the FOLLOWING LayerNorm folded into its epilogue (open_clip/transformer.py:241-242: the residual
 * add of out_proj / c_proj, then ln_2 / the next block's ln_1, :15-30): x_out f32 [M,N] = A Wl^T + bias + res;
 * y bf16 [M,N] = LayerNorm(x_out) * gamma + beta (biased variance, eps inside the sqrt, fp32 statistics); mean, rstd f32 [M]
 * as neat_ln_fwd saves them.  Saves the separate LayerNorm launch and its re-read of x_out.  Units of 64 rows x N columns
 * (csrc/gemm_rowln.hip); N == 512 only, K % 128 == 0, K >= 256; Wp = neat_pack_weight_bf16 layout; res must not alias
 * x_out.  Other widths: neat_gemm_bf16(NEAT_EPI_BIAS_RES_F32) + neat_ln_fwd. */
int neat_gemm_bf16_res_ln(const uint16_t* A, const uint16_t* Wp, const float* bias, const float* res, const float* gamma,
                          const float* beta, float* x_out, uint16_t* y, float* mean, float* rstd, int M, int N, int K,
                          float eps, void* stream);
/* NEAT_EPI_BIAS_GELU on tile256 for a forward pass whose LAST rows have no backward pass (the `neg` captions of the TTA
 * step, reference train_TTA.py:35: coarse alignment runs under no_grad): out1 bf16 [M,N] = QuickGELU(A Wl^T + bias) for
 * all rows, the pre-activation out0 bf16 [out0_rows,N] only for the first out0_rows <= M rows. */
int neat_gemm_bf16_t256_gelu_rows(const uint16_t* A, const uint16_t* Wp, const float* bias, void* out0, int out0_rows,
                                  void* out1, int M, int N, int K, void* stream);

/* ---- image preprocessing of the TTA gallery loader (reference src/data/__init__.py:14-20: torchvision
 * Resize((S,S), BICUBIC) on a PIL image -> ToTensor -> Normalize; the resize is Pillow's Image.resize, restated
 * bit-exactly: horizontal then vertical 8-bit pass, antialiased bicubic, 22-bit fixed-point coefficients) ---- */
/* src u8 [H, W, 3] (RGB, row-major) -> dst f32 [3, S, S] = ((resized / 255) - mean[c]) / std[c].
 * workspace: neat_image_preprocess_ws_bytes(H, W, S) bytes of device memory (coefficient tables + the 8-bit
 * intermediate), caller-owned, 256-B aligned. */
long neat_image_preprocess_ws_bytes(int H, int W, int S);
int neat_image_preprocess(const uint8_t* src, int H, int W, float* dst, int S, float mean0, float mean1, float mean2,
                          float std0, float std1, float std2, void* workspace, void* stream);
/* n images of mixed sizes in one launch set (a gallery batch): srcs = DEVICE array of n device pointers (u8 [H_i,W_i,3]),
 * hw = DEVICE int32 [n][2] = (H_i, W_i); max_h / max_w = the largest of them (host values); dst f32 [n, 3, S, S];
 * workspace = n * neat_image_preprocess_ws_bytes(max(max_h,S), max(max_w,S), S) bytes.  Bit-identical to n single calls. */
int neat_image_preprocess_batch(const uint8_t* const* srcs, const int32_t* hw, int n, int max_h, int max_w, float* dst,
                                int S, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                                void* workspace, void* stream);

/* ---- video-frame preprocessing of the MSR-VTT gallery loader (replaces the host transform of src/data/__init__.py:22-26:
 * ConvertImageDtype(float32) -> Resize((S,S), BICUBIC, antialias=True) on a tensor -> Normalize; the resize is ATen's
 * upsample_bicubic2d_aa: separable float convolution, cubic a = -0.5, horizontal pass first, result not clamped) ---- */
/* src u8 [T, H, W, 3] (decoder layout: frames, rows, RGB) -> dst f32 [T, 3, S, S] = (resize(src / 255) - mean[c]) / std[c].
 * workspace: neat_frames_preprocess_ws_bytes(T, H, W, S) bytes (coefficient tables + the fp32 intermediate [T,3,H,S]),
 * caller-owned, 256-B aligned. */
long neat_frames_preprocess_ws_bytes(int T, int H, int W, int S);
int neat_frames_preprocess(const uint8_t* src, int T, int H, int W, float* dst, int S, float mean0, float mean1, float mean2,
                           float std0, float std1, float std2, void* workspace, void* stream);

/* ---- attention core of nn.MultiheadAttention, open_clip/transformer.py:202,227-229,655-661 ---- */
/* qkv bf16 [rows, 3*C] (per token: q | k | v, heads of 64 inside each C); out bf16 [rows, C];
 * lse f32 [nseq, heads, L] (log-sum-exp of the scaled scores, saved for backward; may be NULL).
 * cu_seqlens == NULL: fixed-length batch, rows = nseq*L, sequence s owns rows s*L .. s*L+L-1.
 * cu_seqlens i32 [nseq+1] (packed batch, see neat_pack_plan): sequence s owns rows cu[s] .. cu[s+1]-1, L is the
 * maximum length (<= 80) and the row pitch of lse.
 * causal: bit 0 = causal mask (NEAT_ATTN_CAUSAL); bit 1 = head-major qkv / dqkv columns (NEAT_ATTN_HEAD_MAJOR): per token
 * [head][q | k | v][64] instead of the nn.MultiheadAttention order [q | k | v][head][64], i.e. the in_proj weight rows
 * permuted once before packing -- a head's three pieces are then one 384-B run per token (attention 8 % faster); out /
 * dout keep the [rows, C] order either way. */
#define NEAT_ATTN_CAUSAL 1
#define NEAT_ATTN_HEAD_MAJOR 2
int neat_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens, int nseq, int L,
                  int heads, int causal, void* stream);
/* dout bf16 [rows, C] -> dqkv bf16 [rows, 3*C]; L <= 80 */
int neat_attn_bwd(const uint16_t* qkv, const uint16_t* dout, const float* lse, uint16_t* dqkv,
                  const int32_t* cu_seqlens, int nseq, int L, int heads, int causal, void* stream);

/* ---- text embedding: nn.Embedding + positional add, open_clip/model.py:272-274 ---- */
int neat_embed_fwd(const int64_t* tokens, const float* tok_emb, const float* pos_emb, float* x, int nseq,
                   int L, int C, int vocab, void* stream);
/* argmax over each token row (EOT position), open_clip/transformer.py:555-558; first max wins */
int neat_eot_index(const int64_t* tokens, int32_t* eot, int nseq, int L, void* stream);

/* ---- packed text batches.  No reference counterpart as code: the reference encodes all 77 positions
 *      (open_clip/model.py:269-286); with its causal mask (transformer.py:655-661) and EOT pooling (:555-558)
 *      positions after the EOT token can neither influence the pooled feature nor receive gradient, so they
 *      may be dropped without changing any output of the path. ---- */
/* tokens i64 [nseq, L] -> eot i32 [nseq] (argmax, first max), cu_seqlens i32 [nseq+1] = exclusive prefix sums of
 * eot+1 (cu[nseq] = packed row count).  nseq <= 12000. */
int neat_pack_plan(const int64_t* tokens, int32_t* eot, int32_t* cu_seqlens, int nseq, int L, void* stream);
/* x f32 [cu[nseq], C]: row cu[s]+p = tok_emb[tokens[s,p]] + pos_emb[p] for p <= eot[s] */
int neat_embed_packed_fwd(const int64_t* tokens, const int32_t* cu_seqlens, const float* tok_emb,
                          const float* pos_emb, float* x, int nseq, int L, int C, int vocab, void* stream);

/* ---- pooled head: ln_final/ln_post on the pooled row -> @ projection -> F.normalize
 *      open_clip/model.py:278-286 (text), open_clip/transformer.py:538-542 + model.py:267 (vision) ---- */
/* x f32 [nseq*L, C]; row index per sequence (NULL -> row 0 = class token); proj_t f32 [D, C] = the TRANSPOSE of
 * the projection parameter (frozen: transposed once by the caller).  L == 0: row_idx holds ABSOLUTE rows of x
 * (packed batches).  Outputs: feat f32 [nseq, D] (normalised iff normalize), and for backward: xe f32 [nseq,C]
 * (pooled rows), mean/rstd [nseq], nrm [nseq] (L2 norm before clamping); backward-only outputs may be NULL.
 * ws: neat_pool_proj_ws_floats(nseq, C, D) floats.  C % 16 == 0. */
long neat_pool_proj_ws_floats(int nseq, int C, int D);
int neat_pool_proj_fwd(const float* x, const int32_t* row_idx, const float* gamma, const float* beta,
                       const float* proj_t, float* feat, float* xe, float* mean, float* rstd, float* nrm, float* ws,
                       int nseq, int L, int C, int D, int normalize, float eps, void* stream);
/* dfeat f32 [nseq, D] -> gradient rows scattered into dx_f32 / dx_bf16 [rows, C] (must be zeroed by the caller);
 * proj f32 [C, D] (the parameter as stored); ln_final itself is not adapted (open_clip/factory.py:405-407).
 * D % 16 == 0. */
int neat_pool_proj_bwd(const float* dfeat, const float* feat, const float* nrm, const float* xe,
                       const float* mean, const float* rstd, const float* gamma, const float* proj,
                       const int32_t* row_idx, float* dx_f32, uint16_t* dx_bf16, float* ws, int nseq, int L, int C,
                       int D, int normalize, void* stream);

/* ---- vision stem: conv1 (stride = kernel) as a GEMM, open_clip/transformer.py:362,503-513 ---- */
/* images f32 NCHW [n,3,H,W] -> patches bf16 [n*gh*gw, Kpad], (c,kh,kw) order, zero padded to Kpad */
int neat_patchify(const float* images, uint16_t* patches, int n, int H, int W, int patch, int Kpad,
                  void* stream);
/* x[b,0]=cls+pos[0]; x[b,1+p]=patch_emb[b,p]+pos[1+p]; then ln_pre -> x f32 [n*(P+1), C] */
int neat_vision_embed(const float* patch_emb, const float* cls, const float* pos, const float* gamma,
                      const float* beta, float* x, int n, int P, int C, float eps, void* stream);
/* VideoCLIP.encode_video pooling, utils/video_utils/model.py:62-74,90-91: feat f32 [n*F, D] (already
 * per-frame normalised) -> out[n, D] = normalize(mean_f feat) */
int neat_frame_mean_norm(const float* feat, float* out, int n, int F, int D, void* stream);

/* ---- coarse alignment, train_TTA.py:31-42 ---- */
/* q,pos,neg f32 [B,D]; G f32 [N,D].  S_out f32 [B,N] = q G^T; idx_out i32 [B] = argmax_j
 * (pos G^T)(1-relu(neg G^T)) (first max); coarse_out f32 [B,D] = G[idx].  ws f32 workspace of
 * neat_coarse_ws_floats(B,N) floats. */
long neat_coarse_ws_floats(int B, int N);
int neat_coarse_align(const float* q, const float* pos, const float* neg, const float* G, float* S_out,
                      int32_t* idx_out, float* coarse_out, float* ws, int B, int N, int D, void* stream);

/* Image-to-text coarse alignment, train_TTA.py:20-28 (coarse_aligning_i2t -- defined by the reference and never called
 * by it; built for the `--tta-retrieval i2t` direction SURVEY 8(f4) lists).  q f32 [B,D] (image queries); G, Gp, Gn f32
 * [N,D] (caption, positive-caption and negated-caption galleries).  idx_out i32 [B] = argmax_j
 * (q Gp^T)(1 - sigmoid(q Gn^T)) (first max); coarse_out f32 [B,D] = G[idx].  ws: neat_coarse_ws_floats(B,N) floats. */
int neat_coarse_align_i2t(const float* q, const float* G, const float* Gp, const float* Gn, int32_t* idx_out,
                          float* coarse_out, float* ws, int B, int N, int D, void* stream);

/* The same computation with NO workspace argument and NO allocation of any kind (no hipMalloc / hipFree, no allocating
 * library call).  It has no counterpart in the reference: it exists for callers that cannot allocate -- code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  The per-tile partial results live in a 2-MiB
 * file-scope device array that every GPU the library runs on holds until the process ends.  Limit:
 * B * ceil(N / 64) <= neat_coarse_static_max_slots() (= 262 144, e.g. B = 2048 with N = 8192); larger problems return
 * NEAT_ERR_SHAPE.  All calls on one GPU share that array: calls that overlap on one GPU (different streams, or two host
 * threads) race.  Set-up: the array's device address is looked up once per device (hipGetSymbolAddress, a synchronous
 * call that may load the code object) -- by neat_coarse_static_init() with that device current, or else by the first
 * neat_coarse_align_static call on the device.  Under HIP graph capture call neat_coarse_static_init() (or make one
 * warm-up call) on every device BEFORE the capture begins. */
long neat_coarse_static_max_slots(void);
int neat_coarse_static_init(void);
int neat_coarse_align_static(const float* q, const float* pos, const float* neg, const float* G, float* S_out,
                             int32_t* idx_out, float* coarse_out, int B, int N, int D, void* stream);

/* ---- TTA loss forward + gradient, train_TTA.py:138-143 with :45-48, :53-65, :77-91 ---- */
/* text,pos,inv f32 [Bl,D] are this rank's rows [row0,row0+Bl) of the global batch; text_all/coarse_all
 * f32 [B,D] are all B rows (== text/coarse when single-GPU).  losses f32 [3]: [0] = entropy term of the
 * GLOBAL batch (identical on every rank); [1], [2] = this rank's share of the sr / triplet terms (sum over
 * its rows, already divided by B) -- their sum over ranks is the global value, and with Bl == B they ARE the
 * global values.  dtext,dpos,dinv f32 [Bl,D] = d(total loss)/d(local rows).
 * ws f32 workspace of neat_loss_ws_floats(B,Bl) floats. */
long neat_loss_ws_floats(int B, int Bl);
int neat_tta_loss_fwd_bwd(const float* text, const float* pos, const float* inv, const float* text_all,
                          const float* coarse_all, float T1, float T2, float* losses, float* dtext,
                          float* dpos, float* dinv, float* ws, int B, int Bl, int row0, int D,
                          void* stream);

/* ---- AdamW, torch.optim.AdamW(main_TTA.py:271) single-tensor update rule on a flat buffer ---- */
/* hyper-parameters are doubles: torch forms 1-beta, lr*wd, lr/bias_correction in double before the fp32 update */
int neat_adamw_flat(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                    double beta2, double eps, double wd, int step, void* stream);

/* ---- retrieval scoring, evaluation/retrieval.py:280-301 ---- */
/* Row-wise top-k of a strided f32 matrix (element (r,c) at scores[r*ld_row + c*ld_col]; one of the
 * strides must be 1 -> accepts scores and scores.T).  idx i32 [R,k] descending, ties -> lowest index;
 * val f32 [R,k] (may be NULL).  k <= 32.  ws: neat_topk_ws_bytes(R,Ncol,k) bytes, 256-B aligned, always required
 * (row flags + partial lists).  Selection runs in two tiers with identical results: short per-lane lists that prove
 * their own sufficiency (HBM-bound), then the exact k-entry lists on the rows that could not prove it. */
long neat_topk_ws_bytes(int R, int Ncol, int k);
int neat_topk_rows(const float* scores, long ld_row, long ld_col, int R, int Ncol, int k, int32_t* idx,
                   float* val, void* ws, void* stream);
/* recall[r] = |topk(r) ∩ positives(r)| / |positives(r)| with positives a strided uint8/bool matrix */
int neat_recall_from_topk(const int32_t* idx, const uint8_t* pos, long ld_row, long ld_col, int R,
                          int Ncol, int k, float* recall, void* stream);
/* Recall for several cut-offs from ONE top-K list per row, positives given as owner vectors (the caption -> image list
 * the reference builds its bool matrix from, evaluation/retrieval.py:229-240) instead of a dense matrix: column c is a
 * positive of row r iff row_owner[r] == c (text -> image; row_owner i32 [R] or NULL) or col_owner[c] == r (image -> text;
 * col_owner i32 [Ncol] or NULL).  idx i32 [R,K] from neat_topk_rows; npos i32 [R] = positives per row (NULL: 1);
 * ks i32 [nk] ascending cut-offs (DEVICE array); recall f32 [nk, R] = |top-ks[q](r) ∩ positives(r)| / npos[r], the value
 * recall_at_k returns (:298-300).  One pass over the score matrix then serves k = 1, 5, 10 (:243-250 makes three). */
int neat_recall_owner_topk(const int32_t* idx, int R, int K, const int32_t* row_owner, const int32_t* col_owner,
                           const int32_t* npos, const int32_t* ks, int nk, float* recall, void* stream);
/* median-rank support, evaluation/retrieval.py:186-214: rank[r] (1-based, scores descending, equal scores by
 * index) of the best-ranked positive of row r; 0 if the row has none.  scores / pos strided like neat_topk_rows. */
int neat_best_positive_rank(const float* scores, long ld_row, long ld_col, const uint8_t* pos, long pld_row,
                            long pld_col, int R, int Ncol, int32_t* rank, void* stream);
/* the same rank with the positives as ONE owner vector (see neat_recall_owner_topk): row_owner i32 [R] (the positive of
 * row r is column row_owner[r]) or col_owner i32 [Ncol] (the positives of row r are the columns c with col_owner[c] == r).
 * ws: 8 * R bytes of device scratch, 256-B aligned, or NULL.  With ws, a transposed view (ld_row == 1; the image -> text
 * direction of :248-250) is read once, coalesced, with lanes along the logical rows; without it every logical row is a
 * strided column read by its own workgroup (13x slower at 5 000 x 25 014). */
int neat_best_positive_rank_owner(const float* scores, long ld_row, long ld_col, const int32_t* row_owner,
                                  const int32_t* col_owner, int R, int Ncol, int32_t* rank, void* ws, void* stream);
/* multiple-choice scoring, evaluation/mcq.py:92-97, :199-205: img f32 [B,D], txt f32 [nopt,B,D] ->
 * logits f32 [B,nopt] = einsum('bf,nbf->bn') (may be NULL), pred i32 [B] = argmax_n (first max) */
int neat_mcq_scores(const float* img, const float* txt, float* logits, int32_t* pred, int B, int nopt, int D,
                    void* stream);
/* fp32 similarity GEMM C[M,N] = A[M,K] B[N,K]^T (exact fp32 FMA chain; feeds argmax/top-k) */
int neat_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, void* stream);
/* Fused similarity + top-k for the offline path: replaces `texts_emb @ images_emb.t()` (evaluation/retrieval.py:70) and
 * the torch.topk of recall_at_k (:294-300) in one call that never holds the [Nq, Ng] score matrix in HBM.  Q f32 [Nq,D],
 * G f32 [Ng,D] (row-major, D % 16 == 0) -> idx i32 [Nq,k] (rows of G, best first, ties -> lowest index), val f32 [Nq,k]
 * (may be NULL).  The scores go through ws (neat_sim_topk_ws_bytes bytes, 256-B aligned) in row chunks of at most 128 MiB,
 * which stay in the Infinity Cache between the exact-fp32 GEMM that writes them and the selection that reads them.
 * Results are bit-identical to neat_gemm_f32_nt followed by neat_topk_rows.  For the other direction swap Q and G. */
long neat_sim_topk_ws_bytes(int Nq, int Ng, int k);
int neat_sim_topk(const float* Q, const float* G, int k, int32_t* idx_out, float* val_out, int Nq, int Ng, int D,
                  void* ws, void* stream);

/* ---- collectives of the data-parallel step, one RCCL communicator per process (= per GPU).  Stand in
 * for the reference's torch.distributed helpers (src/ddp.py:6-70: all_gather of features, gradients summed over ranks)
 * for callers that bind this library without torch.distributed.  RCCL is resolved at run time (dlopen): the library
 * loads without it and these calls then return NEAT_ERR_COMM.
 *   neat_comm_unique_id : rank 0 fills id (NEAT_COMM_ID_BYTES bytes) and hands it to the other ranks out of band
 *                         (file, socket, MPI, a torch.distributed broadcast ...)
 *   neat_comm_init      : every rank, with its GPU current (hipSetDevice), the same id, world size and its rank; blocks
 *                         until all ranks have called it
 *   neat_comm_allgather : recv[r*bytes .. (r+1)*bytes) = rank r's send[0 .. bytes)  (device pointers, same bytes on every
 *                         rank; used for the [text | coarse] feature shards and the gallery embeddings)
 *   neat_comm_allreduce_f32 : recv = sum over ranks of send (n floats, in place allowed; the flat LN-gradient buffer)
 *   neat_comm_destroy   : frees the communicator (no-op without one)
 * All collectives are enqueued on `stream` and do not synchronise the host. */
#define NEAT_COMM_ID_BYTES 128
int neat_comm_unique_id(void* id);
int neat_comm_init(const void* id, int world, int rank);
int neat_comm_world(void);
int neat_comm_rank(void);
int neat_comm_allgather(const void* send, void* recv, long bytes, void* stream);
int neat_comm_allreduce_f32(const float* send, float* recv, long n, void* stream);
int neat_comm_destroy(void);

/* f32 -> bf16 cast (weights at load time, n elements) */
int neat_cast_bf16(const float* src, uint16_t* dst, long n, void* stream);
/* f32 [R,C] -> bf16 [C,R] transpose-cast */
int neat_transpose_cast_bf16(const float* src, uint16_t* dst, int R, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEAT_HIP_H */
