"""Tensor-level wrappers over the C ABI (include/neat_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation is a HIP kernel of
libneat_hip.so.  Wrappers validate dtype / contiguity / device (the kernels index raw pointers) and
raise RuntimeError on any failure -- there is no eager fallback.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import lib

bf16, f32, i32, i64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


# bench.py installs a KernelTimer here to bracket ONE kernel's launches with HIP events (on the stream the
# kernel runs on) during the timed region; None in normal operation.
KERNEL_TIMER = None


class KernelTimer:
    """Brackets the launches of one GEMM instance (epilogue id) with HIP events on the current stream."""

    def __init__(self, epilogue: int):
        self.epilogue = epilogue
        self.events = []
        self.shapes = []
        self.saved_rows = []            # rows of the launch whose pre-activation was stored (BIAS_GELU: <= M)
        self.enabled = False

    def match(self, kind, epilogue, M, N, K):
        return self.enabled and kind == "gemm" and epilogue == self.epilogue

    def region(self, M, N, K, saved_rows=None):
        timer = self

        class _R:
            def __enter__(self_r):
                self_r.a = torch.cuda.Event(enable_timing=True)
                self_r.b = torch.cuda.Event(enable_timing=True)
                self_r.a.record(torch.cuda.current_stream())

            def __exit__(self_r, *exc):
                self_r.b.record(torch.cuda.current_stream())
                timer.events.append((self_r.a, self_r.b))
                timer.shapes.append((M, N, K))
                timer.saved_rows.append(M if saved_rows is None else saved_rows)
        return _R()

    def summary(self):
        """(avg ms per launch, avg algorithmic FLOP per launch, n launches) -- call after a sync."""
        if not self.events:
            return 0.0, 0.0, 0
        ms = [a.elapsed_time(b) for a, b in self.events]
        fl = [2.0 * m * n * k for m, n, k in self.shapes]
        return sum(ms) / len(ms), sum(fl) / len(fl), len(ms)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The raw handle of torch's current stream on the current device.  Every library call passes it, ~530 times per
    adaptation batch: torch.cuda.current_stream() builds a Stream object behind several availability checks (8 us per call,
    a third of the host time of a packed step under cProfile, tools/host_enqueue_probe.py); the two C accessors it ends in
    are called directly when this torch has them."""
    if _raw_stream is not None and _cur_device is not None:
        try:
            return _raw_stream(_cur_device())
        except RuntimeError:                      # before torch has initialised the GPU: take the slow, initialising path
            pass
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (libneat_hip has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


# ---- LayerNorm ----
def ln_fwd(x, gamma, beta, eps=1e-5, out_bf16=True, out_f32=False, save_stats=True):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, dtype=bf16, device=x.device) if out_bf16 else None
    yf = torch.empty(x.shape, dtype=f32, device=x.device) if out_f32 else None
    mean = torch.empty(rows, dtype=f32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=f32, device=x.device) if save_stats else None
    lib.call("neat_ln_fwd", _p(x, f32, "x"), _p(gamma, f32, "gamma"), _p(beta, f32, "beta"), _p(y), _p(yf),
             _p(mean), _p(rstd), rows, cols, eps, _stream())
    return y, yf, mean, rstd


def ln_bwd(dy, x, gamma, mean, rstd, dres=None, want_f32=True, want_bf16=True, dgamma=None, dbeta=None,
           accumulate=False, inplace_res=False, partial=None):
    """Returns (dx_f32, dx_bf16).  If dgamma/dbeta are given they receive (or accumulate) the parameter grads.
    ``partial`` (fp32 [ln_bwd_blocks(rows), 2, cols], caller-owned): leave the per-block partial sums there and skip
    the finalize pass -- the caller reduces a whole backward pass at once with ``ln_bwd_finalize_batched``."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dxf = (dres if (inplace_res and dres is not None) else torch.empty(x.shape, dtype=f32, device=x.device)) \
        if want_f32 else None
    dxb = torch.empty(x.shape, dtype=bf16, device=x.device) if want_bf16 else None
    deferred = partial is not None
    if dgamma is not None and not deferred:
        partial = torch.empty(ln_bwd_blocks(rows), 2, cols, dtype=f32, device=x.device)
    if deferred and tuple(partial.shape) != (ln_bwd_blocks(rows), 2, cols):
        raise RuntimeError(f"ln_bwd: partial shape {tuple(partial.shape)} != {(ln_bwd_blocks(rows), 2, cols)}")
    lib.call("neat_ln_bwd", _p(dy, bf16, "dy"), _p(x, f32, "x"), _p(gamma, f32, "gamma"), _p(mean, f32), _p(rstd, f32),
             _p(dres, f32), _p(dxf), _p(dxb), _p(partial, f32, "partial") if partial is not None else None, rows, cols,
             _stream())
    if dgamma is not None and not deferred:
        lib.call("neat_ln_bwd_finalize", _p(partial), partial.shape[0], cols, _p(dgamma, f32, "dgamma"),
                 _p(dbeta, f32, "dbeta"), int(accumulate), _stream())
    return dxf, dxb


def ln_bwd_blocks(rows):
    return lib.load().neat_ln_bwd_blocks(rows)


_PTR_TABLES = {}


def ln_bwd_finalize_batched(partials, dsts, accumulate=False):
    """partials fp32 [count, nblk, 2, cols]; dsts = [(dgamma_0, dbeta_0), ...] fp32 [cols] tensors: one launch reduces
    them all (fixed order, deterministic).  The device table of destination pointers is cached per destination set."""
    count, nblk, _, cols = partials.shape
    if len(dsts) != count:
        raise RuntimeError("ln_bwd_finalize_batched: one (dgamma, dbeta) pair per partial block expected")
    ptrs = []
    for dg, db in dsts:
        for t in (dg, db):
            if t.dtype != f32 or t.numel() != cols or not t.is_contiguous() or t.device != partials.device:
                raise RuntimeError("ln_bwd_finalize_batched: destinations must be contiguous fp32 [cols] on the device")
            ptrs.append(t.data_ptr())
    key = (partials.device.index, tuple(ptrs))
    table = _PTR_TABLES.get(key)
    if table is None:
        if len(_PTR_TABLES) > 64:
            _PTR_TABLES.clear()
        table = torch.tensor(ptrs, dtype=torch.int64).to(partials.device)
        _PTR_TABLES[key] = table
    lib.call("neat_ln_bwd_finalize_batched", _p(partials, f32, "partials"), nblk, cols, count, _p(table), int(accumulate),
             _stream())


# ---- GEMM ----
class PackedWeight:
    """A frozen weight matrix packed for the GEMM kernels (see include/neat_hip.h): ``data`` in MFMA-fragment order
    (neat_gemm_bf16) and, for shapes the tile256 kernel takes, ``data256`` in its LDS image (neat_gemm_bf16_t256).
    ``N`` / ``K`` are the LOGICAL dims of out[M,N] = A[M,K] @ Wl[N,K]^T."""

    __slots__ = ("data", "data256", "N", "K")

    def __init__(self, data, N, K, data256=None):
        self.data, self.N, self.K, self.data256 = data, N, K, data256


def t256_shape_ok(N, K):
    return N % 256 == 0 and K % 128 == 0 and N <= 4096


# NEAT_GEMM=panel | t256 forces one kernel wherever its shape rules allow (A/B timing); default: the library's choice
_GEMM_FORCE = os.environ.get("NEAT_GEMM", "")


def pack_weight(w, transposed=False) -> PackedWeight:
    """w fp32 [N,K] (transposed=False) or fp32 [K,N] whose transpose is the logical matrix (dgrad).  Both kernel
    layouts are kept (N*K bf16 each): which kernel runs is decided per call from M and the epilogue."""
    if w.dim() != 2:
        raise RuntimeError("pack_weight: expected a 2-D weight")
    N, K = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    dst = torch.empty(N * K, dtype=bf16, device=w.device)
    lib.call("neat_pack_weight_bf16", _p(w, f32, "w"), _p(dst), N, K, int(transposed), _stream())
    dst256 = None
    if t256_shape_ok(N, K) and _GEMM_FORCE != "panel":
        dst256 = torch.empty(N * K, dtype=bf16, device=w.device)
        lib.call("neat_pack_weight_bf16_t256", _p(w, f32, "w"), _p(dst256), N, K, int(transposed), _stream())
    return PackedWeight(dst, N, K, dst256)


def gemm(A, W: PackedWeight, epilogue, bias=None, res=None, aux=None, out0=None, out1=None, kernel=None, out0_rows=None):
    """out[M,N] = epilogue(A[M,K] @ Wl[N,K]^T); allocates outputs that are not passed in.
    kernel: None = the library's choice (neat_gemm_prefers_t256), "panel" / "t256" = that kernel (tests, timing).
    out0_rows (EPI_BIAS_GELU only): the pre-activation is needed for the first out0_rows rows only; where the tile256
    kernel runs, out0 comes back as [out0_rows, N] and the other rows are never stored."""
    M, K = A.shape
    N = W.N
    if W.K != K:
        raise RuntimeError(f"gemm: K mismatch {tuple(A.shape)} x packed [{W.N},{W.K}]")
    dev = A.device
    force = kernel or _GEMM_FORCE
    if (epilogue == lib.EPI_BIAS_GELU and out0_rows is not None and 0 <= out0_rows < M and out0 is None
            and W.data256 is not None and force != "panel"
            and (force == "t256" or lib.load().neat_gemm_prefers_t256(M, N, K, epilogue) == 1)):
        if bias is None or bias.dtype != f32 or bias.numel() != N:
            raise RuntimeError("gemm: BIAS_GELU needs a float32 bias of N elements")
        out0 = torch.empty(out0_rows, N, dtype=bf16, device=dev)
        if out1 is None:
            out1 = torch.empty(M, N, dtype=bf16, device=dev)
        args = (_p(A, bf16, "A"), _p(W.data256, bf16, "W"), _p(bias), _p(out0), out0_rows, _p(out1), M, N, K, _stream())
        t = KERNEL_TIMER
        if t is not None and t.match("gemm", epilogue, M, N, K):
            with t.region(M, N, K, saved_rows=out0_rows):
                lib.call("neat_gemm_bf16_t256_gelu_rows", *args)
        else:
            lib.call("neat_gemm_bf16_t256_gelu_rows", *args)
        return out0, out1
    if out0 is None:
        out0 = torch.empty(M, N, dtype=f32 if epilogue in (lib.EPI_BIAS_RES_F32, lib.EPI_F32) else bf16, device=dev)
    if epilogue == lib.EPI_BIAS_GELU and out1 is None:
        out1 = torch.empty(M, N, dtype=bf16, device=dev)
    for t, nm in ((bias, "bias"), (res, "res")):
        if t is not None and t.dtype != f32:
            raise RuntimeError(f"gemm: {nm} must be float32")
    if bias is not None and bias.numel() != N:
        raise RuntimeError("gemm: bias size mismatch")
    if res is not None and tuple(res.shape) != (M, N):
        raise RuntimeError("gemm: residual shape mismatch")
    if aux is not None and (aux.dtype != bf16 or aux.shape[1] != N or aux.shape[0] < M):
        raise RuntimeError("gemm: aux shape/dtype mismatch")
    if force == "t256" and kernel and W.data256 is None:
        raise RuntimeError(f"gemm: no tile256 layout for packed [{W.N},{W.K}] (need N % 256 == 0, K % 128 == 0)")
    use256 = W.data256 is not None and force != "panel" and (
        force == "t256" or lib.load().neat_gemm_prefers_t256(M, N, K, epilogue) == 1)
    fn = "neat_gemm_bf16_t256" if use256 else "neat_gemm_bf16"
    args = (_p(A, bf16, "A"), _p(W.data256 if use256 else W.data, bf16, "W"), _p(bias), _p(res), _p(aux), _p(out0),
            _p(out1), M, N, K, epilogue, _stream())
    t = KERNEL_TIMER
    if t is not None and t.match("gemm", epilogue, M, N, K):
        with t.region(M, N, K):
            lib.call(fn, *args)
    else:
        lib.call(fn, *args)
    return out0, out1


def gemm_res_ln_ok(N, K):
    return N == 512 and K % 128 == 0 and K >= 256


def gemm_res_ln(A, W: PackedWeight, bias, res, gamma, beta, eps=1e-5):
    """x_out = A @ Wl^T + bias + res (fp32) and y = LayerNorm(x_out) * gamma + beta (bf16) in ONE launch
    (neat_gemm_bf16_res_ln) -> (x_out, y, mean, rstd)."""
    M, K = A.shape
    N = W.N
    if W.K != K or not gemm_res_ln_ok(N, K):
        raise RuntimeError(f"gemm_res_ln: unsupported shape {tuple(A.shape)} x packed [{W.N},{W.K}]")
    if tuple(res.shape) != (M, N) or res.dtype != f32:
        raise RuntimeError("gemm_res_ln: residual must be float32 [M, N]")
    dev = A.device
    x_out = torch.empty(M, N, dtype=f32, device=dev)
    y = torch.empty(M, N, dtype=bf16, device=dev)
    mean = torch.empty(M, dtype=f32, device=dev)
    rstd = torch.empty(M, dtype=f32, device=dev)
    lib.call("neat_gemm_bf16_res_ln", _p(A, bf16, "A"), _p(W.data, bf16, "W"), _p(bias, f32, "bias"), _p(res, f32, "res"),
             _p(gamma, f32, "gamma"), _p(beta, f32, "beta"), _p(x_out), _p(y), _p(mean), _p(rstd), M, N, K, float(eps),
             _stream())
    return x_out, y, mean, rstd


# ---- image preprocessing ----
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def image_preprocess(img_u8, size=224, mean=CLIP_MEAN, std=CLIP_STD, out=None):
    """img_u8 uint8 [H, W, 3] on the GPU -> fp32 [3, size, size]: Pillow-exact antialiased bicubic resize, / 255,
    Normalize (reference src/data/__init__.py:14-20)."""
    if img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise RuntimeError(f"image_preprocess: expected uint8 [H, W, 3], got {tuple(img_u8.shape)}")
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    if out is None:
        out = torch.empty(3, size, size, dtype=f32, device=img_u8.device)
    elif tuple(out.shape) != (3, size, size):
        raise RuntimeError("image_preprocess: out must be fp32 [3, size, size]")
    nbytes = lib.load().neat_image_preprocess_ws_bytes(H, W, size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=img_u8.device)
    lib.call("neat_image_preprocess", _p(img_u8, torch.uint8, "img"), H, W, _p(out, f32, "out"), size,
             float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]), _p(ws), _stream())
    return out


def image_preprocess_batch(images, size=224, mean=CLIP_MEAN, std=CLIP_STD, out=None):
    """images: list of uint8 [H_i, W_i, 3] GPU tensors (mixed sizes) -> fp32 [n, 3, size, size] in ONE launch set;
    bit-identical to ``image_preprocess`` per image."""
    n = len(images)
    if n == 0:
        raise RuntimeError("image_preprocess_batch: empty batch")
    dev = images[0].device
    for im in images:
        if im.dim() != 3 or im.shape[2] != 3 or im.dtype != torch.uint8 or im.device != dev or not im.is_contiguous():
            raise RuntimeError("image_preprocess_batch: expected contiguous uint8 [H, W, 3] tensors on one GPU")
    hw = [(int(im.shape[0]), int(im.shape[1])) for im in images]
    max_h, max_w = max(h for h, _ in hw), max(w for _, w in hw)
    if out is None:
        out = torch.empty(n, 3, size, size, dtype=f32, device=dev)
    elif tuple(out.shape) != (n, 3, size, size):
        raise RuntimeError("image_preprocess_batch: out must be fp32 [n, 3, size, size]")
    table = torch.tensor([im.data_ptr() for im in images] + [v for p in hw for v in p], dtype=torch.int64).to(dev)
    ptrs, hw_d = table[:n], table[n:].to(torch.int32)
    stride = lib.load().neat_image_preprocess_ws_bytes(max(max_h, size), max(max_w, size), size)
    ws = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    lib.call("neat_image_preprocess_batch", _p(ptrs), _p(hw_d), n, max_h, max_w, _p(out, f32, "out"), size,
             float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]), _p(ws), _stream())
    return out


def frames_preprocess(frames_u8, size=224, mean=CLIP_MEAN, std=CLIP_STD, out=None):
    """frames_u8 uint8 [T, H, W, 3] on the GPU (decoder layout) -> fp32 [T, 3, size, size]: / 255, antialiased bicubic
    resize as torch does it on float tensors, Normalize (reference src/data/__init__.py:22-26)."""
    if frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise RuntimeError(f"frames_preprocess: expected uint8 [T, H, W, 3], got {tuple(frames_u8.shape)}")
    T, H, W = (int(v) for v in frames_u8.shape[:3])
    if out is None:
        out = torch.empty(T, 3, size, size, dtype=f32, device=frames_u8.device)
    elif tuple(out.shape) != (T, 3, size, size):
        raise RuntimeError("frames_preprocess: out must be fp32 [T, 3, size, size]")
    nbytes = lib.load().neat_frames_preprocess_ws_bytes(T, H, W, size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=frames_u8.device)
    lib.call("neat_frames_preprocess", _p(frames_u8, torch.uint8, "frames"), T, H, W, _p(out, f32, "out"), size,
             float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]), _p(ws), _stream())
    return out


# ---- attention ----
def attn_fwd(qkv, nseq, L, heads, causal, save_lse=True, cu=None, head_major=False):
    """cu (int32 [nseq+1], packed batch) makes L the maximum sequence length; rows = qkv.shape[0].
    head_major: qkv columns are [head][q|k|v][64] (``head_major_perm``) instead of [q|k|v][head][64]."""
    C = heads * 64
    rows = qkv.shape[0]
    if qkv.shape[1] != 3 * C or (cu is None and rows != nseq * L):
        raise RuntimeError(f"attn_fwd: qkv shape {tuple(qkv.shape)} does not match nseq={nseq}, L={L}, C={C}")
    out = torch.empty(rows, C, dtype=bf16, device=qkv.device)
    lse = torch.empty(nseq, heads, L, dtype=f32, device=qkv.device) if save_lse else None
    lib.call("neat_attn_fwd", _p(qkv, bf16, "qkv"), _p(out), _p(lse), _p(cu, i32, "cu"), nseq, L, heads,
             int(bool(causal)) | (2 if head_major else 0), _stream())
    return out, lse


def attn_bwd(qkv, dout, lse, nseq, L, heads, causal, cu=None, head_major=False):
    C = heads * 64
    rows = dout.shape[0]
    if qkv.shape[0] < rows or qkv.shape[1] != 3 * C or dout.shape[1] != C or (cu is None and rows != nseq * L):
        raise RuntimeError("attn_bwd: shape mismatch")
    dqkv = torch.empty(rows, 3 * C, dtype=bf16, device=qkv.device)
    lib.call("neat_attn_bwd", _p(qkv, bf16, "qkv"), _p(dout, bf16, "dout"), _p(lse, f32, "lse"), _p(dqkv),
             _p(cu, i32, "cu"), nseq, L, heads, int(bool(causal)) | (2 if head_major else 0), _stream())
    return dqkv


def head_major_perm(C: int, device=None):
    """Row permutation of an nn.MultiheadAttention in_proj weight / bias ([q | k | v] x [head] x 64 rows) that makes the
    projection's output columns head-major: [head][q | k | v][64]."""
    heads = C // 64
    j = torch.arange(64, device=device)
    return torch.cat([part * C + h * 64 + j for h in range(heads) for part in range(3)])


# ---- embedding / pooled head ----
def embed_fwd(tokens, tok_emb, pos_emb):
    nseq, L = tokens.shape
    vocab, C = tok_emb.shape
    if pos_emb.shape[0] < L or pos_emb.shape[1] != C:
        raise RuntimeError("embed_fwd: positional embedding too short")
    x = torch.empty(nseq * L, C, dtype=f32, device=tokens.device)
    lib.call("neat_embed_fwd", _p(tokens, i64, "tokens"), _p(tok_emb, f32, "tok_emb"), _p(pos_emb, f32, "pos_emb"),
             _p(x), nseq, L, C, vocab, _stream())
    return x


def pack_plan(tokens):
    """tokens int64 [nseq, L] -> (eot int32 [nseq], cu_seqlens int32 [nseq+1]) for the packed layout."""
    nseq, L = tokens.shape
    eot = torch.empty(nseq, dtype=i32, device=tokens.device)
    cu = torch.empty(nseq + 1, dtype=i32, device=tokens.device)
    lib.call("neat_pack_plan", _p(tokens, i64, "tokens"), _p(eot), _p(cu), nseq, L, _stream())
    return eot, cu


def embed_packed_fwd(tokens, cu, rows, tok_emb, pos_emb):
    nseq, L = tokens.shape
    vocab, C = tok_emb.shape
    if pos_emb.shape[0] < L or pos_emb.shape[1] != C:
        raise RuntimeError("embed_packed_fwd: positional embedding too short")
    x = torch.empty(rows, C, dtype=f32, device=tokens.device)
    lib.call("neat_embed_packed_fwd", _p(tokens, i64, "tokens"), _p(cu, i32, "cu"), _p(tok_emb, f32, "tok_emb"),
             _p(pos_emb, f32, "pos_emb"), _p(x), nseq, L, C, vocab, _stream())
    return x


def eot_index(tokens):
    nseq, L = tokens.shape
    eot = torch.empty(nseq, dtype=i32, device=tokens.device)
    lib.call("neat_eot_index", _p(tokens, i64, "tokens"), _p(eot), nseq, L, _stream())
    return eot


def pool_proj_fwd(x, row_idx, gamma, beta, proj_t, nseq, L, normalize, save=True, eps=1e-5):
    """proj_t: [D, C] = transpose of the projection parameter (frozen; transposed once by the caller)."""
    D, C = proj_t.shape
    if x.shape[0] < nseq * L or x.shape[1] != C or (L == 0 and row_idx is None):
        raise RuntimeError("pool_proj_fwd: shape mismatch")
    dev = x.device
    feat = torch.empty(nseq, D, dtype=f32, device=dev)
    xe = torch.empty(nseq, C, dtype=f32, device=dev) if save else None
    mean = torch.empty(nseq, dtype=f32, device=dev) if save else None
    rstd = torch.empty(nseq, dtype=f32, device=dev) if save else None
    nrm = torch.empty(nseq, dtype=f32, device=dev) if save else None
    ws = torch.empty(nseq * (C + D), dtype=f32, device=dev)
    lib.call("neat_pool_proj_fwd", _p(x, f32, "x"), _p(row_idx, i32), _p(gamma, f32), _p(beta, f32), _p(proj_t, f32),
             _p(feat), _p(xe), _p(mean), _p(rstd), _p(nrm), _p(ws), nseq, L, C, D, int(normalize), eps, _stream())
    return feat, (xe, mean, rstd, nrm)


def pool_proj_bwd(dfeat, feat, saved, gamma, proj, row_idx, nseq, L, normalize, dxf, dxb):
    xe, mean, rstd, nrm = saved
    C, D = proj.shape
    ws = torch.empty(nseq * (C + D), dtype=f32, device=dfeat.device)
    lib.call("neat_pool_proj_bwd", _p(dfeat, f32, "dfeat"), _p(feat, f32), _p(nrm, f32), _p(xe, f32), _p(mean, f32),
             _p(rstd, f32), _p(gamma, f32), _p(proj, f32), _p(row_idx, i32), _p(dxf, f32), _p(dxb, bf16), _p(ws), nseq, L,
             C, D, int(normalize), _stream())


# ---- vision stem ----
def patchify(images, patch, Kpad):
    n, c, H, W = images.shape
    if c != 3:
        raise RuntimeError("patchify: expected 3 channels")
    out = torch.empty(n * (H // patch) * (W // patch), Kpad, dtype=bf16, device=images.device)
    lib.call("neat_patchify", _p(images, f32, "images"), _p(out), n, H, W, patch, Kpad, _stream())
    return out


def vision_embed(patch_emb, cls, pos, gamma, beta, n, P, eps=1e-5):
    C = patch_emb.shape[1]
    x = torch.empty(n * (P + 1), C, dtype=f32, device=patch_emb.device)
    lib.call("neat_vision_embed", _p(patch_emb, f32), _p(cls, f32), _p(pos, f32), _p(gamma, f32), _p(beta, f32), _p(x),
             n, P, C, eps, _stream())
    return x


def frame_mean_norm(feat, n, F):
    D = feat.shape[1]
    out = torch.empty(n, D, dtype=f32, device=feat.device)
    lib.call("neat_frame_mean_norm", _p(feat, f32, "feat"), _p(out), n, F, D, _stream())
    return out


# ---- coarse alignment / loss / optimizer ----
def coarse_align(q, pos, neg, G):
    B, D = q.shape
    N = G.shape[0]
    dev = q.device
    S = torch.empty(B, N, dtype=f32, device=dev)
    idx = torch.empty(B, dtype=i32, device=dev)
    coarse = torch.empty(B, D, dtype=f32, device=dev)
    ws = torch.empty(lib.load().neat_coarse_ws_floats(B, N), dtype=f32, device=dev)
    lib.call("neat_coarse_align", _p(q, f32, "q"), _p(pos, f32, "pos"), _p(neg, f32, "neg"), _p(G, f32, "G"), _p(S),
             _p(idx), _p(coarse), _p(ws), B, N, D, _stream())
    return coarse, idx, S


def coarse_align_i2t(q, G, Gp, Gn):
    """neat_coarse_align_i2t (reference train_TTA.py:20-28): q f32 [B,D]; G, Gp, Gn f32 [N,D] ->
    (G[idx] f32 [B,D], idx int32 [B]) with idx = argmax_j (q Gp^T)(1 - sigmoid(q Gn^T))."""
    B, D = q.shape
    N = G.shape[0]
    if tuple(Gp.shape) != (N, D) or tuple(Gn.shape) != (N, D) or G.shape[1] != D:
        raise RuntimeError("coarse_align_i2t: G, Gp, Gn must all be [N, D] with q's D")
    dev = q.device
    idx = torch.empty(B, dtype=i32, device=dev)
    coarse = torch.empty(B, D, dtype=f32, device=dev)
    ws = torch.empty(lib.load().neat_coarse_ws_floats(B, N), dtype=f32, device=dev)
    lib.call("neat_coarse_align_i2t", _p(q, f32, "q"), _p(G, f32, "G"), _p(Gp, f32, "Gp"), _p(Gn, f32, "Gn"), _p(idx),
             _p(coarse), _p(ws), B, N, D, _stream())
    return coarse, idx


def coarse_align_static(q, pos, neg, G, S, idx, coarse):
    """neat_coarse_align_static: the caller passes every output; the call allocates nothing (the per-tile partials live
    in the library's static device array; B * ceil(N/64) <= neat_coarse_static_max_slots())."""
    B, D = q.shape
    N = G.shape[0]
    if tuple(S.shape) != (B, N) or tuple(idx.shape) != (B,) or tuple(coarse.shape) != (B, D):
        raise RuntimeError("coarse_align_static: output shapes must be S [B,N], idx [B], coarse [B,D]")
    lib.call("neat_coarse_align_static", _p(q, f32, "q"), _p(pos, f32, "pos"), _p(neg, f32, "neg"), _p(G, f32, "G"),
             _p(S, f32, "S"), _p(idx, i32, "idx"), _p(coarse, f32, "coarse"), B, N, D, _stream())
    return coarse, idx, S


def tta_loss_fwd_bwd(text, pos, inv, text_all, coarse_all, T1, T2, row0=0):
    Bl, D = text.shape
    B = text_all.shape[0]
    dev = text.device
    losses = torch.empty(3, dtype=f32, device=dev)
    dt = torch.empty(Bl, D, dtype=f32, device=dev)
    dp = torch.empty(Bl, D, dtype=f32, device=dev)
    dv = torch.empty(Bl, D, dtype=f32, device=dev)
    ws = torch.empty(lib.load().neat_loss_ws_floats(B, Bl), dtype=f32, device=dev)
    lib.call("neat_tta_loss_fwd_bwd", _p(text, f32, "text"), _p(pos, f32, "pos"), _p(inv, f32, "inv"),
             _p(text_all, f32, "text_all"), _p(coarse_all, f32, "coarse_all"), float(T1), float(T2), _p(losses), _p(dt),
             _p(dp), _p(dv), _p(ws), B, Bl, row0, D, _stream())
    return losses, dt, dp, dv


def adamw_flat(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0):
    lib.call("neat_adamw_flat", _p(p, f32, "p"), _p(g, f32, "g"), _p(m, f32, "m"), _p(v, f32, "v"), p.numel(), float(lr),
             float(beta1), float(beta2), float(eps), float(wd), int(step), _stream())


# ---- retrieval ----
def _strides2d(t, name):
    if t.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D tensor")
    sr, sc = t.stride()
    if t.shape[1] == 1:
        sc = 1
    if t.shape[0] == 1 and sc == 1:
        sr = t.shape[1]
    if sc != 1 and sr != 1:
        raise RuntimeError(f"{name}: one stride must be 1 (got {t.stride()})")
    return sr, sc


def topk_rows(scores, k) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-wise top-k (values, int32 indices) of a [R, N] fp32 matrix or of a transposed view of one."""
    if not scores.is_cuda or scores.dtype != f32:
        raise RuntimeError("topk_rows: expected a float32 GPU tensor")
    R, N = scores.shape
    sr, sc = _strides2d(scores, "scores")
    idx = torch.empty(R, k, dtype=i32, device=scores.device)
    val = torch.empty(R, k, dtype=f32, device=scores.device)
    ws = torch.empty(lib.load().neat_topk_ws_bytes(R, N, k), dtype=torch.uint8, device=scores.device)
    lib.call("neat_topk_rows", scores.data_ptr(), sr, sc, R, N, k, _p(idx), _p(val), _p(ws), _stream())
    return val, idx


def sim_topk(Q, G, k) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k rows of G for every row of Q by exact-fp32 dot product (values, int32 indices), without the [Nq, Ng] score
    matrix in HBM (reference evaluation/retrieval.py:70 + :294-300).  Bit-identical to gemm_f32_nt + topk_rows."""
    Nq, D = Q.shape
    Ng = G.shape[0]
    if G.shape[1] != D:
        raise RuntimeError(f"sim_topk: feature sizes differ ({tuple(Q.shape)} vs {tuple(G.shape)})")
    idx = torch.empty(Nq, k, dtype=i32, device=Q.device)
    val = torch.empty(Nq, k, dtype=f32, device=Q.device)
    ws = torch.empty(lib.load().neat_sim_topk_ws_bytes(Nq, Ng, k), dtype=torch.uint8, device=Q.device)
    lib.call("neat_sim_topk", _p(Q, f32, "Q"), _p(G, f32, "G"), k, _p(idx), _p(val), Nq, Ng, D, _p(ws), _stream())
    return val, idx


def recall_from_topk(idx, positive_pairs):
    if positive_pairs.dtype == torch.bool:
        positive_pairs = positive_pairs.view(torch.uint8)
    if positive_pairs.dtype != torch.uint8 or not positive_pairs.is_cuda:
        raise RuntimeError("recall_from_topk: positives must be a bool/uint8 GPU tensor")
    R, N = positive_pairs.shape
    sr, sc = _strides2d(positive_pairs, "positive_pairs")
    out = torch.empty(R, dtype=f32, device=idx.device)
    lib.call("neat_recall_from_topk", _p(idx, i32, "idx"), positive_pairs.data_ptr(), sr, sc, R, N, idx.shape[1], _p(out),
             _stream())
    return out


def best_positive_rank(scores, positive_pairs):
    """int32 [R]: 1-based rank of the best-ranked positive per row (0: no positive).  Accepts transposed views."""
    if positive_pairs.dtype == torch.bool:
        positive_pairs = positive_pairs.view(torch.uint8)
    if not scores.is_cuda or scores.dtype != f32 or positive_pairs.dtype != torch.uint8 or not positive_pairs.is_cuda:
        raise RuntimeError("best_positive_rank: expected float32 scores and bool/uint8 positives on the GPU")
    R, N = scores.shape
    sr, sc = scores.stride()
    pr, pc = positive_pairs.stride()
    out = torch.empty(R, dtype=i32, device=scores.device)
    lib.call("neat_best_positive_rank", scores.data_ptr(), sr, sc, positive_pairs.data_ptr(), pr, pc, R, N, _p(out),
             _stream())
    return out


_KS_TABLES = {}


def recall_owner_topk(idx, ks, row_owner=None, col_owner=None, npos=None):
    """Recall for every cut-off of ``ks`` (ascending ints) from ONE top-K index list idx int32 [R, K]; positives as owner
    vectors (int32 GPU tensors): row_owner [R] and/or col_owner [Ncol]; npos int32 [R] positives per row (None: 1).
    Returns fp32 [len(ks), R] -- the values ``recall_at_k`` returns for each k."""
    R, K = idx.shape
    ks = tuple(int(k) for k in ks)
    if list(ks) != sorted(ks) or not ks or ks[0] < 1:
        raise RuntimeError("recall_owner_topk: cut-offs must be ascending positive ints")
    key = (idx.device.index, ks)
    table = _KS_TABLES.get(key)
    if table is None:
        table = torch.tensor(ks, dtype=i32).to(idx.device)
        _KS_TABLES[key] = table
    out = torch.empty(len(ks), R, dtype=f32, device=idx.device)
    lib.call("neat_recall_owner_topk", _p(idx, i32, "idx"), R, K, _p(row_owner, i32, "row_owner"),
             _p(col_owner, i32, "col_owner"), _p(npos, i32, "npos"), _p(table), len(ks), _p(out), _stream())
    return out


def best_positive_rank_owner(scores, row_owner=None, col_owner=None):
    """int32 [R]: 1-based rank of the best-ranked positive per row, positives as ONE owner vector (see recall_owner_topk)."""
    if not scores.is_cuda or scores.dtype != f32:
        raise RuntimeError("best_positive_rank_owner: expected float32 scores on the GPU")
    R, N = scores.shape
    sr, sc = scores.stride()
    out = torch.empty(R, dtype=i32, device=scores.device)
    ws = torch.empty(8 * R, dtype=torch.uint8, device=scores.device)      # transposed views: best positive per logical row
    lib.call("neat_best_positive_rank_owner", scores.data_ptr(), sr, sc, _p(row_owner, i32, "row_owner"),
             _p(col_owner, i32, "col_owner"), R, N, _p(out), _p(ws), _stream())
    return out


def mcq_scores(image_features, text_features):
    """image_features f32 [B,D], text_features f32 [n,B,D] -> (logits [B,n], pred int32 [B])."""
    n, B, D = text_features.shape
    logits = torch.empty(B, n, dtype=f32, device=image_features.device)
    pred = torch.empty(B, dtype=i32, device=image_features.device)
    lib.call("neat_mcq_scores", _p(image_features.contiguous(), f32, "image_features"),
             _p(text_features.contiguous(), f32, "text_features"), _p(logits), _p(pred), B, n, D, _stream())
    return logits, pred


def gemm_f32_nt(A, B):
    M, K = A.shape
    N = B.shape[0]
    Cm = torch.empty(M, N, dtype=f32, device=A.device)
    lib.call("neat_gemm_f32_nt", _p(A, f32, "A"), _p(B, f32, "B"), _p(Cm), M, N, K, _stream())
    return Cm


def cast_bf16(src):
    dst = torch.empty(src.shape, dtype=bf16, device=src.device)
    lib.call("neat_cast_bf16", _p(src, f32, "src"), _p(dst), src.numel(), _stream())
    return dst


def transpose_cast_bf16(src):
    R, Cc = src.shape
    dst = torch.empty(Cc, R, dtype=bf16, device=src.device)
    lib.call("neat_transpose_cast_bf16", _p(src, f32, "src"), _p(dst), R, Cc, _stream())
    return dst
