"""Encoder towers on HIP kernels: frozen weights in kernel layout, forward with saved activations and
the hand-written backward that yields the LayerNorm gamma/beta gradients.

Data layout in HBM (row-major, rows = tokens of all sequences back to back):
  residual stream x        fp32 [M, C]      one buffer per LayerNorm input (saved for backward)
  LN output / attn output  bf16 [M, C]      GEMM A operands (temporaries)
  qkv                      bf16 [M, 3C]     saved per layer (attention backward recomputes P from lse)
  MLP pre-activation h     bf16 [M, 4C]     saved per layer (QuickGELU')
  weights                  bf16, packed once in MFMA-fragment order (W for forward, W^T for the
                           data-gradient GEMMs; weights are frozen: no wgrad)
  LN gamma/beta, biases, statistics, features, loss, optimizer state: fp32.

Reference call sites: open_clip/model.py:265-286 (encode_image / encode_text),
open_clip/transformer.py:188-243 (ResidualAttentionBlock), :502-547 (VisionTransformer.forward).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import lib, ops

f32, bf16 = torch.float32, torch.bfloat16

# rows up to which the LayerNorm after a residual GEMM is folded into that GEMM (Stack.forward, ops.gemm_res_ln).  OFF by
# default (0): measured in round 3, the fused launch wins in isolation at the packed text batches' size but not inside the
# step, and loses at full length (numbers at the call site and in DESIGN.md §4); NEAT_FUSE_LN_MAX_ROWS=<rows> turns it on.
FUSE_LN_MAX_ROWS = int(os.environ.get("NEAT_FUSE_LN_MAX_ROWS", "0"))


class BlockWeights:
    """One ResidualAttentionBlock in kernel layout.  ln_* are the LIVE fp32 parameter tensors."""

    def __init__(self, blk, need_bwd: bool):
        self.ln1_w, self.ln1_b = blk.ln_1.weight, blk.ln_1.bias
        self.ln2_w, self.ln2_b = blk.ln_2.weight, blk.ln_2.bias
        # the qkv projection is packed with its rows in head-major order ([head][q|k|v][64]): the attention kernels then
        # read a head's three pieces as one 384-B run per token (ops.head_major_perm; frozen weights: permuted once)
        perm = ops.head_major_perm(blk.attn.in_proj_weight.shape[1], blk.attn.in_proj_weight.device)
        mats = dict(qkv=blk.attn.in_proj_weight.detach()[perm], o=blk.attn.out_proj.weight, fc=blk.mlp.c_fc.weight,
                    proj=blk.mlp.c_proj.weight)
        # packed once: weights are frozen during TTA (only LN gamma/beta adapt)
        self.w = {k: ops.pack_weight(v.detach().contiguous()) for k, v in mats.items()}
        self.wT = {k: ops.pack_weight(v.detach().contiguous(), transposed=True) for k, v in mats.items()} \
            if need_bwd else None
        self.b = dict(qkv=blk.attn.in_proj_bias.detach()[perm].contiguous(), o=blk.attn.out_proj.bias.detach(),
                      fc=blk.mlp.c_fc.bias.detach(), proj=blk.mlp.c_proj.bias.detach())


class Stack:
    """Transformer stack (open_clip/transformer.py:291-325) on HIP kernels."""

    def __init__(self, transformer, heads: int, causal: bool, need_bwd: bool):
        self.blocks = [BlockWeights(b, need_bwd) for b in transformer.resblocks]
        self.heads, self.causal = heads, causal

    def forward(self, x, nseq: int, L: int, save: Optional[list], cu=None, grad_rows=None):
        """x fp32 [rows, C] -> fp32 [rows, C] (rows = nseq*L, or cu[nseq] for a packed batch with L = longest
        sequence).  With ``save`` a list, per-layer tensors needed by ``backward`` are appended (and x buffers
        are never overwritten).  grad_rows: only the first grad_rows rows will see a backward pass (the MLP
        pre-activation of the other rows is not kept)."""
        # The LayerNorm that follows a residual GEMM (ln_2 after the out-projection, the NEXT block's ln_1 after c_proj) is
        # folded into that GEMM's epilogue (ops.gemm_res_ln, csrc/gemm_rowln.hip) when FUSE_LN_MAX_ROWS allows it (width 512
        # only).  Measured (tools/rowln_bench.py, profiles/r03_rowln_*): in isolation at 14 848 rows the out-projection
        # 46.6 vs 50.7 us and c_proj 79.7 vs 90.6 us for GEMM + LayerNorm; inside the packed step the fused launches average
        # 68.7 us against 56.7 + 9.3 us (the 64-row unit re-reads the weights twice as often, and in the step they are not
        # L2-resident): 16.7 vs 15.9 ms per batch; at 78 848 rows 173.5 vs 144.5 us and 296 vs 251 us.  So: off.
        rows, C = x.shape
        fuse = FUSE_LN_MAX_ROWS > 0 and rows <= FUSE_LN_MAX_ROWS and ops.gemm_res_ln_ok(C, C)
        stats = save is not None
        w0 = self.blocks[0]
        y1, _, mu1, rs1 = ops.ln_fwd(x, w0.ln1_w, w0.ln1_b, save_stats=stats)
        for k, w in enumerate(self.blocks):
            nxt = self.blocks[k + 1] if k + 1 < len(self.blocks) else None
            qkv, _ = ops.gemm(y1, w.w["qkv"], lib.EPI_BIAS_BF16, bias=w.b["qkv"])
            o, lse = ops.attn_fwd(qkv, nseq, L, self.heads, self.causal, save_lse=stats, cu=cu, head_major=True)
            if fuse:
                x1, y2, mu2, rs2 = ops.gemm_res_ln(o, w.w["o"], w.b["o"], x, w.ln2_w, w.ln2_b)
            else:
                x1, _ = ops.gemm(o, w.w["o"], lib.EPI_BIAS_RES_F32, bias=w.b["o"], res=x)
                y2, _, mu2, rs2 = ops.ln_fwd(x1, w.ln2_w, w.ln2_b, save_stats=stats)
            h, u = ops.gemm(y2, w.w["fc"], lib.EPI_BIAS_GELU, bias=w.b["fc"],
                            out0_rows=grad_rows if save is not None else 0)
            if fuse and nxt is not None and ops.gemm_res_ln_ok(C, u.shape[1]):
                x2, y1n, mu1n, rs1n = ops.gemm_res_ln(u, w.w["proj"], w.b["proj"], x1, nxt.ln1_w, nxt.ln1_b)
            else:
                x2, _ = ops.gemm(u, w.w["proj"], lib.EPI_BIAS_RES_F32, bias=w.b["proj"], res=x1)
                if nxt is not None:
                    y1n, _, mu1n, rs1n = ops.ln_fwd(x2, nxt.ln1_w, nxt.ln1_b, save_stats=stats)
            if save is not None:
                save.append(dict(x0=x, mu1=mu1, rs1=rs1, qkv=qkv, lse=lse, x1=x1, mu2=mu2, rs2=rs2, h=h))
            x = x2
            if nxt is not None:
                y1, mu1, rs1 = y1n, mu1n, rs1n
        return x

    def backward(self, dxf, dxb, nseq: int, L: int, saved: list, grads: List[Dict[str, torch.Tensor]],
                 accumulate: bool, cu=None, need_input_grad: bool = True):
        """dxf fp32 / dxb bf16 [M, C]: gradient w.r.t. the stack output for the FIRST nseq sequences of the saved
        forward (M = nseq*L rows, or cu[nseq] rows of a packed batch).  grads[i] = dict(ln1_w, ln1_b, ln2_w,
        ln2_b) receive d(gamma), d(beta).  need_input_grad=False: nothing below the stack is trainable (the frozen
        embeddings of the text tower), so the first block's ln_1 only reduces its parameter gradients; the gradient with
        respect to the stack input is neither formed nor written and (None, None) is returned."""
        M = dxf.shape[0]
        C = dxf.shape[1]
        # the per-block partial sums of d(gamma), d(beta) of all 2*layers LayerNorms are reduced by ONE launch at the end
        nl = len(self.blocks)
        parts = torch.empty(2 * nl, ops.ln_bwd_blocks(M), 2, C, dtype=f32, device=dxf.device)
        dsts = []
        for j, (w, s, g) in enumerate(zip(reversed(self.blocks), reversed(saved), reversed(grads))):
            dh, _ = ops.gemm(dxb, w.wT["proj"], lib.EPI_GELU_BWD, aux=s["h"])
            dy2, _ = ops.gemm(dh, w.wT["fc"], lib.EPI_BF16)
            dxf, dxb = ops.ln_bwd(dy2, s["x1"][:M], w.ln2_w, s["mu2"], s["rs2"], dres=dxf, inplace_res=True,
                                  partial=parts[2 * j])
            dsts.append((g["ln2_w"], g["ln2_b"]))
            do, _ = ops.gemm(dxb, w.wT["o"], lib.EPI_BF16)
            dqkv = ops.attn_bwd(s["qkv"], do, s["lse"], nseq, L, self.heads, self.causal, cu=cu, head_major=True)
            dy1, _ = ops.gemm(dqkv, w.wT["qkv"], lib.EPI_BF16)
            last = j == nl - 1 and not need_input_grad
            dxf, dxb = ops.ln_bwd(dy1, s["x0"][:M], w.ln1_w, s["mu1"], s["rs1"], dres=None if last else dxf,
                                  inplace_res=not last, want_f32=not last, want_bf16=not last, partial=parts[2 * j + 1])
            dsts.append((g["ln1_w"], g["ln1_b"]))
        ops.ln_bwd_finalize_batched(parts, dsts, accumulate=accumulate)
        return dxf, dxb


class TextTower:
    """CLIP.encode_text (open_clip/model.py:269-286)."""

    def __init__(self, model, need_bwd: bool = True):
        self.model = model
        C = model.positional_embedding.shape[1]
        self.C, self.D = C, model.text_projection.shape[1]
        self.heads = C // 64
        self.stack = Stack(model.transformer, self.heads, True, need_bwd)
        self.tok = model.token_embedding.weight.detach()
        self.pos = model.positional_embedding.detach()
        self.lnf_w, self.lnf_b = model.ln_final.weight.detach(), model.ln_final.bias.detach()
        self.proj = model.text_projection.detach().contiguous()
        self.proj_t = self.proj.t().contiguous()          # frozen: transposed once for the forward tile GEMM

    def plan(self, tokens):
        """The packed layout of a token batch: (tokens, cu int32 [nseq+1] on the device, cu on the host).  It depends on
        the tokens only, so a caller that encodes the same batch several times (the tta_steps of one adaptation batch)
        computes it -- and pays its device-to-host copy -- once and passes it to ``forward(..., plan=)``."""
        tokens = tokens.contiguous()
        _, cu = ops.pack_plan(tokens)
        return tokens, cu, cu.cpu()

    def forward(self, tokens, normalize: bool, save: bool, packed: bool = False, n_grad=None, plan=None):
        """tokens int64 [nseq, L] -> (feat fp32 [nseq, D], ctx).  n_grad: only the first n_grad sequences will be
        passed to ``backward`` (default: all).  plan: ``self.plan(tokens)`` of these very tokens (packed runs).

        ``packed``: run only positions 0..EOT of every sequence (rows back to back).  With the causal mask and
        EOT pooling the positions after EOT cannot reach the pooled feature, nor receive gradient from it, so
        features, losses and LN gradients are those of the full-length run (up to fp32 summation order)."""
        tokens = tokens.contiguous()
        nseq, L = tokens.shape
        saved = [] if save else None
        if packed:
            if plan is not None:
                tokens, cu, cu_host = plan
                if tuple(tokens.shape) != (nseq, L):
                    raise RuntimeError("TextTower.forward: the plan belongs to another token batch")
            else:
                _, cu = ops.pack_plan(tokens)
                cu_host = cu.cpu()                                # one small D2H copy: row counts size the buffers
            rows, Lmax = int(cu_host[-1]), int((cu_host[1:] - cu_host[:-1]).max())
            x = ops.embed_packed_fwd(tokens, cu, rows, self.tok, self.pos)
            x = self.stack.forward(x, nseq, Lmax, saved, cu=cu,
                                   grad_rows=None if n_grad is None else int(cu_host[n_grad]))
            last = (cu[1:] - 1).contiguous()                      # absolute row of each EOT token
            feat, head = ops.pool_proj_fwd(x, last, self.lnf_w, self.lnf_b, self.proj_t, nseq, 0, normalize, save=save)
            ctx = dict(saved=saved, eot=last, head=head, feat=feat, nseq=nseq, L=Lmax, normalize=normalize, cu=cu,
                       cu_host=cu_host) if save else None
            return feat, ctx
        x = ops.embed_fwd(tokens, self.tok, self.pos)
        x = self.stack.forward(x, nseq, L, saved, grad_rows=None if n_grad is None else n_grad * L)
        eot = ops.eot_index(tokens)
        feat, head = ops.pool_proj_fwd(x, eot, self.lnf_w, self.lnf_b, self.proj_t, nseq, L, normalize, save=save)
        ctx = dict(saved=saved, eot=eot, head=head, feat=feat, nseq=nseq, L=L, normalize=normalize, cu=None) if save else None
        return feat, ctx

    def backward(self, ctx, dfeat, grads, accumulate: bool):
        """dfeat fp32 [n_grad, D] for the first n_grad sequences of the forward."""
        n, L, cu = dfeat.shape[0], ctx["L"], ctx["cu"]
        rows = n * L if cu is None else int(ctx["cu_host"][n])
        dxf = torch.zeros(rows, self.C, dtype=f32, device=dfeat.device)
        dxb = torch.zeros(rows, self.C, dtype=bf16, device=dfeat.device)
        ops.pool_proj_bwd(dfeat.contiguous(), ctx["feat"], ctx["head"], self.lnf_w, self.proj, ctx["eot"], n,
                          L if cu is None else 0, ctx["normalize"], dxf, dxb)
        self.stack.backward(dxf, dxb, n, L, ctx["saved"], grads, accumulate, cu=cu, need_input_grad=False)


class VisionTower:
    """CLIP.encode_image -> VisionTransformer.forward (open_clip/transformer.py:502-547), forward only."""

    def __init__(self, model):
        v = model.visual
        self.C = v.conv1.weight.shape[0]
        self.patch = v.conv1.weight.shape[-1]
        self.D = v.proj.shape[1]
        K = 3 * self.patch * self.patch
        self.Kpad = (K + 127) // 128 * 128
        w = v.conv1.weight.detach().reshape(self.C, K)
        if self.Kpad != K:
            w = torch.nn.functional.pad(w, (0, self.Kpad - K))
        self.w_patch = ops.pack_weight(w.contiguous())
        self.cls = v.class_embedding.detach()
        self.pos = v.positional_embedding.detach()
        self.lnpre = (v.ln_pre.weight.detach(), v.ln_pre.bias.detach())
        self.lnpost = (v.ln_post.weight.detach(), v.ln_post.bias.detach())
        self.proj_t = v.proj.detach().t().contiguous()
        self.stack = Stack(v.transformer, self.C // 64, False, need_bwd=False)

    def forward(self, images, normalize: bool):
        images = images.contiguous().float()
        n, _, H, W = images.shape
        P = (H // self.patch) * (W // self.patch)
        if P + 1 != self.pos.shape[0]:
            raise RuntimeError(f"encode_image: image {H}x{W} gives {P + 1} tokens, model expects {self.pos.shape[0]}")
        patches = ops.patchify(images, self.patch, self.Kpad)
        pe, _ = ops.gemm(patches, self.w_patch, lib.EPI_F32)
        x = ops.vision_embed(pe, self.cls, self.pos, self.lnpre[0], self.lnpre[1], n, P)
        x = self.stack.forward(x, n, P + 1, None)
        feat, _ = ops.pool_proj_fwd(x, None, self.lnpost[0], self.lnpost[1], self.proj_t, n, P + 1, normalize, save=False)
        return feat
