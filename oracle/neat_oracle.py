"""CPU fp32 restatement of NEAT's test-time-adaptation hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a plain-PyTorch (CPU, fp32) restatement of what the reference computes
on the path ``train_one_epoch`` -> ``encode_image`` / ``encode_text`` -> loss -> ``backward`` ->
``AdamW.step`` -> ``recall_at_k``.  It is imported only by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never by the product path under ``neat_amd/`` (which
fails loudly if the HIP library is missing).

Everything is written with explicit forward *and* backward formulas (no autograd), because that is
what the HIP kernels implement; the explicit backward is checked against the reference's autograd
through the golden fixtures.

Parity pin: ``tests/golden/make_golden.py`` imports the reference from ``/root/reference`` in the
build container (nothing was denied), feeds it the seeded synthetic weights / inputs
from ``neat_amd.synth`` and stores its outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.  The reference has no tests
or golden vectors of its own.

Each function cites the reference file:line it restates (paths relative to /root/reference/src).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

LN_EPS = 1e-5           # nn.LayerNorm default, open_clip/transformer.py:24-30
NORM_EPS = 1e-12        # F.normalize default, open_clip/model.py:267,286
PDIST_EPS = 1e-6        # nn.PairwiseDistance default, train_TTA.py:87
HEAD_DIM = 64           # CLIPVisionCfg.head_width / text width//heads, open_clip/model.py:30


# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------
def layer_norm_fwd(x, gamma, beta, eps: float = LN_EPS):
    """open_clip/transformer.py:24-30 (F.layer_norm: biased variance, eps inside the sqrt)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * gamma + beta, mean.squeeze(-1), rstd.squeeze(-1)


def layer_norm_bwd(dy, x, mean, rstd, gamma):
    """Backward of F.layer_norm: dx, dgamma, dbeta (SURVEY.md §8a')."""
    xhat = (x - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    g = dy * gamma
    dx = rstd.unsqueeze(-1) * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))
    red = tuple(range(dy.dim() - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


def quick_gelu(x):
    """open_clip/transformer.py:33-36."""
    return x * torch.sigmoid(1.702 * x)


def quick_gelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1.0 + 1.702 * x * (1.0 - s))


def l2_normalize(x, eps: float = NORM_EPS):
    """F.normalize(dim=-1), open_clip/model.py:267,286."""
    n = x.norm(dim=-1, keepdim=True).clamp_min(eps)
    return x / n, n


def l2_normalize_bwd(dy, y, n):
    return (dy - y * (dy * y).sum(-1, keepdim=True)) / n


def _split_heads(x, heads):  # [B,L,C] -> [B,H,L,hd]
    B, L, C = x.shape
    return x.view(B, L, heads, C // heads).permute(0, 2, 1, 3)


def _merge_heads(x):  # [B,H,L,hd] -> [B,L,C]
    B, H, L, hd = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * hd)


def attention_fwd(qkv, heads: int, causal: bool):
    """nn.MultiheadAttention core (open_clip/transformer.py:202,227-229): softmax(QK^T/sqrt(hd)+M)V,
    M = -inf strictly above the diagonal for text (transformer.py:655-661)."""
    B, L, C3 = qkv.shape
    C = C3 // 3
    q, k, v = (_split_heads(t, heads) for t in qkv.split(C, dim=-1))
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(C // heads))
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    p = torch.softmax(s, dim=-1)
    return _merge_heads(p @ v), p


def attention_bwd(do, qkv, p, heads: int):
    B, L, C3 = qkv.shape
    C = C3 // 3
    scale = 1.0 / math.sqrt(C // heads)
    q, k, v = (_split_heads(t, heads) for t in qkv.split(C, dim=-1))
    do_h = _split_heads(do, heads)
    dv = p.transpose(-1, -2) @ do_h
    dp = do_h @ v.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    dq = (ds @ k) * scale
    dk = (ds.transpose(-1, -2) @ q) * scale
    return torch.cat([_merge_heads(dq), _merge_heads(dk), _merge_heads(dv)], dim=-1)


# --------------------------------------------------------------------------------------------
# transformer stack (open_clip/transformer.py:188-243, 291-325)
# --------------------------------------------------------------------------------------------
def _blk(sd, prefix, i):
    p = f"{prefix}resblocks.{i}."
    return {k: sd[p + k] for k in (
        "ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias",
        "attn.out_proj.weight", "attn.out_proj.bias", "ln_2.weight", "ln_2.bias",
        "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")}


def n_layers(sd, prefix):
    i = 0
    while f"{prefix}resblocks.{i}.ln_1.weight" in sd:
        i += 1
    return i


def transformer_fwd(sd, prefix, x, heads, causal, save: Optional[list] = None):
    """x: [B,L,C] fp32.  ``save`` (if given) receives what the backward needs per layer."""
    for i in range(n_layers(sd, prefix)):
        w = _blk(sd, prefix, i)
        y1, mu1, rs1 = layer_norm_fwd(x, w["ln_1.weight"], w["ln_1.bias"])
        qkv = y1 @ w["attn.in_proj_weight"].t() + w["attn.in_proj_bias"]
        o, p = attention_fwd(qkv, heads, causal)
        x1 = x + (o @ w["attn.out_proj.weight"].t() + w["attn.out_proj.bias"])
        y2, mu2, rs2 = layer_norm_fwd(x1, w["ln_2.weight"], w["ln_2.bias"])
        h = y2 @ w["mlp.c_fc.weight"].t() + w["mlp.c_fc.bias"]
        x2 = x1 + (quick_gelu(h) @ w["mlp.c_proj.weight"].t() + w["mlp.c_proj.bias"])
        if save is not None:
            save.append(dict(x0=x, mu1=mu1, rs1=rs1, qkv=qkv, p=p, x1=x1, mu2=mu2, rs2=rs2, h=h))
        x = x2
    return x


def transformer_bwd(sd, prefix, dx, heads, saved, grads: Dict[str, torch.Tensor]):
    """Data-gradient through the frozen weights plus d(gamma), d(beta) of ln_1 / ln_2, which are the
    only adapted parameters (open_clip/factory.py:395-437).  ``grads`` is accumulated in place."""
    for i in reversed(range(n_layers(sd, prefix))):
        w, s = _blk(sd, prefix, i), saved[i]
        p = f"{prefix}resblocks.{i}."
        du = dx @ w["mlp.c_proj.weight"]
        dh = du * quick_gelu_grad(s["h"])
        dy2 = dh @ w["mlp.c_fc.weight"]
        dx1_ln, dg2, db2 = layer_norm_bwd(dy2, s["x1"], s["mu2"], s["rs2"], w["ln_2.weight"])
        dx1 = dx + dx1_ln
        do = dx1 @ w["attn.out_proj.weight"]
        dqkv = attention_bwd(do, s["qkv"], s["p"], heads)
        dy1 = dqkv @ w["attn.in_proj_weight"]
        dx0_ln, dg1, db1 = layer_norm_bwd(dy1, s["x0"], s["mu1"], s["rs1"], w["ln_1.weight"])
        dx = dx1 + dx0_ln
        for k, g in ((p + "ln_1.weight", dg1), (p + "ln_1.bias", db1),
                     (p + "ln_2.weight", dg2), (p + "ln_2.bias", db2)):
            grads[k] = grads[k] + g if k in grads else g
    return dx


# --------------------------------------------------------------------------------------------
# encoders (open_clip/model.py:265-286, open_clip/transformer.py:502-562)
# --------------------------------------------------------------------------------------------
def text_heads(sd):
    return sd["positional_embedding"].shape[1] // HEAD_DIM


def encode_text_fwd(sd, tokens, normalize: bool = True, save: bool = False):
    """CLIP.encode_text (open_clip/model.py:269-286).  tokens: [B,L] int64."""
    B, L = tokens.shape
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][:L]
    saved: Optional[list] = [] if save else None
    x = transformer_fwd(sd, "transformer.", x, text_heads(sd), True, saved)
    eot = tokens.argmax(dim=-1)                      # transformer.py:555-558
    xe = x[torch.arange(B), eot]                     # LN is row-wise: pooling first is equivalent
    z, mu, rs = layer_norm_fwd(xe, sd["ln_final.weight"], sd["ln_final.bias"])
    f = z @ sd["text_projection"]
    if normalize:
        out, nrm = l2_normalize(f)
    else:
        out, nrm = f, None
    cache = dict(saved=saved, eot=eot, xe=xe, mu=mu, rs=rs, nrm=nrm, out=out, shape=(B, L)) if save else None
    return out, cache


def encode_text_bwd(sd, cache, dout, grads: Dict[str, torch.Tensor]):
    """Backward of encode_text w.r.t. the adapted LN parameters (ln_final is NOT adapted: it lives
    outside model.transformer, open_clip/factory.py:405-407, model.py:245)."""
    B, L = cache["shape"]
    df = l2_normalize_bwd(dout, cache["out"], cache["nrm"]) if cache["nrm"] is not None else dout
    dz = df @ sd["text_projection"].t()
    dxe, _, _ = layer_norm_bwd(dz, cache["xe"], cache["mu"], cache["rs"], sd["ln_final.weight"])
    dx = torch.zeros(B, L, dxe.shape[-1])
    dx[torch.arange(B), cache["eot"]] = dxe
    transformer_bwd(sd, "transformer.", dx, text_heads(sd), cache["saved"], grads)
    return grads


def patchify(images, patch: int):
    """[B,3,H,W] -> [B, gh*gw, 3*p*p] in (c, kh, kw) order = conv1 weight flattening
    (stride = kernel conv, open_clip/transformer.py:362,503-505)."""
    B, C, H, W = images.shape
    gh, gw = H // patch, W // patch
    x = images.view(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def encode_image_fwd(sd, images, normalize: bool = True):
    """CLIP.encode_image -> VisionTransformer.forward (open_clip/transformer.py:502-547)."""
    w = sd["visual.conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    x = patchify(images, patch) @ w.reshape(width, -1).t()
    cls = sd["visual.class_embedding"].view(1, 1, -1).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x, _, _ = layer_norm_fwd(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = transformer_fwd(sd, "visual.transformer.", x, width // HEAD_DIM, False)
    z, _, _ = layer_norm_fwd(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    f = z @ sd["visual.proj"]
    return l2_normalize(f)[0] if normalize else f


def encode_video_fwd(sd, clips):
    """VideoCLIP.encode_video (utils/video_utils/model.py:41-91): per-frame encode + L2-norm,
    mean over frames, re-normalise.  clips: [N,F,3,H,W]."""
    N, F = clips.shape[:2]
    f = encode_image_fwd(sd, clips.reshape(N * F, *clips.shape[2:]), normalize=True)
    return l2_normalize(f.view(N, F, -1).mean(1))[0]


# --------------------------------------------------------------------------------------------
# coarse alignment + loss (train_TTA.py:31-91, 137-143)
# --------------------------------------------------------------------------------------------
def coarse_align_t2i(t, p, n, G):
    """train_TTA.py:31-42.  Returns (coarse [B,D], idx [B], S [B,N]); argmax = first max."""
    S = t @ G.t()
    A = (p @ G.t()) * (1.0 - torch.relu(n @ G.t()))
    idx = A.argmax(dim=1)
    return G[idx], idx, S


def coarse_align_i2t(q, G, Gp, Gn):
    """train_TTA.py:20-28 (coarse_aligning_i2t; the reference defines it and never calls it).  q [B,D] image queries;
    G, Gp, Gn [N,D] caption / positive-caption / negated-caption galleries.  Returns (G[idx] [B,D], idx [B])."""
    A = (q @ Gp.t()) * (1.0 - torch.sigmoid(q @ Gn.t()))
    idx = A.argmax(dim=1)
    return G[idx], idx


def tta_loss_fwd_bwd(t, p, v, c, T1: float, T2: float):
    """loss_entropy + loss_sr + loss_tri and its gradient w.r.t. t, p, v (c carries no grad).

    train_TTA.py:138-143 with softmax_entropy :45-48, semantics_reversion_loss :53-65,
    textual_triplet_loss :77-91.  Formulas: SURVEY.md §8(a')."""
    B = t.shape[0]
    # entropy over columns of outputs = rows of Z = c t^T / T1
    Z = (c @ t.t()) / T1
    logp = torch.log_softmax(Z, dim=1)
    P = logp.exp()
    H = -(P * logp).sum(1)
    l_ent = H.mean()
    dZ = -P * (logp + H.unsqueeze(1)) / B
    dt = (dZ.t() @ c) / T1
    # semantics reversion on U = v c^T
    U = v @ c.t()
    diag = U.diag().unsqueeze(1)
    raw = diag - U
    cost = raw.clamp(min=0)
    eye = torch.eye(B, dtype=torch.bool)
    cost = cost.masked_fill(eye, 0.0)
    m, j = cost.max(dim=1)
    l_sr = m.mean() / T2
    dU = torch.zeros_like(U)
    ar = torch.arange(B)
    # gradient reaches (i, j*) only where the clamp passes (raw >= 0) and the entry is not the
    # in-place-masked diagonal; through d1 the same amount returns to (i, i).
    act = (raw[ar, j] >= 0) & (j != ar)
    gsr = act.to(t.dtype) / (B * T2)
    dU[ar, j] -= gsr
    dU[ar, ar] += gsr
    dv = dU @ c
    # triplet
    e_tv = t - v + PDIST_EPS
    e_tp = t - p + PDIST_EPS
    d_tv = e_tv.norm(dim=1)
    d_tp = e_tp.norm(dim=1)
    l_tri = (2.0 - d_tv).mean() + 5.0 * d_tp.mean()
    g_tv = e_tv / d_tv.unsqueeze(1) / B
    g_tp = e_tp / d_tp.unsqueeze(1) / B
    dt = dt - g_tv + 5.0 * g_tp
    dv = dv + g_tv
    dp = -5.0 * g_tp
    return torch.stack([l_ent, l_sr, l_tri]), dt, dp, dv


# --------------------------------------------------------------------------------------------
# AdamW (main_TTA.py:271 -> torch.optim.AdamW defaults; train_TTA.py:146-147)
# --------------------------------------------------------------------------------------------
class AdamWState:
    def __init__(self, params: Dict[str, torch.Tensor]):
        self.step = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}


def adamw_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], st: AdamWState,
               lr: float, wd: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """In-place decoupled-decay Adam, torch.optim.AdamW single-tensor formula."""
    st.step += 1
    bc1 = 1.0 - beta1 ** st.step
    bc2 = 1.0 - beta2 ** st.step
    for k, p in params.items():
        g = grads[k]
        p.mul_(1.0 - lr * wd)
        st.m[k].lerp_(g, 1.0 - beta1)
        st.v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st.m[k], denom, value=-(lr / bc1))


def adapted_param_names(sd, only_visual: bool = False) -> List[str]:
    """freeze_parameters / collect_params (open_clip/factory.py:395-437): LayerNorm affine params
    under model.transformer (t2i) or model.visual (i2t), in module-walk order."""
    if only_visual:
        names = ["visual.ln_pre.weight", "visual.ln_pre.bias"]
        pre = "visual.transformer."
    else:
        names, pre = [], "transformer."
    for i in range(n_layers(sd, pre)):
        for ln in ("ln_1", "ln_2"):
            for wb in ("weight", "bias"):
                names.append(f"{pre}resblocks.{i}.{ln}.{wb}")
    if only_visual:
        names += ["visual.ln_post.weight", "visual.ln_post.bias"]
    return names


# --------------------------------------------------------------------------------------------
# the TTA loop (train_TTA.py:98-153)
# --------------------------------------------------------------------------------------------
def tta_step(sd, opt: AdamWState, batch: Sequence[torch.Tensor], G, lr, wd, T1, T2):
    """One inner step (train_TTA.py:133-147).  Returns (losses[3], idx, S, grads)."""
    text, pos, neg, inv = batch
    t, ct = encode_text_fwd(sd, text, True, save=True)
    p, cp = encode_text_fwd(sd, pos, True, save=True)
    n, _ = encode_text_fwd(sd, neg, True, save=False)   # only used under no_grad (:35)
    v, cv = encode_text_fwd(sd, inv, True, save=True)
    c, idx, S = coarse_align_t2i(t, p, n, G)
    losses, dt, dp, dv = tta_loss_fwd_bwd(t, p, v, c, T1, T2)
    grads: Dict[str, torch.Tensor] = {}
    encode_text_bwd(sd, ct, dt, grads)
    encode_text_bwd(sd, cp, dp, grads)
    encode_text_bwd(sd, cv, dv, grads)
    names = adapted_param_names(sd)
    adamw_step({k: sd[k] for k in names}, grads, opt, lr, wd)
    return losses, idx, S, grads, (t, p, n, v)


def train_one_epoch(sd, text_batches, image_batches, n_text: int, n_image: int,
                    lr=2e-4, wd=1e-3, T1=0.03, T2=0.07, steps=3, opt: Optional[AdamWState] = None,
                    log: Optional[list] = None):
    """train_TTA.py:98-153 (t2i branch).  ``text_batches`` yields (text,pos,neg,inv,idx) token
    tensors, ``image_batches`` yields (images, idx).  ``sd`` LN params are updated in place.
    Returns score_matrix_t2i [n_text, n_image] (init -100, :109)."""
    D = sd["text_projection"].shape[1]
    G = torch.zeros(n_image, D)
    for images, index in image_batches:
        G[index] = encode_image_fwd(sd, images, True)
    score = torch.full((n_text, n_image), -100.0)
    if opt is None:
        opt = AdamWState({k: sd[k] for k in adapted_param_names(sd)})
    for text, pos, neg, inv, idx in text_batches:
        for _ in range(steps):
            losses, cidx, S, grads, _ = tta_step(sd, opt, (text, pos, neg, inv), G, lr, wd, T1, T2)
            if log is not None:
                log.append(dict(losses=losses.clone(), idx=cidx.clone()))
        score[idx] = S          # from the LAST step's forward (:150)
    return score, G


# --------------------------------------------------------------------------------------------
# retrieval scoring (evaluation/retrieval.py:186-214, 280-333)
# --------------------------------------------------------------------------------------------
def topk_rows(scores, k: int):
    """Row-wise top-k, descending, ties -> lowest column index first.  (torch.topk's tie order on
    CPU is unspecified; hits depend on the SET only, and that is pinned by the golden cases.)"""
    n = scores.shape[1]
    # stable sort on (-score, index)
    order = torch.sort(-scores.double(), dim=1, stable=True)[1][:, :k]
    return scores.gather(1, order), order


def recall_at_k(scores, positive_pairs, k: int):
    """evaluation/retrieval.py:280-301: |topk ∩ positives| / |positives| per row."""
    idx = topk_rows(scores, k)[1]
    tp = positive_pairs.gather(1, idx).sum(1)
    return tp / positive_pairs.sum(1)


def recall_metrics(scores, positive_pairs, ks=(1, 5, 10)):
    """evaluation/retrieval.py:243-250: image_retrieval = rows of scores, text_retrieval = rows of
    scores.T; metric = mean(recall > 0)."""
    out = {}
    for k in ks:
        out[f"image_retrieval_recall@{k}"] = (recall_at_k(scores, positive_pairs, k) > 0).float().mean().item()
        out[f"text_retrieval_recall@{k}"] = (recall_at_k(scores.t(), positive_pairs.t(), k) > 0).float().mean().item()
    return out


def median_rank(scores, positive_pairs):
    """evaluation/retrieval.py:186-214: median over queries of the best (1-based) rank of a positive."""
    order = torch.sort(-scores.double(), dim=1, stable=True)[1]
    ranks = torch.empty_like(order)
    ranks.scatter_(1, order, torch.arange(scores.shape[1]).expand_as(order))
    pr = (ranks + 1).double().masked_fill(~positive_pairs, float("inf")).min(1)[0]
    pr = pr[pr != float("inf")]
    return pr.median().item()
