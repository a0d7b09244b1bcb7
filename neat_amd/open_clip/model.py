"""CLIP model facade whose encoders run on libneat_hip.so.

Mirrors the reference's model surface for the TTA path (open_clip/model.py:220-317,
open_clip/transformer.py:188-243,291-325,328-547): same attribute names (``visual``, ``transformer``,
``token_embedding``, ``positional_embedding``, ``ln_final``, ``text_projection``, ``logit_scale``), same
state-dict keys (so OpenAI / open_clip checkpoints load unchanged) and the same ``encode_image`` /
``encode_text`` signatures.  The modules below are parameter containers only: all arithmetic is done by
the HIP kernels through ``neat_amd.engine`` -- there is no eager PyTorch path, and a CPU tensor or a
missing library raises.
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from .. import engine, ops


class LayerNorm(nn.LayerNorm):
    """Parameter holder; subclassing nn.LayerNorm keeps ``collect_params``' isinstance test
    (open_clip/factory.py:421,428) and the ``ln`` name filter of ``freeze_parameters`` working."""

    def forward(self, x):  # pragma: no cover - containers are never called
        raise RuntimeError("neat_amd: LayerNorm runs inside the fused HIP kernels, not as a module call")


class _Linear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))


class _MHA(nn.Module):
    """Parameter layout of nn.MultiheadAttention (in_proj_weight [3C,C], in_proj_bias, out_proj)."""

    def __init__(self, d_model):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * d_model))
        self.out_proj = _Linear(d_model, d_model)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.ln_1 = LayerNorm(d_model)
        self.attn = _MHA(d_model)
        self.ln_2 = LayerNorm(d_model)
        width = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", _Linear(d_model, width)), ("c_proj", _Linear(width, d_model))]))


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        if width != heads * 64:
            raise ValueError(f"neat_amd attention kernels need head_dim 64 (width {width}, heads {heads})")
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, mlp_ratio) for _ in range(layers)])

    def get_cast_dtype(self) -> torch.dtype:
        return self.resblocks[0].mlp.c_fc.weight.dtype


class _Conv(nn.Module):
    def __init__(self, width, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))


class VisionTransformer(nn.Module):
    def __init__(self, image_size: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 mlp_ratio: float = 4.0):
        super().__init__()
        self.image_size = (image_size, image_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.output_dim = output_dim
        self.conv1 = _Conv(width, patch_size)
        self.class_embedding = nn.Parameter(torch.empty(width))
        self.positional_embedding = nn.Parameter(torch.empty(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(torch.empty(width, output_dim))


class _EncodeText(torch.autograd.Function):
    """Autograd bridge for the drop-in API: forward = HIP text tower with saved activations, backward =
    hand-written HIP backward that deposits d(gamma), d(beta) into the LN parameters' .grad."""

    @staticmethod
    def forward(ctx, model, tokens, normalize, save, *ln_params):
        tower = model._text_tower()
        feat, tctx = tower.forward(tokens, normalize, save, packed=bool(getattr(model, "pack_tokens", False)))
        ctx.tower, ctx.tctx, ctx.model = tower, tctx, model
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        if ctx.tctx is None:
            raise RuntimeError("encode_text backward requested but the forward ran without saving activations")
        blocks = ctx.model.transformer.resblocks
        dev = dfeat.device
        grads = [dict(ln1_w=torch.empty_like(b.ln_1.weight), ln1_b=torch.empty_like(b.ln_1.bias),
                      ln2_w=torch.empty_like(b.ln_2.weight), ln2_b=torch.empty_like(b.ln_2.bias)) for b in blocks]
        ctx.tower.backward(ctx.tctx, dfeat.contiguous().float(), grads, accumulate=False)
        out = []
        for g in grads:
            out += [g["ln1_w"], g["ln1_b"], g["ln2_w"], g["ln2_b"]]
        return (None, None, None, None, *out)


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, vision_cfg: dict, text_cfg: dict, quick_gelu: bool = True,
                 init_logit_scale: float = math.log(1 / 0.07), **_unused):
        super().__init__()
        if not quick_gelu:
            raise ValueError("neat_amd implements the QuickGELU CLIP family (OpenAI weights, model.py:498) only")
        v, t = dict(vision_cfg), dict(text_cfg)
        self.visual = VisionTransformer(v["image_size"], v["patch_size"], v["width"], v["layers"],
                                        v["width"] // v.get("head_width", 64), embed_dim, v.get("mlp_ratio", 4.0))
        self.transformer = Transformer(t["width"], t["layers"], t["heads"], t.get("mlp_ratio", 4.0))
        self.context_length = t["context_length"]
        self.vocab_size = t["vocab_size"]
        self.token_embedding = nn.Embedding(t["vocab_size"], t["width"])
        self.positional_embedding = nn.Parameter(torch.empty(t["context_length"], t["width"]))
        self.ln_final = LayerNorm(t["width"])
        self.text_projection = nn.Parameter(torch.empty(t["width"], embed_dim))
        self.text_pool_type = "argmax"
        self.logit_scale = nn.Parameter(torch.ones([]) * init_logit_scale)
        self.logit_bias = None
        self._towers = {}
        # default ON: the text tower runs only positions 0..EOT of each caption (engine.TextTower.forward, packed=True):
        # behind the causal mask (reference transformer.py:655-661) and EOT pooling (:555-558) the later positions can
        # neither reach the pooled feature nor receive gradient from it -- same features, losses and LN gradients up to
        # fp32 summation order.  NEAT_PACK_TOKENS=0 (or model.pack_tokens = False) runs all context_length positions
        # like the reference does.
        self.pack_tokens = os.environ.get("NEAT_PACK_TOKENS", "1") not in ("", "0")
        self.init_parameters()

    # -- init: same distributions as the reference's text tower (open_clip/transformer.py:628-649) --
    def init_parameters(self):
        def init_stack(tr):
            proj_std = (tr.width ** -0.5) * ((2 * tr.layers) ** -0.5)
            attn_std, fc_std = tr.width ** -0.5, (2 * tr.width) ** -0.5
            for b in tr.resblocks:
                nn.init.normal_(b.attn.in_proj_weight, std=attn_std)
                nn.init.zeros_(b.attn.in_proj_bias)
                nn.init.normal_(b.attn.out_proj.weight, std=proj_std)
                nn.init.zeros_(b.attn.out_proj.bias)
                nn.init.normal_(b.mlp.c_fc.weight, std=fc_std)
                nn.init.zeros_(b.mlp.c_fc.bias)
                nn.init.normal_(b.mlp.c_proj.weight, std=proj_std)
                nn.init.zeros_(b.mlp.c_proj.bias)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        init_stack(self.transformer)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)
        v = self.visual
        scale = v.transformer.width ** -0.5
        nn.init.normal_(v.class_embedding, std=scale)
        nn.init.normal_(v.positional_embedding, std=scale)
        nn.init.normal_(v.proj, std=scale)
        nn.init.normal_(v.conv1.weight, std=(3 * v.patch_size[0] ** 2) ** -0.5)
        init_stack(v.transformer)

    # -- engine plumbing --
    def invalidate_kernel_weights(self):
        """Call after changing any FROZEN weight in place (LN gamma/beta are read live and need no call)."""
        self._towers = {}

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_kernel_weights()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._towers = {}
        return r

    def _text_tower(self) -> engine.TextTower:
        if "text" not in self._towers:
            self._require_gpu(self.text_projection)
            self._towers["text"] = engine.TextTower(self, need_bwd=True)
        return self._towers["text"]

    def _vision_tower(self) -> engine.VisionTower:
        if "vision" not in self._towers:
            self._require_gpu(self.visual.proj)
            self._towers["vision"] = engine.VisionTower(self)
        return self._towers["vision"]

    @staticmethod
    def _require_gpu(p):
        if not p.is_cuda:
            raise RuntimeError("neat_amd: the model must live on an MI355X GPU (model.to('cuda')); no CPU path")
        if p.dtype != torch.float32:
            raise RuntimeError("neat_amd: keep master weights in float32 (kernels cast GEMM operands to bf16)")

    def _text_ln_params(self):
        out = []
        for b in self.transformer.resblocks:
            out += [b.ln_1.weight, b.ln_1.bias, b.ln_2.weight, b.ln_2.bias]
        return out

    # -- the reference's public surface --
    def encode_image(self, image, normalize: bool = False):
        """open_clip/model.py:265-267 (forward only: the gallery is built under no_grad, train_TTA.py:120)."""
        if not image.is_cuda:
            raise RuntimeError("encode_image: expected a GPU tensor")
        return self._vision_tower().forward(image, normalize)

    def encode_text(self, text, normalize: bool = False):
        """open_clip/model.py:269-286; differentiable w.r.t. the text tower's LayerNorm gamma/beta."""
        if not text.is_cuda:
            raise RuntimeError("encode_text: expected a GPU tensor")
        if text.dtype != torch.int64:
            text = text.long()
        ln = self._text_ln_params()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in ln)   # activations kept only if needed
        return _EncodeText.apply(self, text, normalize, save, *ln)

    def get_logits(self, image, text):
        image_features = self.encode_image(image, normalize=True)
        text_features = self.encode_text(text, normalize=True)
        image_logits = self.logit_scale.exp() * ops.gemm_f32_nt(image_features.contiguous(), text_features.detach().contiguous())
        return image_logits, image_logits.T

    def forward(self, image: Optional[torch.Tensor] = None, text: Optional[torch.Tensor] = None):
        image_features = self.encode_image(image, normalize=True) if image is not None else None
        text_features = self.encode_text(text, normalize=True) if text is not None else None
        return image_features, text_features, self.logit_scale.exp()


class VideoCLIP(nn.Module):
    """utils/video_utils/model.py:24-91: frames folded into the batch, per-frame L2-norm, mean over frames,
    re-normalise.  Exposes .transformer/.visual so freeze/collect work (model.py:24-27)."""

    def __init__(self, clip: CLIP):
        super().__init__()
        self.clip = clip
        self.transformer = clip.transformer
        self.visual = clip.visual

    def encode_text(self, text, normalize: bool = False):
        return self.clip.encode_text(text, normalize)

    def encode_video(self, video, normalize: bool = True):
        n, F = video.shape[:2]
        feat = self.clip.encode_image(video.reshape(n * F, *video.shape[2:]), normalize=True)
        return ops.frame_mean_norm(feat.contiguous(), n, F)

    def encode_image(self, image, normalize: bool = False):
        return self.encode_video(image, normalize) if image.dim() == 5 else self.clip.encode_image(image, normalize)
