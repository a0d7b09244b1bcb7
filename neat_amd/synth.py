"""Deterministic synthetic weights and inputs for the NEAT TTA hot path.

There are no checkpoints or NegBench data offline, so every parity
test and the benchmark run on seeded synthetic tensors.  The same generators are used
by ``tests/golden/make_golden.py`` (which feeds them to the *reference* model through
``load_state_dict``) and by the GPU tests (which feed them to the HIP engine), so both
sides see bit-identical fp32 weights and inputs without the reference ever travelling.

State-dict keys follow open_clip's CLIP layout (reference ``src/open_clip/model.py:455-479``,
``src/open_clip/transformer.py:201-213,362-431``) so real checkpoints load unchanged.
"""
from __future__ import annotations

import hashlib
from typing import Dict

import torch

# Architecture facts (reference src/open_clip/model_configs/*.json).  ``neat-tiny`` is a
# test-only config with head_dim 64 so the HIP attention kernels run on it unchanged.
MODEL_CFGS: Dict[str, dict] = {
    "ViT-B-32": dict(embed_dim=512,
                     vision_cfg=dict(image_size=224, layers=12, width=768, patch_size=32),
                     text_cfg=dict(context_length=77, vocab_size=49408, width=512, heads=8, layers=12)),
    "ViT-B-16": dict(embed_dim=512,
                     vision_cfg=dict(image_size=224, layers=12, width=768, patch_size=16),
                     text_cfg=dict(context_length=77, vocab_size=49408, width=512, heads=8, layers=12)),
    "ViT-L-14": dict(embed_dim=768,
                     vision_cfg=dict(image_size=224, layers=24, width=1024, patch_size=14),
                     text_cfg=dict(context_length=77, vocab_size=49408, width=768, heads=12, layers=12)),
    # embed_dim stays 512 because the reference hard-codes a 512-column gallery buffer
    # (train_TTA.py:119), and the golden fixtures come from its unmodified train_one_epoch.
    "neat-tiny": dict(embed_dim=512,
                      vision_cfg=dict(image_size=32, layers=2, width=128, patch_size=16),
                      text_cfg=dict(context_length=16, vocab_size=512, width=128, heads=2, layers=2)),
}

HEAD_WIDTH = 64  # CLIPVisionCfg.head_width default (reference model.py:30)


def get_cfg(name: str) -> dict:
    name = name.replace("/", "-")
    if name not in MODEL_CFGS:
        raise KeyError(f"unknown model config {name!r}; known: {sorted(MODEL_CFGS)}")
    c = MODEL_CFGS[name]
    return dict(embed_dim=c["embed_dim"], vision_cfg=dict(c["vision_cfg"]), text_cfg=dict(c["text_cfg"]))


def _gen(seed: int, key: str) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:7], "little"))
    return g


def _approx_normal(shape, g: torch.Generator) -> torch.Tensor:
    """Unit-variance bell-shaped samples built from integer draws only (sum of four U{0..2^20-1}),
    so the values are bit-identical on every host: ``torch.randn`` goes through vectorised
    log/cos whose last bits may depend on the CPU's ISA, ``torch.randint`` is exact."""
    acc = torch.zeros(shape, dtype=torch.int32)
    for _ in range(4):
        acc += torch.randint(0, 1 << 20, shape, generator=g, dtype=torch.int32)
    acc -= 2 * ((1 << 20) - 1)
    return acc.to(torch.float32) * (3.0 ** 0.5 / float(1 << 20))


def _randn(seed, key, shape, std=1.0, mean=0.0):
    x = _approx_normal(tuple(shape), _gen(seed, key))
    if std != 1.0:
        x = x * std
    if mean != 0.0:
        x = x + mean
    return x


def _block_keys(prefix: str, width: int, layers: int, seed: int, sd: dict, mlp_ratio: float = 4.0,
                branch_gain: float = 1.0):
    """One transformer stack; std values follow the text-tower init (transformer.py:634-641).  ``branch_gain`` scales the
    two matrices that write into the residual stream (out_proj, c_proj)."""
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5) * branch_gain
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    hidden = int(width * mlp_ratio)
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        # LayerNorm affine: perturbed away from (1, 0) so that relative-error gates on the
        # adapted parameters are meaningful (pretrained CLIP LN params are far from identity).
        sd[p + "ln_1.weight"] = _randn(seed, p + "ln_1.weight", (width,), 0.2, 1.0)
        sd[p + "ln_1.bias"] = _randn(seed, p + "ln_1.bias", (width,), 0.1)
        sd[p + "attn.in_proj_weight"] = _randn(seed, p + "attn.in_proj_weight", (3 * width, width), attn_std)
        sd[p + "attn.in_proj_bias"] = _randn(seed, p + "attn.in_proj_bias", (3 * width,), 0.02)
        sd[p + "attn.out_proj.weight"] = _randn(seed, p + "attn.out_proj.weight", (width, width), proj_std)
        sd[p + "attn.out_proj.bias"] = _randn(seed, p + "attn.out_proj.bias", (width,), 0.02)
        sd[p + "ln_2.weight"] = _randn(seed, p + "ln_2.weight", (width,), 0.2, 1.0)
        sd[p + "ln_2.bias"] = _randn(seed, p + "ln_2.bias", (width,), 0.1)
        sd[p + "mlp.c_fc.weight"] = _randn(seed, p + "mlp.c_fc.weight", (hidden, width), fc_std)
        sd[p + "mlp.c_fc.bias"] = _randn(seed, p + "mlp.c_fc.bias", (hidden,), 0.02)
        sd[p + "mlp.c_proj.weight"] = _randn(seed, p + "mlp.c_proj.weight", (width, hidden), proj_std)
        sd[p + "mlp.c_proj.bias"] = _randn(seed, p + "mlp.c_proj.bias", (width,), 0.02)


def synth_state_dict(model_name: str, seed: int = 0, branch_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 CLIP weights keyed like an open_clip ``CLIP.state_dict()``.

    Each tensor draws from its own generator (seed, key) so the result does not depend on
    construction order.  Token-embedding std is larger than open_clip's 0.02 so that different
    captions give clearly different features (a collapsed random transformer would make the
    cosine gates meaningless)."""
    cfg = get_cfg(model_name)
    v, t, D = cfg["vision_cfg"], cfg["text_cfg"], cfg["embed_dim"]
    sd: Dict[str, torch.Tensor] = {}
    vw = v["width"]
    grid = v["image_size"] // v["patch_size"]
    scale = vw ** -0.5
    sd["visual.class_embedding"] = _randn(seed, "visual.class_embedding", (vw,), scale)
    sd["visual.positional_embedding"] = _randn(seed, "visual.positional_embedding", (grid * grid + 1, vw), scale)
    sd["visual.proj"] = _randn(seed, "visual.proj", (vw, D), scale)
    fan_in = 3 * v["patch_size"] ** 2
    sd["visual.conv1.weight"] = _randn(seed, "visual.conv1.weight", (vw, 3, v["patch_size"], v["patch_size"]),
                                       fan_in ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = _randn(seed, f"visual.{nm}.weight", (vw,), 0.2, 1.0)
        sd[f"visual.{nm}.bias"] = _randn(seed, f"visual.{nm}.bias", (vw,), 0.1)
    _block_keys("visual.transformer.", vw, v["layers"], seed, sd, branch_gain=branch_gain)

    tw = t["width"]
    sd["token_embedding.weight"] = _randn(seed, "token_embedding.weight", (t["vocab_size"], tw), 0.3)
    sd["positional_embedding"] = _randn(seed, "positional_embedding", (t["context_length"], tw), 0.05)
    _block_keys("transformer.", tw, t["layers"], seed, sd, branch_gain=branch_gain)
    sd["ln_final.weight"] = _randn(seed, "ln_final.weight", (tw,), 0.2, 1.0)
    sd["ln_final.bias"] = _randn(seed, "ln_final.bias", (tw,), 0.1)
    sd["text_projection"] = _randn(seed, "text_projection", (tw, D), tw ** -0.5)
    sd["logit_scale"] = torch.tensor(2.6592600345611572, dtype=torch.float32)  # ln(1/0.07), model.py:229
    return sd


def synth_tokens(n: int, context_length: int, vocab_size: int, gen: torch.Generator,
                 min_len: int = 5, max_len: int = 20) -> torch.Tensor:
    """Rows ``[SOT, r_1..r_l, EOT, 0...]`` (int64), l ~ U{min_len..max_len}, r ~ U{1..SOT-1}.

    EOT is the largest id in the row, which is what ``text_global_pool(..., 'argmax')`` relies on
    (reference transformer.py:555-558).  SOT/EOT are vocab_size-2 / vocab_size-1 (49406/49407 for
    the CLIP BPE vocabulary, reference tokenizer.py:127-259)."""
    sot, eot = vocab_size - 2, vocab_size - 1
    max_len = min(max_len, context_length - 2)
    min_len = min(min_len, max_len)
    lens = torch.randint(min_len, max_len + 1, (n,), generator=gen)
    body = torch.randint(1, sot, (n, context_length), generator=gen)
    out = torch.zeros(n, context_length, dtype=torch.int64)
    out[:, 0] = sot
    for i in range(n):
        l = int(lens[i])
        out[i, 1:1 + l] = body[i, :l]
        out[i, 1 + l] = eot
    return out


def synth_caption_sets(n: int, context_length: int, vocab_size: int, seed: int,
                       min_len: int = 5, max_len: int = 20):
    """Four independent token sets (text, pos, neg, inv), SURVEY.md §8(d) config 1."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return tuple(synth_tokens(n, context_length, vocab_size, g, min_len, max_len) for _ in range(4))


def _edit_caption(row: torch.Tensor, vocab_size: int, gen: torch.Generator, n_sub: int, insert=()) -> torch.Tensor:
    """A caption derived from ``row``: n_sub body tokens replaced by random ids, ``insert`` spliced in at a random place
    (when the context has room).  Integer draws only."""
    sot, eot = vocab_size - 2, vocab_size - 1
    L = row.shape[0]
    l = int(row.argmax())                                   # EOT position
    body = row[1:l].clone()
    if n_sub:
        at = torch.randint(0, len(body), (n_sub,), generator=gen)
        body[at] = torch.randint(1, sot, (n_sub,), generator=gen)
    if len(insert) and l + len(insert) < L - 1:
        at = int(torch.randint(0, len(body) + 1, (1,), generator=gen))
        body = torch.cat([body[:at], torch.tensor(list(insert), dtype=row.dtype), body[at:]])
    out = torch.zeros_like(row)
    out[0] = sot
    out[1:1 + len(body)] = body
    out[1 + len(body)] = eot
    return out


def synth_related_caption_sets(n: int, context_length: int, vocab_size: int, seed: int,
                               min_len: int = 5, max_len: int = 20):
    """(text, pos, neg, inv) token sets that are RELATED the way NegBench's are (reference
    data/coco_negated_dataset.py:63-79: a caption, its paraphrase, a caption about what is absent, its negated rewrite):
    pos = the caption with one token replaced; inv = the caption with a two-token "negation" spliced in and one token
    replaced; neg = an edit of ANOTHER caption (row (r + n//2) % n).  Unlike ``synth_caption_sets`` (four independent
    draws) the four features of a row are then correlated like those of a real row."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    text = synth_tokens(n, context_length, vocab_size, g, min_len, max_len)
    marker = (783, 1000) if vocab_size > 2000 else (7, 11)
    pos = torch.stack([_edit_caption(text[r], vocab_size, g, 1) for r in range(n)])
    neg = torch.stack([_edit_caption(text[(r + n // 2) % n], vocab_size, g, 1) for r in range(n)])
    inv = torch.stack([_edit_caption(text[r], vocab_size, g, 1, marker) for r in range(n)])
    return text, pos, neg, inv


SPREAD_BRANCH_GAIN = 4.0


def spread_state_dict(model_name: str, seed: int = 0) -> Dict[str, torch.Tensor]:
    """The "spread" weights of tests/golden/make_spread_weights.py: the synthetic weights with the residual branches
    scaled by SPREAD_BRANCH_GAIN (every layer matters, as in a trained tower) and with ``visual.proj`` /
    ``text_projection`` replaced by the two FITTED matrices stored (as bf16 bit patterns, exact in fp32) in
    tests/golden/spread_<model>.npz, which place gallery images and captions in a shared space the way a trained CLIP
    does: features are far from collinear and recall@1 is far from chance."""
    import os

    import numpy as np
    sd = synth_state_dict(model_name, seed, branch_gain=SPREAD_BRANCH_GAIN)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        f"spread_{model_name.lower().replace('-', '')}.npz")
    z = np.load(path)
    for key in ("visual.proj", "text_projection"):
        bits = torch.from_numpy(z[key.replace(".", "_")].astype(np.int32)) << 16
        w = bits.view(torch.float32)
        assert w.shape == sd[key].shape
        sd[key] = w.clone()
    return sd


def synth_images(n: int, image_size: int, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return _approx_normal((n, 3, image_size, image_size), g)
