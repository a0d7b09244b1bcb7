"""Shared helpers for the parity tests: fixture loading and the synthetic batch recipe.

``make_batches`` must stay identical to the recipe in tests/golden/make_golden.py (which cannot be
imported on the GPU box because it needs /root/reference)."""
import json
import os

import numpy as np
import torch

from neat_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN, f"{tag}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "meta" in d:
        d["meta"] = json.loads(str(d["meta"]))
    return d


def make_batches(cfg, n_text, n_image, bs_text, bs_image, data_seed, related=False):
    t = cfg["text_cfg"]
    sets = (synth.synth_related_caption_sets if related else synth.synth_caption_sets)(
        n_text, t["context_length"], t["vocab_size"], data_seed)
    images = synth.synth_images(n_image, cfg["vision_cfg"]["image_size"], data_seed + 1)
    g = torch.Generator().manual_seed(data_seed + 2)
    perm_t = torch.randperm(n_text, generator=g)
    perm_i = torch.randperm(n_image, generator=g)
    tb = [tuple(s[perm_t[i:i + bs_text]] for s in sets) + (perm_t[i:i + bs_text],)
          for i in range(0, n_text, bs_text)]
    ib = [(images[perm_i[i:i + bs_image]], perm_i[i:i + bs_image]) for i in range(0, n_image, bs_image)]
    return tb, ib


def weight_checksums(sd):
    keys = ["token_embedding.weight", "transformer.resblocks.0.attn.in_proj_weight", "visual.conv1.weight",
            "text_projection"]
    return np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()] for k in keys])


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def min_cosine(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return torch.nn.functional.cosine_similarity(a, b, dim=-1).min().item()
