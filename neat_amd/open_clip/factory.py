"""Model construction and the parameter-selection helpers of NEAT's TTA path.

``freeze_parameters`` / ``collect_params`` follow reference open_clip/factory.py:395-437 exactly: they
define WHICH parameters adapt (LayerNorm affine under ``model.transformer`` for t2i, under
``model.visual`` for i2t) and in which order the optimizer sees them.
"""
from __future__ import annotations

import json
import os
from copy import deepcopy
from pathlib import Path
from typing import Optional, Union

import torch

from ..synth import MODEL_CFGS
from .model import CLIP, VideoCLIP

_MODEL_CONFIGS = {k: deepcopy(v) for k, v in MODEL_CFGS.items()}


def list_models():
    return list(_MODEL_CONFIGS.keys())


def add_model_config(path):
    """Register architecture JSON file(s) (reference factory.py:62-67)."""
    path = Path(path)
    files = [path] if path.is_file() else sorted(path.glob("*.json"))
    for cf in files:
        with open(cf) as f:
            cfg = json.load(f)
        if all(a in cfg for a in ("embed_dim", "vision_cfg", "text_cfg")):
            _MODEL_CONFIGS[cf.stem] = cfg


def get_model_config(model_name):
    return deepcopy(_MODEL_CONFIGS.get(model_name.replace("/", "-")))


OPENAI_FORMAT_KEYS = ("input_resolution", "context_length", "vocab_size")


def _read_state_dict(checkpoint_path):
    """-> (state dict, is_openai_format).  An OpenAI-format state dict carries three scalar entries beside the weights
    (reference open_clip/model.py:450-506 strips them); those checkpoints were trained with QuickGELU (:498)."""
    sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    if next(iter(sd)).startswith("module"):
        sd = {k[7:]: v for k, v in sd.items()}
    openai = all(k in sd for k in OPENAI_FORMAT_KEYS)
    for k in OPENAI_FORMAT_KEYS:
        sd.pop(k, None)
    return sd, openai


def load_checkpoint(model, checkpoint_path, strict=True, state_dict=None):
    sd = state_dict if state_dict is not None else _read_state_dict(checkpoint_path)[0]
    return model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=strict)


def create_model(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32",
                 device: Union[str, torch.device] = "cpu", force_quick_gelu: bool = False, **model_kwargs):
    """reference factory.py:166-321.  ``pretrained`` may be a local checkpoint path; named tags need the
    network and are unavailable offline.  ``precision`` is accepted for CLI compatibility: master weights
    stay fp32 and the kernels compute bf16 x bf16 -> fp32 regardless.

    Activation: the kernels implement QuickGELU.  A checkpoint is accepted when the config says ``quick_gelu``, when the
    caller passes ``force_quick_gelu=True``, or when the file is an OpenAI-format state dict (it carries
    ``input_resolution / context_length / vocab_size``; the reference enables QuickGELU itself for those,
    open_clip/model.py:498) -- any other checkpoint raises, because an nn.GELU checkpoint has the same keys and would
    load cleanly and give wrong features."""
    cfg = get_model_config(model_name)
    if cfg is None:
        raise RuntimeError(f"Model config for {model_name} not found; available: {list_models()}")
    cfg = dict(cfg, **{k: v for k, v in model_kwargs.items() if k in ("embed_dim", "vision_cfg", "text_cfg")})
    # The kernels implement QuickGELU (OpenAI-family weights; reference model.py:498 builds those with quick_gelu=True).
    # Without a checkpoint the weights are random and QuickGELU is simply this model's activation.  WITH a checkpoint
    # the activation must be the one the weights were trained with: an nn.GELU checkpoint (e.g. laion ViT-B-32) has the
    # same state-dict keys and would load cleanly and give wrong features, so the caller has to say QuickGELU.
    quick_gelu = bool(cfg.get("quick_gelu", False) or force_quick_gelu)
    state_dict = None
    if pretrained and os.path.exists(pretrained):
        state_dict, openai_format = _read_state_dict(pretrained)
        quick_gelu = quick_gelu or openai_format
    if pretrained and not quick_gelu:
        raise RuntimeError(f"create_model({model_name!r}, pretrained={pretrained!r}): only QuickGELU (OpenAI-family) "
                           f"weights are supported; pass force_quick_gelu=True (or a *-quickgelu config) if the "
                           f"checkpoint was trained with QuickGELU")
    model = CLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], quick_gelu=True)
    if pretrained:
        if state_dict is not None:
            load_checkpoint(model, pretrained, state_dict=state_dict)
        else:
            raise RuntimeError(f"Pretrained weights ({pretrained}) not found locally; named tags need network access")
    if "pack_tokens" in model_kwargs:
        model.pack_tokens = bool(model_kwargs["pack_tokens"])
    model.to(device=torch.device(device))
    return model


def create_model_and_transforms(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32",
                                device: Union[str, torch.device] = "cpu", video: bool = False, **kw):
    """reference factory.py:445-510: (model, preprocess_train, preprocess_val).  The transforms are the host half of
    the image pipeline (-> uint8 [H, W, 3]); resize / ToTensor / Normalize run on the GPU per batch in
    ``neat_amd.data.GpuImageLoader`` (see open_clip/transform.py)."""
    from .transform import image_transform
    mean, std = kw.pop("image_mean", None), kw.pop("image_std", None)
    model = create_model(model_name, pretrained, precision=precision, device=device, **kw)
    size = get_model_config(model_name)["vision_cfg"]["image_size"]
    if video:
        model = VideoCLIP(model)
    return model, image_transform(size, True, mean, std), image_transform(size, False, mean, std)


def freeze_parameters(model, only_visual):
    """reference factory.py:395-409."""
    model.train()
    model.requires_grad_(False)
    tower = model.visual if only_visual else model.transformer
    for name, param in tower.named_parameters():
        if "ln" in name:
            param.requires_grad_(True)
    return model


def collect_params(model, only_visual):
    """reference factory.py:411-437: (params, names) of LayerNorm weight/bias in module-walk order."""
    params, names = [], []
    tower = model.visual if only_visual else model.transformer
    for nm, m in tower.named_modules():
        if isinstance(m, torch.nn.LayerNorm):
            for np_, p in m.named_parameters():
                if np_ in ("weight", "bias"):
                    params.append(p)
                    names.append(f"{nm}.{np_}")
    return params, names
