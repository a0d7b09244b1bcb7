"""Host-side helpers with the reference's names (reference src/tta_utils.py): LR schedules (:4-23), rank
helpers (:233-259) and the LayerNorm export used by the PLIP workflow (:287-341)."""
import math

import torch
import torch.distributed as dist


def cosine_lr_schedule(optimizer, epoch, max_epoch, init_lr, min_lr):
    lr = (init_lr - min_lr) * 0.5 * (1. + math.cos(math.pi * epoch / max_epoch)) + min_lr
    for g in optimizer.param_groups:
        g['lr'] = lr


def warmup_lr_schedule(optimizer, step, max_step, init_lr, max_lr):
    lr = min(max_lr, init_lr + (max_lr - init_lr) * step / max_step)
    for g in optimizer.param_groups:
        g['lr'] = lr


def step_lr_schedule(optimizer, epoch, init_lr, min_lr, decay_rate):
    lr = max(min_lr, init_lr * (decay_rate ** epoch))
    for g in optimizer.param_groups:
        g['lr'] = lr


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def export_text_layernorm_params(model):
    """Adapted text-tower LN parameters as a {name: cpu tensor} dict (checkpoint / parity dump)."""
    out = {}
    for i, b in enumerate(model.transformer.resblocks):
        for ln in ("ln_1", "ln_2"):
            m = getattr(b, ln)
            out[f"transformer.resblocks.{i}.{ln}.weight"] = m.weight.detach().float().cpu().clone()
            out[f"transformer.resblocks.{i}.{ln}.bias"] = m.bias.detach().float().cpu().clone()
    return out


def replace_text_layernorm_params(hf_model, openai_model):
    """Copy the adapted text-tower LayerNorm parameters into a HuggingFace CLIPModel
    (reference tta_utils.py:287-341): resblocks[i].ln_1 -> layers[i].layer_norm1, ln_2 -> layer_norm2."""
    enc = hf_model.text_model
    tr = openai_model.transformer
    n = len(enc.encoder.layers)
    for i in range(n):
        hl, ol = enc.encoder.layers[i], tr.resblocks[i]
        if hasattr(hl, 'layer_norm1'):
            hl.layer_norm1.weight.data = ol.ln_1.weight.data.detach().clone().to(hl.layer_norm1.weight.device)
            hl.layer_norm1.bias.data = ol.ln_1.bias.data.detach().clone().to(hl.layer_norm1.bias.device)
        if hasattr(hl, 'layer_norm2'):
            hl.layer_norm2.weight.data = ol.ln_2.weight.data.detach().clone().to(hl.layer_norm2.weight.device)
            hl.layer_norm2.bias.data = ol.ln_2.bias.data.detach().clone().to(hl.layer_norm2.bias.device)
    if hasattr(tr, 'ln_final') and hasattr(enc, 'final_layer_norm'):
        enc.final_layer_norm.weight.data = tr.ln_final.weight.data.detach().clone()
        enc.final_layer_norm.bias.data = tr.ln_final.bias.data.detach().clone()
    return n
