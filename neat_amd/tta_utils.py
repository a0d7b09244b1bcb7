"""Host-side helpers of the TTA path with the reference's names (reference src/tta_utils.py): the rank helpers
train_one_epoch uses (:233-259) and the hand-over of adapted LayerNorm parameters to a HuggingFace CLIP text tower
(:287-341, the PLIP workflow).  The reference's LR schedules and meters are never called on this path and are not here."""
import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def _text_ln_pairs(model):
    """(state-dict name, module) of every adapted LayerNorm of the text tower, in block order."""
    for i, blk in enumerate(model.transformer.resblocks):
        yield f"transformer.resblocks.{i}.ln_1", blk.ln_1
        yield f"transformer.resblocks.{i}.ln_2", blk.ln_2


def export_text_layernorm_params(model):
    """The adapted parameters as a {state-dict key: fp32 CPU tensor} dict -- the only state a TTA run produces
    (24 576 floats for ViT-B-32); ``model.load_state_dict(d, strict=False)`` restores it."""
    out = {}
    for name, ln in _text_ln_pairs(model):
        out[name + ".weight"] = ln.weight.detach().float().cpu().clone()
        out[name + ".bias"] = ln.bias.detach().float().cpu().clone()
    return out


@torch.no_grad()
def replace_text_layernorm_params(hf_model, openai_model):
    """Write the text tower's LayerNorm parameters of ``openai_model`` (open_clip layout) into a HuggingFace ``CLIPModel``:
    block i's ln_1 / ln_2 become ``text_model.encoder.layers[i].layer_norm1 / layer_norm2``, and ``ln_final`` becomes
    ``text_model.final_layer_norm`` when both sides have one (same mapping and return value as reference
    tta_utils.py:287-341: the number of HF layers visited).  Values are copied into the HF parameters in place, on
    whatever device and dtype they live."""
    hf_text = hf_model.text_model
    layers = hf_text.encoder.layers
    blocks = openai_model.transformer.resblocks
    for hf_layer, blk in zip(layers, blocks):
        for hf_name, src in (("layer_norm1", blk.ln_1), ("layer_norm2", blk.ln_2)):
            dst = getattr(hf_layer, hf_name, None)
            if dst is not None:
                dst.weight.copy_(src.weight.detach().to(dst.weight))
                dst.bias.copy_(src.bias.detach().to(dst.bias))
    src_final = getattr(openai_model.transformer, "ln_final", None)     # as the reference looks it up (absent in CLIP)
    dst_final = getattr(hf_text, "final_layer_norm", None)
    if src_final is not None and dst_final is not None:
        dst_final.weight.copy_(src_final.weight.detach().to(dst_final.weight))
        dst_final.bias.copy_(src_final.bias.detach().to(dst_final.bias))
    return len(layers)
