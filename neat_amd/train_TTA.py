"""NEAT test-time adaptation loop on HIP kernels -- host side.

Same entry point and semantics as reference src/train_TTA.py:98-153 (``train_one_epoch``) plus the loss
pieces it calls (:31-42, :45-48, :53-65, :77-91) as thin wrappers over the fused kernels.

What differs from the reference, by design:
  * the four caption variants of a batch go through the text tower as ONE batch of 4*B sequences
    (bigger GEMM M), ordered [text, pos, inv, neg]; the backward runs over the first 3*B only because
    neg_text_feat is used solely under no_grad (train_TTA.py:35,137);
  * loss, gradient and the coarse alignment are fused fp32 kernels; AdamW is one launch when the optimizer
    is ``neat_amd.optim.AdamW`` (any torch optimizer over the same parameters also works);
  * the gallery buffer is sized from the model's embed dim (the reference hard-codes 512, :119);
  * opt-in ``model.pack_tokens = True``: the text tower runs only positions 0..EOT of every caption (dead rows
    behind the causal mask are dropped; same features, losses and LN gradients up to fp32 summation order);
  * data parallelism: every rank iterates the same batches, owns a contiguous row shard of
    each, all-gathers text features + coarse image features, and all-reduces the LN gradients.
"""
from __future__ import annotations

import logging
from typing import Optional

import os

import torch
import torch.distributed as dist

from . import ops
from .tta_utils import get_rank, get_world_size


# ---- reference-named helpers (kernel-backed) ----
@torch.no_grad()
def coarse_aligning_t2i(query_feat, pos_query_feat, neg_query_feat, gallery_feat_all):
    """train_TTA.py:31-42.  Returns (gallery_feat, coarse_idx, soft_labels(None), pred_idx_knn, scores_all_):
    ``soft_t2i_labels`` (a [B,B] gather that nothing downstream reads) is not materialised."""
    coarse, idx, S = ops.coarse_align(query_feat.detach().contiguous(), pos_query_feat.detach().contiguous(),
                                      neg_query_feat.detach().contiguous(), gallery_feat_all.contiguous())
    return coarse, idx.long(), None, ops.topk_rows(S, 1)[1][:, 0].long(), S


def tta_loss(text_feat, pos_text_feat, inv_text_feat, coarse_img_feat, T1, T2):
    """loss_entropy, loss_sr, loss_tri (train_TTA.py:138-143) and their gradients, no autograd."""
    return ops.tta_loss_fwd_bwd(text_feat, pos_text_feat, inv_text_feat, text_feat, coarse_img_feat, T1, T2, 0)


# NEAT_FORCE_COLLECTIVES=1: run every collective of the data-parallel step even at world size 1 (a one-GPU rehearsal of the
# RCCL call path: process group, all_gather of padded row shards, all_reduce of the flat LN gradient)
FORCE_COLLECTIVES = os.environ.get("NEAT_FORCE_COLLECTIVES", "") == "1"


def _shard(n: int, world: int, rank: int):
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_rows(x, n_total: int, world: int):
    """Concatenate row shards of (possibly) different sizes from all ranks, in rank order."""
    sizes = [_shard(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == mx for lo, hi in sizes):                    # even shards: one output tensor, no padding, no cat
        out = torch.empty(world * mx, *x.shape[1:], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out
    pad = torch.zeros(mx, *x.shape[1:], dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def _ln_grad_views(model):
    """Per-block dicts of the .grad tensors the backward kernels write into (allocated on demand)."""
    grads = []
    for b in model.transformer.resblocks:
        g = {}
        for key, p in (("ln1_w", b.ln_1.weight), ("ln1_b", b.ln_1.bias), ("ln2_w", b.ln_2.weight), ("ln2_b", b.ln_2.bias)):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            g[key] = p.grad
        grads.append(g)
    return grads


def tta_step(model, optimizer, tokens4, gallery, T1, T2, world=1, rank=0, global_rows=None):
    """One inner step (train_TTA.py:133-147) on this rank's row shard.  tokens4 = (text,pos,neg,inv) int64
    [Bl, L] on the GPU.  Returns (losses[3] tensor, coarse idx, sims_matrix) for the local rows.
    global_rows: rows of the global batch when the caller knows them (shards cut by ``_shard``); without it the
    ranks add up their row counts, which costs a collective and a host synchronisation per step."""
    text, pos, neg, inv = tokens4
    Bl = text.shape[0]
    tower = model._text_tower()
    feats, ctx = tower.forward(torch.cat([text, pos, inv, neg], dim=0), True, True,
                               packed=bool(getattr(model, "pack_tokens", False)), n_grad=3 * Bl)     # neg: no backward
    t, p, v, n = feats[:Bl], feats[Bl:2 * Bl], feats[2 * Bl:3 * Bl], feats[3 * Bl:]
    coarse, idx, S = ops.coarse_align(t, p, n, gallery)
    if world > 1 or FORCE_COLLECTIVES:
        if global_rows is None:
            B = torch.tensor([Bl], device=t.device)
            dist.all_reduce(B)
            B = int(B.item())
        else:
            B = int(global_rows)
        row0 = _shard(B, world, rank)[0]
        D = t.shape[1]
        tc_all = _all_gather_rows(torch.cat([t, coarse], dim=1), B, world)     # one collective for both
        t_all, c_all = tc_all[:, :D].contiguous(), tc_all[:, D:].contiguous()
    else:
        B, row0, t_all, c_all = Bl, 0, t, coarse
    losses, dt, dp, dv = ops.tta_loss_fwd_bwd(t, p, v, t_all, c_all, T1, T2, row0)
    grads = _ln_grad_views(model)
    tower.backward(ctx, torch.cat([dt, dp, dv], dim=0), grads, accumulate=False)
    if world > 1 or FORCE_COLLECTIVES:
        buf = getattr(optimizer, "grad_buf", None)
        if buf is not None:
            # neat_amd.optim.AdamW: gradients + the two row-summed loss terms travel as ONE collective.  An all-gather
            # of the 98-KB buffers and a local sum in rank order instead of an all-reduce: every rank adds the same
            # numbers in the same order (bit-identical parameters on all ranks by construction), and through
            # torch.distributed on this stack the all-gather costs no host time while an all_reduce call blocks the
            # host for 1.7 ms (tools/rccl_probe.py), which the launch-bound tail of the step cannot hide.
            buf[-2:].copy_(losses[1:])
            gathered = torch.empty(world, buf.numel(), dtype=buf.dtype, device=buf.device)
            dist.all_gather_into_tensor(gathered.view(-1), buf)
            torch.sum(gathered, dim=0, out=buf)
            losses[1:].copy_(buf[-2:])
        else:
            for g in grads:
                for x in g.values():
                    dist.all_reduce(x)
            dist.all_reduce(losses[1:])      # sr / triplet are sums over local rows; entropy is already global
    optimizer.step()
    optimizer.zero_grad()
    return losses, idx, S


def train_one_epoch(model, tokenizer, data_loader_text, data_loader_image, optimizer, device, args):
    """reference train_TTA.py:98-153.  Returns (score_matrix_i2t, score_matrix_t2i)."""
    clip = model.clip if hasattr(model, "clip") else model
    len_image = len(data_loader_image.dataset.image)
    len_text = len(data_loader_text.dataset.text)
    world, rank = (get_world_size(), get_rank()) if getattr(args, "distributed", False) else (1, 0)

    score_matrix_i2t = torch.full((len_image, len_text), -100.0, device=device)
    score_matrix_t2i = torch.full((len_text, len_image), -100.0, device=device)
    model.eval()
    if getattr(args, "tta_only_visual", False):
        return score_matrix_i2t, score_matrix_t2i      # reference: `pass` (train_TTA.py:113-114)

    D = clip.text_projection.shape[1]
    all_image_embeds = torch.zeros(len_image, D, device=device)
    with torch.no_grad():
        for image, index in data_loader_image:
            image = image.to(device, non_blocking=True)
            all_image_embeds[index.to(device)] = model.encode_image(image, normalize=True)
    T1, T2 = args.tta_temperature, args.tta_temperature2
    # The host never waits for the batch it has just enqueued: the four token sets go to the device as ONE pinned,
    # non-blocking copy (a pageable copy is ordered behind the previous batch's kernels and blocks the host until they
    # are done), and a batch's losses are logged after the NEXT batch has been enqueued (same lines, same order as the
    # reference's per-batch `.item()` at train_TTA.py:148).  Tokenising batch i+1 then runs under the GPU time of batch i.
    def _h2d(x):
        if x.is_cuda or torch.device(device).type != "cuda":
            return x.to(device)
        return x.pin_memory().to(device, non_blocking=True)

    def _snapshot(losses):
        """Start the device-to-host copy of a batch's losses right behind that batch's kernels (pinned, non-blocking)
        and mark it with an event: reading it later waits for THAT batch only.  (A plain ``.tolist()`` one batch later
        would be ordered behind the batch enqueued in between.)"""
        if not losses.is_cuda:
            return losses, None
        host = torch.empty(losses.shape, dtype=losses.dtype, pin_memory=True)
        host.copy_(losses, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return host, done

    def _log(snap):
        host, done = snap
        if done is not None:
            done.synchronize()
        le, ls, lt = host.tolist()
        logging.info(f"\tloss_entropy: {le:.3f}, loss_sr: {ls:.3f}, loss_tri: {lt:.3f}")

    pending = None
    for text, pos_text, neg_text, inv_text, idx in data_loader_text:
        toks = [tokenizer(x) for x in (text, pos_text, neg_text, inv_text)]
        toks = [x if x.dtype == torch.int64 else x.long() for x in toks]
        if all(not x.is_cuda for x in toks) and len({tuple(x.shape) for x in toks}) == 1:
            toks = list(_h2d(torch.stack(toks)).unbind(0))
        else:
            toks = [_h2d(x) for x in toks]
        Bt = toks[0].shape[0]
        lo, hi = _shard(Bt, world, rank)
        local = [x[lo:hi].contiguous() for x in toks]
        for _ in range(args.tta_steps):
            losses, _, sims = tta_step(clip, optimizer, local, all_image_embeds, T1, T2, world, rank, global_rows=Bt)
        if world > 1:
            sims = _all_gather_rows(sims, Bt, world)
        score_matrix_t2i[_h2d(idx)] = sims
        if pending is not None:
            _log(pending)
        pending = _snapshot(losses)
    if pending is not None:
        _log(pending)
    return score_matrix_i2t, score_matrix_t2i
