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
  * ``model.pack_tokens`` (default True; ``NEAT_PACK_TOKENS=0`` or ``model.pack_tokens = False`` turns it off): the
    text tower runs only positions 0..EOT of every caption (dead rows behind the causal mask are dropped; same
    features, losses and LN gradients up to fp32 summation order);
  * data parallelism: a rank owns a contiguous row shard of every caption batch and a share of the
    gallery; it all-gathers text features + coarse image features, sums the LN gradients of all ranks, and gathers the
    gallery embeddings and the score rows.  Both loader wirings are accepted, per loader: ``DistributedSampler``s as the
    reference's launcher hands them out (main_TTA.py:245-257: every rank sees its own rows, batch = tta_total_bs //
    world), or replicated loaders (every rank iterates the same batches and cuts its shard out of each).
"""
from __future__ import annotations

import logging
from typing import Optional

import os

import numpy as np
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


@torch.no_grad()
def coarse_aligning_i2t(query_feat, gallery_feat_all, pos_gallery_feat_all, neg_gallery_feat_all):
    """train_TTA.py:20-28: the image-to-text direction (image queries against caption / positive-caption / negated-caption
    galleries, sigmoid instead of relu).  The reference defines this helper and never calls it (its ``tta_only_visual``
    branch is ``pass``, :113-114); it is provided with the reference's signature and return value (the gathered rows)."""
    return ops.coarse_align_i2t(query_feat.detach().contiguous(), gallery_feat_all.contiguous(),
                                pos_gallery_feat_all.contiguous(), neg_gallery_feat_all.contiguous())[0]


def tta_loss(text_feat, pos_text_feat, inv_text_feat, coarse_img_feat, T1, T2):
    """loss_entropy, loss_sr, loss_tri (train_TTA.py:138-143) and their gradients, no autograd."""
    return ops.tta_loss_fwd_bwd(text_feat, pos_text_feat, inv_text_feat, text_feat, coarse_img_feat, T1, T2, 0)


# NEAT_FORCE_COLLECTIVES=1: run every collective of the data-parallel step even at world size 1 (a one-GPU rehearsal of the
# RCCL call path: process group, all-gather of padded row shards, all-gather + ordered sum of the flat LN gradient)
FORCE_COLLECTIVES = os.environ.get("NEAT_FORCE_COLLECTIVES", "") == "1"


def _shard(n: int, world: int, rank: int):
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged(x) -> bool:
    """gloo moves GPU tensors through the host by itself, and with several ranks on one card that path has hung on this
    stack (ROCm 7, torch 2.10): under gloo the collectives below stage GPU tensors through host memory explicitly.  RCCL
    ("nccl"), the backend of a real multi-GPU run, takes the device tensors as they are."""
    return x.is_cuda and dist.get_backend() == "gloo"


def _own_comm(x) -> bool:
    """NEAT_COMM=rccl: GPU collectives go through the library's own RCCL communicator (C ABI neat_comm_*), created on
    first use from the existing process group; everything else (CPU tensors, non-float reductions) stays on
    torch.distributed."""
    from . import comm
    if not (comm.ENABLED and x.is_cuda):
        return False
    if not comm.ready():
        comm.init_from_process_group()
    return True


def _gather_into(out, x):
    """dist.all_gather_into_tensor(out, x) (rank order along dim 0)."""
    if _own_comm(x):
        from . import comm
        comm.all_gather_into(out, x)
    elif _host_staged(x):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, x.contiguous().cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, x.contiguous())


def _all_gather_rows(x, n_total: int, world: int):
    """Concatenate row shards of (possibly) different sizes from all ranks, in rank order."""
    sizes = [_shard(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == mx for lo, hi in sizes):                    # even shards: one output tensor, no padding, no cat
        out = torch.empty(world * mx, *x.shape[1:], dtype=x.dtype, device=x.device)
        _gather_into(out, x)
        return out
    pad = torch.zeros(mx, *x.shape[1:], dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = torch.empty(world * mx, *x.shape[1:], dtype=x.dtype, device=x.device)
    _gather_into(out, pad)
    return torch.cat([out[r * mx: r * mx + hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def _all_gather_ragged(x, world: int):
    """Rows of every rank (any row counts, zero included) -> (list of per-rank tensors, in rank order).  One small
    host-synchronising collective for the counts, one padded all-gather for the data: used once per epoch (gallery)."""
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    counts = torch.empty(world, dtype=torch.int64, device=x.device)
    _gather_into(counts, n)
    counts = [int(c) for c in counts.tolist()]
    mx = max(max(counts), 1)
    pad = torch.zeros(mx, *x.shape[1:], dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = torch.empty(world * mx, *x.shape[1:], dtype=x.dtype, device=x.device)
    _gather_into(out, pad)
    return [out[r * mx: r * mx + c] for r, c in enumerate(counts)]


def _distributed_sampler(loader):
    """The DistributedSampler behind a loader (a DataLoader, or one of the GPU loaders that wrap one), or None."""
    from torch.utils.data.distributed import DistributedSampler
    seen = 0
    while loader is not None and seen < 4:
        for attr in ("sampler", "batch_sampler"):
            smp = getattr(loader, attr, None)
            smp = getattr(smp, "sampler", smp) if attr == "batch_sampler" else smp
            if isinstance(smp, DistributedSampler):
                return smp
        loader = getattr(loader, "loader", None)
        seen += 1
    return None


def _loader_is_sharded(loader, world: int, rank: int, what: str) -> bool:
    smp = _distributed_sampler(loader)
    if smp is None or world == 1:
        if smp is not None and smp.num_replicas > 1:
            raise RuntimeError(f"train_one_epoch: the {what} loader has a DistributedSampler over {smp.num_replicas} "
                               f"replicas but args.distributed is off: rows of the other replicas would never be seen")
        return False
    if smp.num_replicas != world or smp.rank != rank:
        raise RuntimeError(f"train_one_epoch: the {what} loader's DistributedSampler is (replicas {smp.num_replicas}, "
                           f"rank {smp.rank}) but the process group is (world {world}, rank {rank})")
    return smp.num_replicas > 1


class GalleryEncoder:
    """Phase A (train_TTA.py:120-124) with the loader's batches GROUPED before they are encoded.  The reference encodes
    one loader batch at a time (tta_total_bs = 256 images, main_TTA.py:256); images are independent, and at ViT-B/32
    a 256-image batch is 12 800 token rows -- less than one round of GEMM units on this chip (a layer's four GEMMs take
    1.36 us per image at 250 images and 0.89 us at 1 000, tools/gemm_bench.py).  Batches are therefore collected until
    ``target_rows`` token rows are at hand (or the loader ends) and encoded together; per image the result is what
    ``model.encode_image(batch, normalize=True)`` gives."""

    def __init__(self, model, device, target_rows: int = 50000):
        self.model, self.device, self.target = model, device, target_rows
        self.images, self.index, self.rows = [], [], 0

    def _tokens_per_image(self, image):
        clip = self.model.clip if hasattr(self.model, "clip") else self.model
        g = getattr(getattr(clip, "visual", None), "grid_size", None)
        per = (g[0] * g[1] + 1) if g else 50
        return per * (image.shape[1] if image.dim() == 5 else 1)          # video clips: frames fold into the batch

    def add(self, image, index):
        """-> list of (index tensor on the device, embeddings) that became ready (possibly empty)."""
        image = image.to(self.device, non_blocking=True)
        self.images.append(image)
        self.index.append(index.to(self.device).view(-1))
        self.rows += image.shape[0] * self._tokens_per_image(image)
        return self.flush() if self.rows >= self.target else []

    def flush(self):
        if not self.images:
            return []
        image = self.images[0] if len(self.images) == 1 else torch.cat(self.images, dim=0)
        index = self.index[0] if len(self.index) == 1 else torch.cat(self.index)
        self.images, self.index, self.rows = [], [], 0
        with torch.no_grad():
            return [(index, self.model.encode_image(image, normalize=True))]


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


def host_cu_seqlens(host_tokens4):
    """The packed layout from tokens still on the HOST: int32 [4*Bl + 1] prefix sums of (EOT position + 1) over the rows of
    [text, pos, inv, neg]; EOT = the first maximum of a row (reference open_clip/transformer.py:555-558, the same rule as
    the library's neat_pack_plan; ``numpy.argmax`` returns the first maximum)."""
    text, pos, neg, inv = host_tokens4
    cat = torch.cat([text, pos, inv, neg], dim=0).numpy()
    cu = np.zeros(cat.shape[0] + 1, dtype=np.int32)
    np.cumsum(cat.argmax(axis=1) + 1, out=cu[1:])
    return torch.from_numpy(cu)


def batch_plan(model, tokens4, host_tokens4=None):
    """What the tta_steps of ONE adaptation batch share: the four token sets as one [4*Bl, L] batch ordered
    [text, pos, inv, neg] and, with packed text batches, its packed layout.  Pass the result to every
    ``tta_step(..., plan=)`` of the batch.

    The layout (cu_seqlens) is needed on the device AND on the host (its totals size the step's buffers).  When the
    caller still holds the tokens on the host (``host_tokens4``: the same rows as ``tokens4``, CPU int64 -- what
    ``train_one_epoch`` has right after tokenising), it is computed there and sent to the device with a pinned,
    non-blocking copy: no device-to-host copy, so the host does not wait for the previous batch's kernels.  Otherwise the
    library computes it from the device tokens and one small device-to-host copy brings it back."""
    text, pos, neg, inv = tokens4
    toks = torch.cat([text, pos, inv, neg], dim=0)
    if bool(getattr(model, "pack_tokens", False)) and toks.is_cuda:
        if host_tokens4 is not None:
            cu_host = host_cu_seqlens(host_tokens4)
            if cu_host.numel() != toks.shape[0] + 1:
                raise RuntimeError("batch_plan: host_tokens4 does not hold the rows of tokens4")
            return toks.contiguous(), cu_host.pin_memory().to(toks.device, non_blocking=True), cu_host
        return model._text_tower().plan(toks)
    return (toks, None, None)


def tta_step(model, optimizer, tokens4, gallery, T1, T2, world=1, rank=0, global_rows=None, plan=None):
    """One inner step (train_TTA.py:133-147) on this rank's row shard.  tokens4 = (text,pos,neg,inv) int64
    [Bl, L] on the GPU.  Returns (losses[3] tensor, coarse idx, sims_matrix) for the local rows.
    global_rows: rows of the global batch when the caller knows them (shards cut by ``_shard``); without it the
    ranks add up their row counts, which costs a collective and a host synchronisation per step."""
    text, pos, neg, inv = tokens4
    Bl = text.shape[0]
    tower = model._text_tower()
    packed = bool(getattr(model, "pack_tokens", False))
    toks = plan[0] if plan is not None else torch.cat([text, pos, inv, neg], dim=0)
    kw = dict(plan=plan) if (packed and plan is not None and plan[1] is not None) else {}
    feats, ctx = tower.forward(toks, True, True, packed=packed, n_grad=3 * Bl, **kw)                 # neg: no backward
    t, p, v, n = feats[:Bl], feats[Bl:2 * Bl], feats[2 * Bl:3 * Bl], feats[3 * Bl:]
    coarse, idx, S = ops.coarse_align(t, p, n, gallery)
    if world > 1 or FORCE_COLLECTIVES:
        if global_rows is None:
            counts = torch.empty(world, dtype=torch.int64, device=t.device)
            _gather_into(counts, torch.tensor([Bl], dtype=torch.int64, device=t.device))
            B = int(counts.sum().item())
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
            _gather_into(gathered.view(-1), buf)
            torch.sum(gathered, dim=0, out=buf)
            losses[1:].copy_(buf[-2:])
        else:
            # any other optimizer over the same parameters (torch.optim.AdamW ...): the per-parameter .grad tensors and
            # the two row-summed loss terms are copied into one flat scratch and take the SAME route -- one all-gather,
            # a sum in rank order, copies back.  No all_reduce on the step (49 of them per step before; what
            # GatherLayer's all_reduce in the reference's unused src/ddp.py:29-56 would have done per feature tensor).
            flat = [x for g in grads for x in g.values()]
            sizes = [x.numel() for x in flat]
            buf = torch.cat([x.reshape(-1) for x in flat] + [losses[1:].reshape(-1)])
            gathered = torch.empty(world, buf.numel(), dtype=buf.dtype, device=buf.device)
            _gather_into(gathered.view(-1), buf)
            total = torch.sum(gathered, dim=0)
            for x, piece in zip(flat, torch.split(total[:-2], sizes)):
                x.copy_(piece.view_as(x))
            losses[1:].copy_(total[-2:])          # sr / triplet are sums over local rows; entropy is already global
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

    if world > 1 or FORCE_COLLECTIVES:
        from . import comm
        if comm.ENABLED and torch.device(device).type == "cuda" and not comm.ready():
            comm.init_from_process_group()        # here, not inside the first gather: all ranks fail or go on together
    # which rows does this rank see?  (raises on a sampler that does not match the process group)
    text_sharded = _loader_is_sharded(data_loader_text, world, rank, "text")
    image_sharded = _loader_is_sharded(data_loader_image, world, rank, "image")

    D = clip.text_projection.shape[1]
    all_image_embeds = torch.zeros(len_image, D, device=device)
    # Phase A (train_TTA.py:120-124).  With W ranks every rank encodes 1/W of the gallery -- its sampler's images, or
    # its row shard of every replicated batch -- and ONE ragged all-gather at the end replicates the embeddings.
    with torch.no_grad():
        mine_idx, mine_emb = [], []
        enc = GalleryEncoder(model, device)

        def _place(ready):
            for index, emb in ready:
                if world == 1:
                    all_image_embeds[index] = emb
                else:
                    mine_idx.append(index)
                    mine_emb.append(emb)

        for image, index in data_loader_image:
            if world > 1 and not image_sharded:
                lo, hi = _shard(len(index), world, rank)
                if hi == lo:
                    continue
                image, index = image[lo:hi], index[lo:hi]
            _place(enc.add(image, index))
        _place(enc.flush())
        if world > 1:
            idx_l = torch.cat(mine_idx) if mine_idx else torch.zeros(0, dtype=torch.int64, device=device)
            emb_l = torch.cat(mine_emb) if mine_emb else torch.zeros(0, D, device=device)
            packed = torch.cat([emb_l, idx_l.to(emb_l.dtype).view(-1, 1)], dim=1)     # indices < 2^24: exact in fp32
            if len_image >= (1 << 24):
                raise RuntimeError("train_one_epoch: gallery too large for the fp32 index column of the gather")
            # rank order, so that an image a padded sampler hands to two ranks is written the same way everywhere
            for part in _all_gather_ragged(packed, world):
                all_image_embeds[part[:, D].long()] = part[:, :D]
    T1, T2 = args.tta_temperature, args.tta_temperature2
    # The host never waits for the batch it has just enqueued: the four token sets go to the device as ONE pinned,
    # non-blocking copy (a pageable copy is ordered behind the previous batch's kernels and blocks the host until they
    # are done), and a batch's losses are logged after the NEXT batch has been enqueued (same lines, same order as the
    # reference's per-batch `.item()` at train_TTA.py:148).  Tokenising batch i+1 then runs under the GPU time of batch i.
    # With packed text batches the layout of a batch is computed from the HOST tokens (batch_plan) for the same reason: the
    # library's plan needs a device-to-host copy, which is ordered behind everything enqueued before it.
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
        host_toks = toks if all(not x.is_cuda for x in toks) else None       # kept for the packed plan (batch_plan)
        if host_toks is not None and len({tuple(x.shape) for x in toks}) == 1:
            toks = list(_h2d(torch.stack(toks)).unbind(0))
        else:
            toks = [_h2d(x) for x in toks]
        idx = _h2d(idx).view(-1)
        if text_sharded:
            # the sampler already cut the global batch: these ARE the local rows.  A DistributedSampler pads its index
            # list, so every rank holds the same number of rows in every batch and rank r owns rows [r*Bl, (r+1)*Bl).
            local, Bl = toks, toks[0].shape[0]
            Bt = Bl * world
            idx = _all_gather_rows(idx.view(-1, 1), Bt, world).view(-1)
        else:
            Bt = toks[0].shape[0]
            if Bt < world:
                # the same decision on every rank, before any collective: a rank with no rows would fail alone and
                # leave the others waiting in the all-gather
                raise RuntimeError(f"train_one_epoch: a batch of {Bt} caption rows cannot be cut over {world} ranks")
            lo, hi = _shard(Bt, world, rank)
            local = [x[lo:hi].contiguous() for x in toks]
            if host_toks is not None:
                host_toks = [x[lo:hi] for x in host_toks]
        plan = batch_plan(clip, local, host_toks)
        for _ in range(args.tta_steps):
            losses, _, sims = tta_step(clip, optimizer, local, all_image_embeds, T1, T2, world, rank, global_rows=Bt,
                                       plan=plan)
        if world > 1:
            sims = _all_gather_rows(sims, Bt, world)
        score_matrix_t2i[idx] = sims
        if pending is not None:
            _log(pending)
        pending = _snapshot(losses)
    if pending is not None:
        _log(pending)
    return score_matrix_i2t, score_matrix_t2i
