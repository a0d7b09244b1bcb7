"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Run from the repo root:  python tests/golden/make_golden.py [case ...]

The reference (/root/reference, read-only, never copied) is imported with import-time stubs for the
two packages it needs only for host-side preprocessing (torchvision, ftfy).  It is
fed the seeded synthetic weights / tokens / images of ``neat_amd.synth`` and its *unmodified*
``train_TTA.train_one_epoch`` is executed; per-step intermediates are captured by wrapping the
module-level helpers it calls (``coarse_aligning_t2i``, ``softmax_entropy``,
``semantics_reversion_loss``, ``textual_triplet_loss``) and the optimizer's ``step``.

Only inputs' seeds and the reference's OUTPUTS are stored (small .npz files).  This script is the
only file in the repository that touches /root/reference; it cannot run on the GPU box.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neat_amd import synth  # noqa: E402

REF = "/root/reference"


def import_reference():
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _D:
        BICUBIC = "bicubic"; BILINEAR = "bilinear"; NEAREST = "nearest"

        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv = stub("torchvision")
    ops = stub("torchvision.ops")
    misc = stub("torchvision.ops.misc", FrozenBatchNorm2d=_D)
    ops.misc = misc
    tv.ops = ops
    names = ["Normalize", "Compose", "RandomResizedCrop", "InterpolationMode", "ToTensor", "Resize",
             "CenterCrop", "ColorJitter", "Grayscale"]
    tr = stub("torchvision.transforms", **{n: _D for n in names})
    trf = stub("torchvision.transforms.functional")
    tr.transforms = tr
    tr.functional = trf
    tv.transforms = tr
    stub("ftfy", fix_text=lambda s: s)
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    import src.open_clip as oc
    import train_TTA
    import src.evaluation.retrieval as retrieval
    return oc, train_TTA, retrieval


def build_ref_model(oc, name, seed, spread=False):
    if name == "neat-tiny":
        d = tempfile.mkdtemp()
        with open(os.path.join(d, "neat-tiny.json"), "w") as f:
            json.dump(synth.MODEL_CFGS["neat-tiny"], f)
        oc.add_model_config(d)
    model = oc.create_model(name, pretrained=None, precision="fp32", device="cpu", force_quick_gelu=True)
    sd = synth.spread_state_dict(name, seed) if spread else synth.synth_state_dict(name, seed)
    missing = model.load_state_dict(sd, strict=True)   # proves the key layout matches open_clip's
    assert not missing.missing_keys and not missing.unexpected_keys
    return model, sd


class _DS:
    pass


class FakeLoader:
    """Just enough of a DataLoader for train_one_epoch (train_TTA.py:105-106,121,125)."""

    def __init__(self, batches, n, attr):
        self.batches = batches
        self.dataset = _DS()
        setattr(self.dataset, attr, list(range(n)))

    def __iter__(self):
        return iter(self.batches)


def make_batches(cfg, n_text, n_image, bs_text, bs_image, data_seed, related=False):
    t = cfg["text_cfg"]
    sets = (synth.synth_related_caption_sets if related else synth.synth_caption_sets)(
        n_text, t["context_length"], t["vocab_size"], data_seed)
    images = synth.synth_images(n_image, cfg["vision_cfg"]["image_size"], data_seed + 1)
    g = torch.Generator().manual_seed(data_seed + 2)
    perm_t = torch.randperm(n_text, generator=g)       # reference loader shuffles (data/__init__.py:60)
    perm_i = torch.randperm(n_image, generator=g)
    tb = [tuple(s[perm_t[i:i + bs_text]] for s in sets) + (perm_t[i:i + bs_text],)
          for i in range(0, n_text, bs_text)]
    ib = [(images[perm_i[i:i + bs_image]], perm_i[i:i + bs_image]) for i in range(0, n_image, bs_image)]
    return tb, ib


class _TorchWithGalleryWidth:
    """``torch`` as train_TTA sees it, with ONE change: the gallery buffer the reference hard-codes to 512 columns
    (train_TTA.py:119) is made ``width`` columns wide, so that a D = 768 model (ViT-L-14) can run at all
    (SURVEY.md §0).  Everything else is forwarded to torch untouched."""

    def __init__(self, width):
        self._w = width

    def __getattr__(self, name):
        return getattr(torch, name)

    def zeros(self, *size, **kw):
        if len(size) == 2 and size[1] == 512:
            size = (size[0], self._w)
        return torch.zeros(*size, **kw)


def run_instrumented(oc, tt, model, tb, ib, n_text, n_image, steps=3, lr=2e-4, wd=1e-3, T1=0.03, T2=0.07,
                     keep_rows=8, gallery_width=None):
    """Unmodified train_one_epoch with capture hooks."""
    oc.freeze_parameters(model, False)
    params, names = oc.collect_params(model, False)
    names = ["transformer." + n for n in names]
    rec = dict(ent=[], sr=[], tri=[], idx=[], feats=[], grads=None, params_after=[])
    o_coarse, o_ent, o_sr, o_tri = tt.coarse_aligning_t2i, tt.softmax_entropy, tt.semantics_reversion_loss, \
        tt.textual_triplet_loss

    def coarse(q, p, n, G):
        out = o_coarse(q, p, n, G)
        rec["idx"].append(out[1].clone())
        if len(rec["feats"]) == 0:
            rec["feats"] = [x.detach()[:keep_rows].clone() for x in (q, p, n)]
            rec["gallery"] = G[:keep_rows].clone()
        return out

    def ent(x):
        r = o_ent(x)
        rec["ent"].append(r.detach().mean(0).item())
        return r

    def sr(s, tau=1):
        r = o_sr(s, tau)
        rec["sr"].append(r.item())
        return r

    def tri(a, b, c):
        r = o_tri(a, b, c)
        rec["tri"].append(r.item())
        if len(rec["feats"]) == 3:
            rec["feats"].append(c.detach()[:keep_rows].clone())
        return r

    class Opt(torch.optim.AdamW):
        def step(self, closure=None):
            if rec["grads"] is None:
                rec["grads"] = [p.grad.detach().clone() for p in params]
            r = super().step(closure)
            if len(rec["params_after"]) < steps:
                rec["params_after"].append(torch.cat([p.detach().flatten() for p in params]).clone())
            return r

    tt.coarse_aligning_t2i, tt.softmax_entropy, tt.semantics_reversion_loss, tt.textual_triplet_loss = \
        coarse, ent, sr, tri
    o_torch = tt.torch
    if gallery_width is not None and gallery_width != 512:
        tt.torch = _TorchWithGalleryWidth(gallery_width)
    try:
        opt = Opt(params, lr=lr, weight_decay=wd)       # main_TTA.py:271
        args = types.SimpleNamespace(distributed=False, tta_only_visual=False, name="neat", tta_steps=steps,
                                     tta_temperature=T1, tta_temperature2=T2)
        t0 = time.time()
        _, score = tt.train_one_epoch(model, lambda x: x, FakeLoader(tb, n_text, "text"),
                                      FakeLoader(ib, n_image, "image"), opt, torch.device("cpu"), args)
        rec["seconds"] = time.time() - t0
    finally:
        tt.torch = o_torch
        tt.coarse_aligning_t2i, tt.softmax_entropy, tt.semantics_reversion_loss, tt.textual_triplet_loss = \
            o_coarse, o_ent, o_sr, o_tri
    rec["score"] = score
    rec["names"] = names
    rec["final_params"] = torch.cat([p.detach().flatten() for p in params])
    return rec


def weight_checksums(sd):
    keys = ["token_embedding.weight", "transformer.resblocks.0.attn.in_proj_weight", "visual.conv1.weight",
            "text_projection"]
    return np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()] for k in keys])


def case_tta(oc, tt, retrieval, tag, model_name, n_text, n_image, bs_text, bs_image, wseed=0, dseed=1234,
             store_score=True, store_grads=True, spread=False):
    cfg = synth.get_cfg(model_name)
    model, sd = build_ref_model(oc, model_name, wseed, spread=spread)
    tb, ib = make_batches(cfg, n_text, n_image, bs_text, bs_image, dseed, related=spread)
    with torch.no_grad():
        img_rows = model.encode_image(ib[0][0][:8], normalize=True).clone()
        txt_unnorm = model.encode_text(tb[0][0][:8], normalize=False).clone()
    rec = run_instrumented(oc, tt, model, tb, ib, n_text, n_image, gallery_width=cfg["embed_dim"])
    out = dict(
        meta=np.array(json.dumps(dict(model=model_name, n_text=n_text, n_image=n_image, bs_text=bs_text,
                                      bs_image=bs_image, wseed=wseed, dseed=dseed, steps=3, lr=2e-4, wd=1e-3,
                                      T1=0.03, T2=0.07, names=rec["names"], seconds=rec["seconds"], spread=spread,
                                      torch=torch.__version__))),
        wsum=weight_checksums(sd),
        loss_ent=np.array(rec["ent"]), loss_sr=np.array(rec["sr"]), loss_tri=np.array(rec["tri"]),
        coarse_idx=torch.stack([torch.nn.functional.pad(i, (0, bs_text - len(i)), value=-1)
                                for i in rec["idx"]]).numpy().astype(np.int32),
        feat_text=rec["feats"][0].numpy(), feat_pos=rec["feats"][1].numpy(), feat_neg=rec["feats"][2].numpy(),
        feat_inv=rec["feats"][3].numpy(), gallery_rows=rec["gallery"].numpy(),
        img_rows=img_rows.numpy(), txt_unnorm=txt_unnorm.numpy(),
        params_after=torch.stack(rec["params_after"]).numpy(),
        final_params=rec["final_params"].numpy(),
    )
    if store_grads:
        out["grads_step1"] = torch.cat([g.flatten() for g in rec["grads"]]).numpy()
    score = rec["score"]
    if store_score:
        out["score"] = score.numpy()
    else:
        out["score_rows"] = score[:8].numpy()
        out["score_sum"] = np.array([score.double().sum().item(), score.double().abs().sum().item()])
    # recall on the produced matrix: caption r <-> image r % n_image (synthetic ground truth)
    pos = torch.zeros_like(score, dtype=torch.bool)
    pos[torch.arange(n_text), torch.arange(n_text) % n_image] = True
    ks = [k for k in (1, 5, 10) if k <= min(n_image, n_text)]
    rec_t2i = [(retrieval.batchify(retrieval.recall_at_k, score, pos, 64, "cpu", k=k) > 0).float().mean().item()
               for k in ks]
    rec_i2t = [(retrieval.batchify(retrieval.recall_at_k, score.T, pos.T, 64, "cpu", k=k) > 0).float().mean().item()
               for k in ks]
    out["recall_ks"] = np.array(ks)
    out["recall_t2i"] = np.array(rec_t2i)
    out["recall_i2t"] = np.array(rec_i2t)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print(tag, "losses", rec["ent"][-1], rec["sr"][-1], rec["tri"][-1], "sec", rec["seconds"])


def case_recall(retrieval):
    """recall_at_k / batchify / median_rank known answers incl. deliberate ties."""
    g = torch.Generator().manual_seed(7)
    n_img, per = 200, 5
    n_txt = n_img * per
    scores = torch.randn(n_txt, n_img, generator=g)
    owner = torch.arange(n_txt) // per
    scores[torch.arange(n_txt), owner] += 1.5          # make positives likely but not certain
    pos = torch.zeros(n_txt, n_img, dtype=torch.bool)
    pos[torch.arange(n_txt), owner] = True
    out = dict(scores=scores.numpy(), owner=owner.numpy().astype(np.int32))
    for k in (1, 5, 10):
        out[f"t2i_{k}"] = retrieval.batchify(retrieval.recall_at_k, scores, pos, 64, "cpu", k=k).numpy()
        out[f"i2t_{k}"] = retrieval.batchify(retrieval.recall_at_k, scores.T, pos.T, 64, "cpu", k=k).numpy()
        out[f"topk_t2i_{k}"] = torch.topk(scores, k, dim=1)[1].numpy().astype(np.int32)
    out["medr_t2i"] = np.array(retrieval.median_rank(scores, pos))
    out["medr_i2t"] = np.array(retrieval.median_rank(scores.T, pos.T))
    # tie cases: quantised scores -> many exact ties; record what torch.topk (the reference's
    # selection, retrieval.py:295) returns so the tie rule of the HIP kernel can be compared
    tie = torch.randint(0, 6, (64, 50), generator=g).float()
    tpos = torch.zeros(64, 50, dtype=torch.bool)
    tpos[torch.arange(64), torch.randint(0, 50, (64,), generator=g)] = True
    out["tie_scores"] = tie.numpy()
    out["tie_pos"] = tpos.numpy()
    for k in (1, 5, 10):
        out[f"tie_topk_{k}"] = torch.topk(tie, k, dim=1)[1].numpy().astype(np.int32)
        out[f"tie_recall_{k}"] = retrieval.recall_at_k(tie, tpos, k).numpy()
    out["tie_argmax"] = tie.argmax(dim=1).numpy().astype(np.int32)
    m, j = tie.max(1)
    out["tie_max_idx"] = j.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "recall.npz"), **out)
    print("recall done", out["medr_t2i"], out["medr_i2t"])


def case_loss_ties(tt):
    """Loss pieces on hand-built inputs where several captions share one coarse image (exact ties in
    the SR hinge, train_TTA.py:53-65) -- values and autograd gradients from the reference functions."""
    g = torch.Generator().manual_seed(11)
    B, D, N = 24, 32, 6
    norm = torch.nn.functional.normalize
    G = norm(torch.randn(N, D, generator=g), dim=-1)
    t = norm(torch.randn(B, D, generator=g), dim=-1).requires_grad_()
    p = norm(torch.randn(B, D, generator=g), dim=-1).requires_grad_()
    n = norm(torch.randn(B, D, generator=g), dim=-1)
    v = norm(torch.randn(B, D, generator=g), dim=-1).requires_grad_()
    coarse, idx, _, _, S = tt.coarse_aligning_t2i(t.detach(), p.detach(), n, G)
    outputs = t @ coarse.t()
    aux = v @ coarse.t()
    le = tt.softmax_entropy(outputs.t() / 0.03).mean(0)
    ls = tt.semantics_reversion_loss(aux, 0.07)
    lt = tt.textual_triplet_loss(t, p, v)
    (le + ls + lt).backward()
    np.savez_compressed(os.path.join(HERE, "loss_ties.npz"), G=G.numpy(), t=t.detach().numpy(),
                        p=p.detach().numpy(), n=n.numpy(), v=v.detach().numpy(), idx=idx.numpy().astype(np.int32),
                        S=S.numpy(), losses=np.array([le.item(), ls.item(), lt.item()]),
                        dt=t.grad.numpy(), dp=p.grad.numpy(), dv=v.grad.numpy())
    print("loss_ties", le.item(), ls.item(), lt.item(), "distinct coarse", len(set(idx.tolist())))


def case_coarse_i2t(tt):
    """The reference's image-to-text coarse alignment (train_TTA.py:20-28, defined there and never called) on seeded unit
    features: B = 70 image queries against N = 333 caption rows (ragged against the kernels' 64-wide tiles), D = 64."""
    g = torch.Generator().manual_seed(23)
    B, N, D = 70, 333, 64
    norm = torch.nn.functional.normalize
    q = norm(torch.randn(B, D, generator=g), dim=-1)
    G = norm(torch.randn(N, D, generator=g), dim=-1)
    Gp = norm(G + 0.5 * torch.randn(N, D, generator=g), dim=-1)
    Gn = norm(torch.randn(N, D, generator=g), dim=-1)
    out = tt.coarse_aligning_i2t(q, G, Gp, Gn)
    A = (q @ Gp.t()) * (1 - torch.sigmoid(q @ Gn.t()))
    idx = A.argmax(dim=1)
    assert torch.equal(out, G[idx])                               # the function returns the gathered rows only
    top2 = A.topk(2, dim=1).values
    np.savez_compressed(os.path.join(HERE, "coarse_i2t.npz"), q=q.numpy(), G=G.numpy(), Gp=Gp.numpy(), Gn=Gn.numpy(),
                        out=out.numpy(), idx=idx.numpy().astype(np.int32), margin=(top2[:, 0] - top2[:, 1]).numpy())
    print("coarse_i2t distinct", len(set(idx.tolist())), "min margin", float((top2[:, 0] - top2[:, 1]).min()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["recall", "loss_ties", "tiny", "vitb32_b32"])
    a = ap.parse_args()
    logging.basicConfig(level=logging.WARNING)
    torch.manual_seed(0)
    oc, tt, retrieval = import_reference()
    for c in a.cases:
        if c == "recall":
            case_recall(retrieval)
        elif c == "loss_ties":
            case_loss_ties(tt)
        elif c == "coarse_i2t":
            case_coarse_i2t(tt)
        elif c == "tiny":
            case_tta(oc, tt, retrieval, "tiny", "neat-tiny", n_text=40, n_image=24, bs_text=16, bs_image=10)
        elif c == "vitb32_b32":
            case_tta(oc, tt, retrieval, "vitb32_b32", "ViT-B-32", n_text=48, n_image=64, bs_text=32, bs_image=32,
                     store_grads=True)
        elif c == "vitb32_b256":
            case_tta(oc, tt, retrieval, "vitb32_b256", "ViT-B-32", n_text=256, n_image=256, bs_text=256,
                     bs_image=64, store_score=False, store_grads=True)
        elif c == "vitl14_b128":
            # config 3's model (ViT-L-14: text width 768, D = 768) at a batch the reference's autograd fits in this
            # container's memory; the only change to the reference is the gallery-buffer width (see above)
            case_tta(oc, tt, retrieval, "vitl14_b128", "ViT-L-14", n_text=128, n_image=32, bs_text=128,
                     bs_image=8, store_score=False, store_grads=True)
        elif c == "vitb32_spread_b256":
            # the "spread" weights + related caption sets (tests/golden/make_spread_weights.py, same data seed): features
            # far from collinear, recall@1 ~0.7 -- the non-degenerate end-to-end fixture
            case_tta(oc, tt, retrieval, "vitb32_spread_b256", "ViT-B-32", n_text=256, n_image=256, bs_text=256,
                     bs_image=64, dseed=4321, store_score=True, store_grads=True, spread=True)
        else:
            raise SystemExit(f"unknown case {c}")


if __name__ == "__main__":
    main()
