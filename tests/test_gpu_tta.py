"""GPU: the whole TTA path (encoders -> coarse alignment -> loss+grad -> backward -> AdamW -> score rows ->
recall@k) on HIP kernels against (a) vectors produced by the REFERENCE (tests/golden/*.npz) and (b) the
CPU oracle run here on the same seeded inputs.

Gates (BASELINE.json north_star): embeddings cosine >= 0.999; each loss term within 1e-3 relative; adapted
LayerNorm parameters within 1e-3 relative (L2 over the adapted vector); top-k indices bit-exact given the
same scores.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers import load_golden, make_batches, min_cosine, rel_l2, weight_checksums

pytestmark = pytest.mark.gpu

from neat_amd import open_clip as oc, ops, synth  # noqa: E402
from neat_amd import train_TTA as T  # noqa: E402
from neat_amd.evaluation import retrieval as R  # noqa: E402
from neat_amd.optim import AdamW  # noqa: E402
from oracle import neat_oracle as O  # noqa: E402

DEV = "cuda"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.json")


def _report(tag, d):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    allr = {}
    if os.path.exists(REPORT):
        try:
            allr = json.load(open(REPORT))
        except Exception:
            allr = {}
    allr[tag] = d
    json.dump(allr, open(REPORT, "w"), indent=1)


def _model(name, seed, spread=False):
    sd = synth.spread_state_dict(name, seed) if spread else synth.synth_state_dict(name, seed)
    m = oc.create_model(name, device=DEV)
    m.load_state_dict(sd, strict=True)
    return m, sd


class _DS:
    pass


class FakeLoader:
    def __init__(self, batches, n, attr):
        self.batches = batches
        self.dataset = _DS()
        setattr(self.dataset, attr, list(range(n)))

    def __iter__(self):
        return iter(self.batches)


def _flat_ln(model):
    return torch.cat([p.detach().flatten() for p in oc.collect_params(model, False)[0]]).cpu()


def _run_hip_manual(tag, packed=False):
    """Step the HIP path by hand to expose per-step intermediates; returns dict of results."""
    g = load_golden(tag)
    m = g["meta"]
    cfg = synth.get_cfg(m["model"])
    model, sd = _model(m["model"], m["wseed"], spread=m.get("spread", False))
    model.pack_tokens = packed
    np.testing.assert_allclose(weight_checksums(sd), g["wsum"], rtol=1e-9)
    tb, ib = make_batches(cfg, m["n_text"], m["n_image"], m["bs_text"], m["bs_image"], m["dseed"],
                          related=m.get("spread", False))
    out = dict(g=g, m=m, sd=sd, tb=tb, ib=ib)
    with torch.no_grad():
        out["img_rows"] = model.encode_image(ib[0][0][:8].to(DEV), normalize=True).cpu()
        out["txt_unnorm"] = model.encode_text(tb[0][0][:8].to(DEV), normalize=False).cpu()
    oc.freeze_parameters(model, False)
    params, names = oc.collect_params(model, False)
    assert ["transformer." + n for n in names] == m["names"]
    opt = AdamW(params, lr=m["lr"], weight_decay=m["wd"])
    G = torch.zeros(m["n_image"], cfg["embed_dim"], device=DEV)
    for images, index in ib:
        G[index.to(DEV)] = model.encode_image(images.to(DEV), normalize=True)
    score = torch.full((m["n_text"], m["n_image"]), -100.0, device=DEV)
    losses, idxs, snaps, grads1, feats = [], [], [], None, None
    for text, pos, neg, inv, idx in tb:
        toks = [x.to(DEV) for x in (text, pos, neg, inv)]
        for s in range(m["steps"]):
            if feats is None:
                with torch.no_grad():
                    feats = [model.encode_text(x[:8], normalize=True).cpu() for x in toks]
            if grads1 is None:
                # first step by hand to capture the gradient before AdamW consumes it
                l, ci, S = T.tta_step(model, _NoStep(opt), toks, G, m["T1"], m["T2"])
                grads1 = opt.flat_grad.detach().cpu().clone()
                opt.step(); opt.zero_grad()
            else:
                l, ci, S = T.tta_step(model, opt, toks, G, m["T1"], m["T2"])
            losses.append(l.cpu()); idxs.append(ci.cpu())
            if len(snaps) < m["steps"]:
                snaps.append(_flat_ln(model))
        score[idx.to(DEV)] = S
    out.update(losses=torch.stack(losses), idxs=idxs, snaps=snaps, grads1=grads1, feats=feats, score=score.cpu(),
               final=_flat_ln(model), gallery=G.cpu())
    return out


class _NoStep:
    """Optimizer wrapper that skips step/zero_grad (used once to peek at the gradient)."""

    def __init__(self, opt):
        self.flat_grad = opt.flat_grad

    def step(self):
        pass

    def zero_grad(self):
        pass


# Gates on what the optimizer actually receives and does.  The adapted-parameter gate of
# north_star (rel-L2 <= 1e-3 on the parameter VECTOR) cannot fail on its own: three AdamW steps at lr 2e-4 move a
# parameter by <= 6e-4, so an implementation that never stepped would score ~8e-4.  These can fail:
#   grad1_cos / grad1_rel : step-1 LayerNorm gradient vs the reference's autograd gradient;
#   update_cos / update_rel: the parameter UPDATE (final - initial) vs the reference's update.  Adam's first steps move
#     every element by ~lr * sign(g) whatever |g| is, so an element whose gradient is smaller than the bf16 noise and
#     flips sign contributes a full 2*lr error: update_rel ~ 2*sqrt(f) for a flipped fraction f (0.13 <-> f ~ 0.4 %;
#     `update_sign_flip_frac` in the report is the measured f).  An un-stepped model has update_rel = 1.
GRAD1_COS_MIN, GRAD1_REL_MAX = 0.999, 0.05
UPDATE_COS_MIN, UPDATE_REL_MAX = 0.99, 0.15


# The spread weights scale every residual branch by 4: bf16 rounding noise compounds through the twelve layers about four
# times as fast as with the plain weights (feature cosine 0.99986 instead of 0.99997; test_gradient_error_attribution
# [spread] splits the gradient error into its forward and backward shares), so the step-1 gradient agrees to ~10 % and
# the three-step Adam update -- which moves every element by ~lr * sign(g) -- to ~28 %.  Stated bounds for that fixture;
# the three controls (no step, negated, halved update) still fail them.
SPREAD_GRAD_GATE, SPREAD_UPDATE_GATE = (0.99, 0.15), (0.93, 0.40)


def _update_gate(final, p0, ref_final, cos_min=UPDATE_COS_MIN, rel_max=UPDATE_REL_MAX):
    """(passes, cos, rel, sign-flip fraction) of the parameter update final - p0 against the reference's."""
    up, ref = (final - p0).double(), (torch.as_tensor(ref_final) - p0).double()
    if float(up.norm()) == 0.0:
        return False, 0.0, 1.0, 1.0
    cos = min_cosine(up.view(1, -1), ref.view(1, -1))
    rel = rel_l2(up, ref)
    flips = float((torch.sign(up) != torch.sign(ref)).double().mean())
    return (cos >= cos_min and rel <= rel_max), cos, rel, flips


def _check_against_reference(tag, loss_tol, param_tol, packed=False, per_term_tol=None, max_idx_flips=None,
                             grad_gate=(GRAD1_COS_MIN, GRAD1_REL_MAX), update_gate=(UPDATE_COS_MIN, UPDATE_REL_MAX)):
    """per_term_tol: gate EVERY loss term (semantics reversion included) relative to itself, SURVEY.md §8(d)'s
    "each loss term and total rel <= 1e-3"; max_idx_flips: exact bound on coarse-index mismatches."""
    r = _run_hip_manual(tag, packed)
    g, m = r["g"], r["m"]
    rep = {}
    rep["cos_img"] = min_cosine(r["img_rows"], g["img_rows"])
    rep["rel_txt_unnorm"] = rel_l2(r["txt_unnorm"], g["txt_unnorm"])
    rep["cos_txt_unnorm"] = min_cosine(r["txt_unnorm"], g["txt_unnorm"])
    for name, f in zip(("feat_text", "feat_pos", "feat_neg", "feat_inv"), r["feats"]):
        rep["cos_" + name] = min_cosine(f, g[name])
        # stricter than the cosine gate: error relative to the SPREAD of the features over the batch
        rep["centered_rel_" + name] = rel_l2(f - f.mean(0), torch.from_numpy(g[name]) - torch.from_numpy(g[name]).mean(0))
    L = r["losses"].numpy()
    rep["loss_rel"] = [float(np.abs(L[:, 0] / g["loss_ent"] - 1).max()), float(np.abs(L[:, 1] - g["loss_sr"]).max() /
                       max(np.abs(g["loss_sr"]).max(), 1e-9)), float(np.abs(L[:, 2] / g["loss_tri"] - 1).max())]
    tot_ref = g["loss_ent"] + g["loss_sr"] + g["loss_tri"]
    rep["loss_total_rel"] = float(np.abs(L.sum(1) / tot_ref - 1).max())
    rep["loss_sr_abs_over_total"] = float((np.abs(L[:, 1] - g["loss_sr"]) / tot_ref).max())
    n_idx_bad = sum(int((ci.numpy() != gi[:len(ci)]).sum()) for ci, gi in zip(r["idxs"], g["coarse_idx"]))
    rep["coarse_idx_mismatch"] = n_idx_bad
    rep["n_idx"] = int(sum(len(ci) for ci in r["idxs"]))
    if "grads_step1" in g:
        rep["grad1_rel"] = rel_l2(r["grads1"], g["grads_step1"])
        rep["grad1_cos"] = min_cosine(r["grads1"].view(1, -1), torch.from_numpy(g["grads_step1"]).view(1, -1))
    rep["param_rel_steps"] = [rel_l2(s, gs) for s, gs in zip(r["snaps"], g["params_after"])]
    rep["param_rel_final"] = rel_l2(r["final"], g["final_params"])
    p0 = torch.cat([r["sd"][k].flatten() for k in m["names"]])
    ok_update, rep["update_cos"], rep["update_rel"], rep["update_sign_flip_frac"] = _update_gate(
        r["final"], p0, g["final_params"], *update_gate)
    # controls: the same gate must FAIL for a model that never stepped and for one that stepped the wrong way
    noop_ok = _update_gate(p0.clone(), p0, g["final_params"], *update_gate)[0]
    wrong_ok = _update_gate(2 * p0 - torch.from_numpy(g["final_params"]), p0, g["final_params"], *update_gate)[0]
    half_ok = _update_gate(p0 + 0.5 * (torch.from_numpy(g["final_params"]) - p0), p0, g["final_params"], *update_gate)[0]
    rep["control_gate_passes"] = dict(no_step=noop_ok, negated_update=wrong_ok, half_update=half_ok)
    rep["param_rel_if_never_stepped"] = rel_l2(p0, g["final_params"])
    ref_score = torch.from_numpy(g["score"]) if "score" in g else None
    if ref_score is not None:
        rep["score_max_abs"] = float((r["score"] - ref_score).abs().max())
    else:
        rep["score_max_abs"] = float((r["score"][:8] - torch.from_numpy(g["score_rows"])).abs().max())
    # recall on OUR score matrix: GPU kernels vs oracle selection on identical scores -> bit-exact
    pos = torch.zeros_like(r["score"], dtype=torch.bool)
    pos[torch.arange(m["n_text"]), torch.arange(m["n_text"]) % m["n_image"]] = True
    ks = [int(k) for k in g["recall_ks"]]
    mine = R.recall_metrics(r["score"].to(DEV), pos.to(DEV), ks)
    orc = O.recall_metrics(r["score"], pos, ks)
    rep["recall_hip"] = mine
    for k in ks:
        assert torch.equal(ops.topk_rows(r["score"].to(DEV), k)[1].cpu().long(), O.topk_rows(r["score"], k)[1])
    assert mine == pytest.approx(orc, abs=1e-7)
    rep["recall_ref_t2i"] = [float(x) for x in g["recall_t2i"]]
    rep["recall_ref_i2t"] = [float(x) for x in g["recall_i2t"]]
    _report(tag + ("_packed" if packed else ""), rep)
    print(json.dumps(rep, indent=1))
    # ---- gates ----
    assert rep["cos_img"] >= 0.999 and rep["cos_txt_unnorm"] >= 0.999
    for name in ("feat_text", "feat_pos", "feat_neg", "feat_inv"):
        assert rep["cos_" + name] >= 0.999, name
    # "The adaptation loss must match within 1e-3 relative" (north_star): the loss the optimizer sees is the
    # SUM of the three terms.  Entropy and triplet terms individually stay within 2e-3.  The semantics-
    # reversion term is a hinge-max over differences of near-equal cosines divided by T2=0.07 and jumps when
    # a coarse index flips, so bf16 feature noise (~1e-2 of the feature spread) moves it by a few 1e-3
    # relative to ITSELF; it is gated relative to the total loss and reported per term in parity_report.json.
    # (The reference's own GPU path runs TF32 matmuls, main_TTA.py:61-67, and has the same sensitivity.)
    assert rep["loss_total_rel"] <= loss_tol, rep
    assert rep["loss_rel"][0] <= 2e-3 and rep["loss_rel"][2] <= 2e-3, rep["loss_rel"]
    assert rep["loss_sr_abs_over_total"] <= loss_tol, rep
    assert rep["param_rel_final"] <= param_tol and max(rep["param_rel_steps"]) <= param_tol
    # the gates that can fail (see above) + their controls
    if "grad1_cos" in rep:
        assert rep["grad1_cos"] >= grad_gate[0] and rep["grad1_rel"] <= grad_gate[1], rep
    assert ok_update, (rep["update_cos"], rep["update_rel"])
    assert not noop_ok and not wrong_ok and not half_ok, rep["control_gate_passes"]
    assert rep["coarse_idx_mismatch"] <= (max(1, rep["n_idx"] // 20) if max_idx_flips is None else max_idx_flips)
    assert rep["score_max_abs"] < 2e-2
    if per_term_tol is not None:
        assert max(rep["loss_rel"]) <= per_term_tol, rep["loss_rel"]
    if ref_score is not None and m.get("spread", False):
        # recall@1 far from chance on this fixture: "matching recall@1" (north_star) means something here
        ref_top1 = ref_score.argmax(1)
        rep["top1_agreement"] = float((r["score"].argmax(1) == ref_top1).float().mean())
        assert rep["recall_ref_t2i"][0] > 0.5
        assert mine["image_retrieval_recall@1"] == pytest.approx(rep["recall_ref_t2i"][0], abs=1e-7), mine
        assert rep["top1_agreement"] == 1.0, rep["top1_agreement"]
        _report(tag + ("_packed" if packed else ""), rep)
    return rep


def test_tiny_epoch_vs_reference():
    _check_against_reference("tiny", loss_tol=1e-3, param_tol=1e-3)


def test_vitb32_b32_epoch_vs_reference():
    _check_against_reference("vitb32_b32", loss_tol=1e-3, param_tol=1e-3)


def test_vitb32_b256_epoch_vs_reference():
    _check_against_reference("vitb32_b256", loss_tol=1e-3, param_tol=1e-3)


def test_vitb32_spread_b256_epoch_vs_reference():
    """The NON-DEGENERATE fixture (VERDICT r2 next #1a): "spread" weights (tests/golden/make_spread_weights.py) and
    related caption sets -- text-text cosine ~0.29, image-image ~0.50, recall@1 ~0.7 -- run through the reference's
    unmodified train_one_epoch at ViT-B-32, B = 256.  Gated per term at SURVEY.md §8(d)'s 1e-3 (the semantics-reversion
    term included), zero coarse-index flips, top-1 retrieval identical to the reference's on every row."""
    _check_against_reference("vitb32_spread_b256", loss_tol=1e-3, param_tol=1e-3, per_term_tol=1e-3, max_idx_flips=0,
                             grad_gate=SPREAD_GRAD_GATE, update_gate=SPREAD_UPDATE_GATE)


def test_vitb32_spread_b256_packed_epoch_vs_reference():
    _check_against_reference("vitb32_spread_b256", loss_tol=1e-3, param_tol=1e-3, packed=True, per_term_tol=1e-3,
                             max_idx_flips=0, grad_gate=SPREAD_GRAD_GATE, update_gate=SPREAD_UPDATE_GATE)


@pytest.mark.parametrize("tag", ["vitb32_b32", "vitb32_b256", "vitb32_spread_b256"])
def test_sr_term_flip_vs_value_attribution(tag):
    """Separates the two ways the semantics-reversion term (train_TTA.py:53-65, a hinge-MAX over a row of outputs_aux)
    can differ from the reference's at step 1, where both sides hold identical parameters (VERDICT r2 next #1b):
      free   -- the HIP path's own coarse indices (an index FLIP swaps a whole column of outputs_aux);
      forced -- the reference's coarse_idx forced on the HIP features (same columns: what is left is the bf16 feature
                noise inside s_ii - s_ij, divided by T2 = 0.07).
    Reported per fixture in parity_report.json; on the spread fixture both must meet 1e-3 of the term itself, on the
    collinear ones the forced value is bounded by 2 % of the term and by 1e-3 of the total loss."""
    g = load_golden(tag)
    m = g["meta"]
    cfg = synth.get_cfg(m["model"])
    spread = m.get("spread", False)
    model, sd = _model(m["model"], m["wseed"], spread=spread)
    tb, ib = make_batches(cfg, m["n_text"], m["n_image"], m["bs_text"], m["bs_image"], m["dseed"], related=spread)
    G = torch.zeros(m["n_image"], cfg["embed_dim"], device=DEV)
    with torch.no_grad():
        for images, index in ib:
            G[index.to(DEV)] = model.encode_image(images.to(DEV), normalize=True)
    text, pos, neg, inv, _ = tb[0]
    B = text.shape[0]
    tower = model._text_tower()
    feats, _ = tower.forward(torch.cat([text, pos, inv, neg], dim=0).to(DEV), True, False)
    t, p, v, n = feats[:B], feats[B:2 * B], feats[2 * B:3 * B], feats[3 * B:]
    coarse, idx, _ = ops.coarse_align(t, p, n, G)
    idx_ref = torch.from_numpy(g["coarse_idx"][0][:B].astype(np.int64)).to(DEV)
    free = ops.tta_loss_fwd_bwd(t, p, v, t, coarse, m["T1"], m["T2"], 0)[0].cpu()
    forced = ops.tta_loss_fwd_bwd(t, p, v, t, G[idx_ref].contiguous(), m["T1"], m["T2"], 0)[0].cpu()
    ref = torch.tensor([g["loss_ent"][0], g["loss_sr"][0], g["loss_tri"][0]], dtype=torch.float64)
    rel = lambda x: [abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(x, ref)]
    rep = dict(rows=B, idx_flips=int((idx.long() != idx_ref).sum()), loss_ref=[float(x) for x in ref],
               loss_free=[float(x) for x in free], loss_forced_ref_idx=[float(x) for x in forced],
               rel_free=rel(free), rel_forced_ref_idx=rel(forced),
               sr_abs_over_total_free=abs(float(free[1]) - float(ref[1])) / float(ref.sum()),
               sr_abs_over_total_forced=abs(float(forced[1]) - float(ref[1])) / float(ref.sum()))
    _report("sr_attribution_" + tag, rep)
    print(json.dumps(rep, indent=1))
    if spread:
        assert rep["idx_flips"] == 0 and max(rep["rel_free"]) <= 1e-3 and max(rep["rel_forced_ref_idx"]) <= 1e-3, rep
    else:
        assert rep["rel_forced_ref_idx"][1] <= 2e-2 and rep["sr_abs_over_total_forced"] <= 1e-3, rep
        assert rep["rel_forced_ref_idx"][0] <= 2e-3 and rep["rel_forced_ref_idx"][2] <= 2e-3, rep


def test_vitl14_b128_epoch_vs_reference():
    """BASELINE config 3's model (ViT-L-14: text width 768, D = 768, 24 vision layers of width 1024) against the
    REFERENCE's own run (tests/golden/vitl14_b128.npz; the reference's hard-coded 512-column gallery buffer widened to
    768, see make_golden.py)."""
    _check_against_reference("vitl14_b128", loss_tol=1e-3, param_tol=1e-3)


@pytest.mark.parametrize("tag", ["tiny", "vitb32_b32", "vitb32_b256"])
def test_packed_tokens_epoch_vs_reference(tag):
    """The opt-in packed text batches against the reference's golden vectors, not only against the full-length run."""
    _check_against_reference(tag, loss_tol=1e-3, param_tol=1e-3, packed=True)


def test_vitl14_step_at_config3_shard_vs_oracle():
    """One full ViT-L-14 adaptation step at config 3's per-GPU shape (Bl = 256 caption rows x 4 variants, D = 768)
    against the fp32 oracle: features cos >= 0.999, loss within 1e-3, gradient direction.  The oracle's side was computed
    once on the host by tests/golden/make_oracle_fixtures.py (minutes of CPU time) and is loaded from the fixture."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "golden", "make_oracle_fixtures.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    g = load_golden("vitl14_bl256_oracle_step")
    m = g["meta"]
    model, sd = _model(m["model"], m["wseed"])
    toks, G = fx.inputs(m)                                   # the same seeded integer draws the fixture was made from
    np.testing.assert_allclose([G.double().sum().item(), G.double().abs().sum().item()], g["gallery_sum"], rtol=1e-9)
    B = m["B"]
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=m["lr"], weight_decay=m["wd"])
    with torch.no_grad():
        feats = [model.encode_text(x[:32].to(DEV), normalize=True).cpu() for x in toks]
    losses, idx, S = T.tta_step(model, _NoStep(opt), [x.to(DEV) for x in toks], G.to(DEV), m["T1"], m["T2"])
    grad = opt.flat_grad.detach().cpu().clone()
    lo, gref = g["losses"], torch.from_numpy(g["grad"])
    rep = dict(cos=[min_cosine(a, b) for a, b in zip(feats, g["feats"])],
               loss=[float(x) for x in losses.cpu()], loss_ref=[float(x) for x in lo],
               loss_total_rel=abs(float(losses.sum()) - float(lo.sum())) / abs(float(lo.sum())),
               idx_mismatch=int((idx.cpu().numpy() != g["idx"]).sum()),
               grad_cos=min_cosine(grad.view(1, -1), gref.view(1, -1)), grad_rel=rel_l2(grad, gref),
               score_max_abs=float((S[:8].cpu() - torch.from_numpy(g["score_rows"])).abs().max()))
    _report("vitl14_bl256_step_vs_oracle", rep)
    assert min(rep["cos"]) >= 0.999, rep
    assert rep["loss_total_rel"] <= 1e-3, rep
    assert abs(rep["loss"][0] - rep["loss_ref"][0]) <= 2e-3 * abs(rep["loss_ref"][0]), rep
    assert abs(rep["loss"][2] - rep["loss_ref"][2]) <= 2e-3 * abs(rep["loss_ref"][2]), rep
    assert rep["idx_mismatch"] <= B // 20 and rep["score_max_abs"] < 2e-2, rep
    assert rep["grad_cos"] >= GRAD1_COS_MIN and rep["grad_rel"] <= GRAD1_REL_MAX, rep


@pytest.mark.parametrize("spread", [False, True])
def test_gradient_error_attribution(spread):
    """Where does the 2-4 % error of the step-1 LayerNorm gradient (grad1_rel) come from?  ViT-B-32, 32 rows.
    (a) forward: the loss gradient w.r.t. the FEATURES, computed by the same fp32 loss kernel from HIP features and from
        the oracle's fp32 features (same coarse images) -- the share caused by bf16 noise in the forward pass;
    (b) backward: the oracle's feature gradient pushed through the HIP backward pass vs through the oracle's -- the
        share caused by bf16 operands in the data-gradient GEMMs.  Reported in parity_report.json; DESIGN.md §2."""
    name = "ViT-B-32"
    model, sd = _model(name, 0, spread=spread)
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    B = 32
    toks = (synth.synth_related_caption_sets if spread else synth.synth_caption_sets)(
        B, t["context_length"], t["vocab_size"], 1234)
    G = torch.nn.functional.normalize(torch.randn(64, cfg["embed_dim"], generator=torch.Generator().manual_seed(3)), dim=-1)
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
    tower = model._text_tower()
    text, pos, neg, inv = [x.to(DEV) for x in toks]
    feats, ctx = tower.forward(torch.cat([text, pos, inv, neg], dim=0), True, True, n_grad=3 * B)
    th, ph, vh = feats[:B], feats[B:2 * B], feats[2 * B:3 * B]
    names = O.adapted_param_names(sd)
    fo, caches = [], []
    for x in (toks[0], toks[1], toks[3]):
        f, c = O.encode_text_fwd(sd, x, True, save=True)
        fo.append(f); caches.append(c)
    n_o = O.encode_text_fwd(sd, toks[2], True)[0]
    c_o, _, _ = O.coarse_align_t2i(fo[0], fo[1], n_o, G)
    cd = c_o.to(DEV)
    _, dth, dph, dvh = ops.tta_loss_fwd_bwd(th, ph, vh, th, cd, 0.03, 0.07)
    to, po, vo = (f.to(DEV) for f in fo)
    _, dto, dpo, dvo = ops.tta_loss_fwd_bwd(to, po, vo, to, cd, 0.03, 0.07)
    # (b) the oracle's feature gradient through both backward passes
    grads = T._ln_grad_views(model)
    tower.backward(ctx, torch.cat([dto, dpo, dvo], dim=0), grads, accumulate=False)
    g_hip = opt.flat_grad.detach().cpu().clone()
    g_or = {}
    for c, d in zip(caches, (dto, dpo, dvo)):
        O.encode_text_bwd(sd, c, d.cpu(), g_or)
    g_ref = torch.cat([g_or[k].flatten() for k in names])
    # (a)+(b): HIP features' gradient through the HIP backward = what the step really uses
    tower.backward(ctx, torch.cat([dth, dph, dvh], dim=0), grads, accumulate=False)
    g_all = opt.flat_grad.detach().cpu().clone()
    rep = dict(feature_centered_rel=rel_l2(th.cpu() - th.cpu().mean(0), fo[0] - fo[0].mean(0)),
               dfeat_rel_from_forward_noise=[rel_l2(a.cpu(), b.cpu()) for a, b in ((dth, dto), (dph, dpo), (dvh, dvo))],
               grad_rel_backward_only=rel_l2(g_hip, g_ref), grad_rel_total=rel_l2(g_all, g_ref),
               grad_cos_backward_only=min_cosine(g_hip.view(1, -1), g_ref.view(1, -1)),
               grad_cos_total=min_cosine(g_all.view(1, -1), g_ref.view(1, -1)))
    _report("gradient_error_attribution_vitb32_b32" + ("_spread" if spread else ""), rep)
    print(json.dumps(rep, indent=1))
    cos_min, rel_max = SPREAD_GRAD_GATE if spread else (GRAD1_COS_MIN, GRAD1_REL_MAX)
    assert rep["grad_cos_total"] >= cos_min and rep["grad_rel_total"] <= rel_max, rep
    assert rep["grad_rel_backward_only"] <= rel_max, rep


@pytest.mark.parametrize("packed", [False, True])
def test_train_one_epoch_dropin_matches_manual_loop(packed):
    """The public entry point (same signature as the reference) == the hand-stepped loop, bit for bit, with the packed
    text batches (the default) and with all context_length positions."""
    r = _run_hip_manual("tiny", packed)
    m = r["m"]
    model, _ = _model(m["model"], m["wseed"])
    model.pack_tokens = packed
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=m["lr"], weight_decay=m["wd"])
    args = types.SimpleNamespace(distributed=False, tta_only_visual=False, name="neat", tta_steps=m["steps"],
                                 tta_temperature=m["T1"], tta_temperature2=m["T2"])
    i2t, t2i = T.train_one_epoch(model, lambda x: x, FakeLoader(r["tb"], m["n_text"], "text"),
                                 FakeLoader(r["ib"], m["n_image"], "image"), opt, torch.device(DEV), args)
    assert torch.equal(t2i.cpu(), r["score"])
    assert torch.equal(_flat_ln(model), r["final"])
    assert (i2t == -100).all() and tuple(i2t.shape) == (m["n_image"], m["n_text"])


def test_autograd_api_and_torch_optimizer():
    """encode_text is differentiable w.r.t. the LN parameters through autograd (the reference's call
    pattern: four encode_text calls, loss.backward(), a stock torch optimizer) and agrees with the fused
    step."""
    g = load_golden("tiny")
    m = g["meta"]
    cfg = synth.get_cfg(m["model"])
    tb, ib = make_batches(cfg, m["n_text"], m["n_image"], m["bs_text"], m["bs_image"], m["dseed"])
    toks = [x.to(DEV) for x in tb[0][:4]]
    # fused path
    model_a, _ = _model(m["model"], m["wseed"])
    oc.freeze_parameters(model_a, False)
    pa, _ = oc.collect_params(model_a, False)
    opt_a = AdamW(pa, lr=m["lr"], weight_decay=m["wd"])
    G = model_a.encode_image(ib[0][0].to(DEV), normalize=True)
    T.tta_step(model_a, _NoStep(opt_a), toks, G, m["T1"], m["T2"])
    grad_a = opt_a.flat_grad.cpu().clone()
    opt_a.step()
    # autograd path with torch.optim.AdamW, loss terms built from the kernel-backed helpers
    model_b, _ = _model(m["model"], m["wseed"])
    oc.freeze_parameters(model_b, False)
    pb, _ = oc.collect_params(model_b, False)
    opt_b = torch.optim.AdamW(pb, lr=m["lr"], weight_decay=m["wd"])
    t = model_b.encode_text(toks[0], normalize=True)
    p = model_b.encode_text(toks[1], normalize=True)
    with torch.no_grad():
        n = model_b.encode_text(toks[2], normalize=True)
    v = model_b.encode_text(toks[3], normalize=True)
    assert t.requires_grad and not n.requires_grad
    coarse, _, _, _, S = T.coarse_aligning_t2i(t, p, n, G)
    losses, dt, dp, dv = T.tta_loss(t.detach(), p.detach(), v.detach(), coarse, m["T1"], m["T2"])
    torch.autograd.backward([t, p, v], [dt, dp, dv])
    grad_b = torch.cat([q.grad.flatten() for q in pb]).cpu()
    assert rel_l2(grad_b, grad_a) < 2e-3        # same math, different batching of the three backward passes
    opt_b.step()
    assert rel_l2(_flat_ln(model_b), _flat_ln(model_a)) < 1e-5


@pytest.mark.parametrize("name", ["ViT-B-16", "ViT-L-14"])
def test_other_clip_configs_match_oracle(name):
    """BASELINE configs 3/5 use ViT-L-14 / ViT-B-16: other widths (768 / 1024), longer vision sequences (197 / 257
    tokens -> the one-wave attention kernels), 24 vision layers, D = 768.  Encoders and one full TTA step vs the
    fp32 oracle (parity of the oracle itself is pinned on ViT-B-32; these configs share every formula)."""
    model, sd = _model(name, 3)
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    toks = synth.synth_caption_sets(3, t["context_length"], t["vocab_size"], 11)
    img = synth.synth_images(2, cfg["vision_cfg"]["image_size"], 12)
    with torch.no_grad():
        fi = model.encode_image(img.to(DEV), normalize=True).cpu()
        ft = model.encode_text(toks[0].to(DEV), normalize=True).cpu()
    assert min_cosine(fi, O.encode_image_fwd(sd, img)) >= 0.999
    assert min_cosine(ft, O.encode_text_fwd(sd, toks[0])[0]) >= 0.999
    # one adaptation step
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
    G = torch.nn.functional.normalize(torch.randn(7, cfg["embed_dim"], generator=torch.Generator().manual_seed(1)), dim=-1)
    losses, idx, S = T.tta_step(model, opt, [x.to(DEV) for x in toks], G.to(DEV), 0.03, 0.07)
    st = O.AdamWState({k: sd[k] for k in O.adapted_param_names(sd)})
    lo, io, So, _, _ = O.tta_step(sd, st, toks, G, 2e-4, 1e-3, 0.03, 0.07)
    assert abs(float(losses.sum()) - float(lo.sum())) <= 2e-3 * abs(float(lo.sum()))
    assert float((S.cpu() - So).abs().max()) < 5e-3
    flat_o = torch.cat([sd[k].flatten() for k in O.adapted_param_names(sd)])
    assert rel_l2(_flat_ln(model), flat_o) < 1e-3


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("B,N", [(1, 1), (1, 5), (2, 3), (65, 2)])
def test_tta_step_edge_batches_vs_oracle(B, N, packed):
    """The smallest and the raggedest adaptation batches -- one caption row (the entropy softmax runs over a single
    coarse image, the hinge and triplet sums over one row), a gallery of one image (every caption aligns to it), 65 rows
    (one more than the 64-row tiles of the loss and coarse kernels) against 2 images -- one step against the fp32 oracle:
    loss terms, coarse indices, score rows and the adapted LayerNorm parameters."""
    name = "neat-tiny"
    model, sd = _model(name, 0)
    model.pack_tokens = packed
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    toks = synth.synth_caption_sets(B, t["context_length"], t["vocab_size"], 100 + B)
    G = torch.nn.functional.normalize(torch.randn(N, cfg["embed_dim"], generator=torch.Generator().manual_seed(B + N)), dim=-1)
    oc.freeze_parameters(model, False)
    opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
    losses, idx, S = T.tta_step(model, opt, [x.to(DEV) for x in toks], G.to(DEV), 0.03, 0.07)
    st = O.AdamWState({k: sd[k] for k in O.adapted_param_names(sd)})
    lo, io, So, _, _ = O.tta_step(sd, st, toks, G, 2e-4, 1e-3, 0.03, 0.07)
    assert tuple(S.shape) == (B, N) and tuple(idx.shape) == (B,)
    assert torch.isfinite(losses).all()
    torch.testing.assert_close(losses.cpu(), lo.float(), rtol=5e-3, atol=2e-3)
    assert float((S.cpu() - So).abs().max()) < 5e-3
    assert int((idx.cpu().long() != io).sum()) <= (0 if N == 1 else max(1, B // 20))
    flat_o = torch.cat([sd[k].flatten() for k in O.adapted_param_names(sd)])
    assert rel_l2(_flat_ln(model), flat_o) < 1e-3


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0 on the 1-GPU box
    try:
        model, _ = _model("neat-tiny", 0)
        oc.freeze_parameters(model, False)
        params, _ = oc.collect_params(model, False)
        opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
        cfg = synth.get_cfg("neat-tiny")["text_cfg"]
        toks = synth.synth_caption_sets(9, cfg["context_length"], cfg["vocab_size"], 77)
        G = torch.nn.functional.normalize(torch.randn(12, 512, generator=torch.Generator().manual_seed(5)), dim=-1).to(DEV)
        lo, hi = T._shard(9, world, rank)
        local = [x[lo:hi].contiguous().to(DEV) for x in toks]
        for _ in range(2):
            losses, idx, S = T.tta_step(model, opt, local, G, 0.03, 0.07, world, rank)
        S_all = T._all_gather_rows(S, 9, world)
        torch.save(dict(losses=losses.cpu(), S=S_all.cpu(), params=_flat_ln(model)), os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_step_equals_single_gpu(tmp_path):
    """SURVEY §8e equality requirement on the real kernels: two ranks (uneven row shards 5 + 4, gloo rendezvous,
    both on cuda:0) must reproduce the single-process step on the same global batch -- losses, LN parameters and
    score rows -- up to fp32 reduction order."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    model, _ = _model("neat-tiny", 0)
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
    cfg = synth.get_cfg("neat-tiny")["text_cfg"]
    toks = [x.to(DEV) for x in synth.synth_caption_sets(9, cfg["context_length"], cfg["vocab_size"], 77)]
    G = torch.nn.functional.normalize(torch.randn(12, 512, generator=torch.Generator().manual_seed(5)), dim=-1).to(DEV)
    for _ in range(2):
        losses, idx, S = T.tta_step(model, opt, toks, G, 0.03, 0.07)
    assert torch.equal(r0["params"], r1["params"])                       # ranks stay in lock step
    assert rel_l2(r0["params"], _flat_ln(model)) < 2e-5
    torch.testing.assert_close(r0["losses"], losses.cpu(), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(r0["S"], S.cpu(), rtol=0, atol=2e-3)


def _dp_epoch_worker(rank, world, port, out):
    """train_one_epoch on ViT-B-32 through the package's own create_sampler / create_loader (DistributedSamplers, as the
    reference's launcher wires them): every rank encodes its half of the gallery and adapts on its half of each batch."""
    import faulthandler
    import torch.distributed as dist
    from neat_amd.data import create_loader, create_sampler
    from test_dp_gloo import _ImageDS, _TextDS
    faulthandler.dump_traceback_later(90, exit=True)          # a hang becomes a failure with a stack, not a silent stall
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0 on the 1-GPU box
    try:
        model, _ = _model("ViT-B-32", 0)
        oc.freeze_parameters(model, False)
        params, _ = oc.collect_params(model, False)
        opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
        sets, images = _dp_epoch_inputs()
        ds = [_TextDS(sets), _ImageDS(images)]
        lt, li = create_loader(ds, create_sampler(ds, [False, False], world, rank), batch_size=[8, 5], num_workers=[0, 0],
                               is_trains=[False, False], collate_fns=[None, None])
        args = types.SimpleNamespace(distributed=True, tta_only_visual=False, name="neat", tta_steps=2,
                                     tta_temperature=0.03, tta_temperature2=0.07)
        _, score = T.train_one_epoch(model, lambda x: x, lt, li, opt, torch.device(DEV), args)
        torch.save(dict(score=score.cpu(), params=_flat_ln(model)), os.path.join(out, f"e{rank}.pt"))
        faulthandler.cancel_dump_traceback_later()
    finally:
        dist.destroy_process_group()


def _dp_epoch_inputs():
    cfg = synth.get_cfg("ViT-B-32")
    t = cfg["text_cfg"]
    sets = synth.synth_caption_sets(27, t["context_length"], t["vocab_size"], 55)     # 27 -> 28 rows: batches 16 + 12
    images = synth.synth_images(13, cfg["vision_cfg"]["image_size"], 56)              # 13 -> 14 images
    return sets, images


@pytest.mark.timeout(400)
def test_data_parallel_epoch_vitb32_sharded_gallery_equals_single_gpu(tmp_path):
    """Two ranks, ViT-B-32, DistributedSampler loaders (sharded gallery + all-gather, text batch already local): scores and
    adapted LN parameters equal the single-GPU epoch on the same global batches up to fp32 reduction order."""
    import socket
    import torch.multiprocessing as mp
    from test_dp_gloo import _sampler_order
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_epoch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "e0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "e1.pt", weights_only=True)
    sets, images = _dp_epoch_inputs()
    per_rank = _sampler_order(27, 2)
    batches_idx = [sum((pr[k:k + 8] for pr in per_rank), []) for k in range(0, len(per_rank[0]), 8)]
    tb = [tuple(x[torch.tensor(b)] for x in sets) + (torch.tensor(b),) for b in batches_idx]
    ib = [(images, torch.arange(13))]
    model, _ = _model("ViT-B-32", 0)
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
    args = types.SimpleNamespace(distributed=False, tta_only_visual=False, name="neat", tta_steps=2,
                                 tta_temperature=0.03, tta_temperature2=0.07)
    _, score = T.train_one_epoch(model, lambda x: x, FakeLoader(tb, 27, "text"), FakeLoader(ib, 13, "image"), opt,
                                 torch.device(DEV), args)
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["score"], r1["score"])   # ranks in lock step
    assert (r0["score"] > -100).all()
    assert rel_l2(r0["params"], _flat_ln(model)) < 2e-5
    torch.testing.assert_close(r0["score"], score.cpu(), rtol=0, atol=2e-3)


@pytest.mark.parametrize("name,B", [("neat-tiny", 9), ("ViT-B-32", 6)])
def test_packed_tokens_equal_full_length(name, B):
    """model.pack_tokens drops the positions after EOT (dead behind the causal mask): features, losses, coarse
    indices, scores and the adapted LN parameters must equal the full-length run up to fp32 summation order."""
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    toks = [x.to(DEV) for x in synth.synth_caption_sets(B, t["context_length"], t["vocab_size"], 31)]
    G = torch.nn.functional.normalize(torch.randn(20, cfg["embed_dim"], generator=torch.Generator().manual_seed(2)), dim=-1).to(DEV)
    out = []
    for packed in (False, True):
        model, _ = _model(name, 0)
        model.pack_tokens = packed
        oc.freeze_parameters(model, False)
        params, _ = oc.collect_params(model, False)
        opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
        with torch.no_grad():
            f = model.encode_text(toks[0], normalize=True).cpu()
        for _ in range(2):
            losses, idx, S = T.tta_step(model, opt, toks, G, 0.03, 0.07)
        out.append(dict(f=f, losses=losses.cpu(), idx=idx.cpu(), S=S.cpu(), p=_flat_ln(model)))
    a, b = out
    assert float((a["f"] - b["f"]).abs().max()) < 2e-6          # same kernels row by row
    torch.testing.assert_close(a["losses"], b["losses"], rtol=1e-5, atol=1e-6)
    assert torch.equal(a["idx"], b["idx"])
    assert float((a["S"] - b["S"]).abs().max()) < 1e-5
    assert rel_l2(b["p"], a["p"]) < 1e-6


@pytest.mark.parametrize("lens", [(60, 75), (75, 75), (1, 75)])
def test_packed_tokens_with_full_length_captions(lens):
    """Packed text batches at the long end: captions that fill the 77-token context (EOT at position 76, the truncation
    case of the reference tokenizer), all of them or one-token captions next to full ones -- ViT-B-32, one step, against the
    full-length run of the same step."""
    name, B = "ViT-B-32", 5
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    toks = [x.to(DEV) for x in synth.synth_caption_sets(B, t["context_length"], t["vocab_size"], 41, *lens)]
    assert int(max(x.argmax(dim=-1).max() for x in toks)) == lens[1] + 1
    G = torch.nn.functional.normalize(torch.randn(9, cfg["embed_dim"], generator=torch.Generator().manual_seed(3)), dim=-1).to(DEV)
    out = []
    for packed in (False, True):
        model, _ = _model(name, 0)
        model.pack_tokens = packed
        oc.freeze_parameters(model, False)
        opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
        losses, idx, S = T.tta_step(model, opt, toks, G, 0.03, 0.07)
        out.append(dict(losses=losses.cpu(), idx=idx.cpu(), S=S.cpu(), p=_flat_ln(model)))
    a, b = out
    torch.testing.assert_close(a["losses"], b["losses"], rtol=1e-5, atol=1e-6)
    assert torch.equal(a["idx"], b["idx"])
    assert float((a["S"] - b["S"]).abs().max()) < 1e-5
    assert rel_l2(b["p"], a["p"]) < 1e-6


def test_batch_plan_from_host_tokens_equals_the_device_plan():
    """train_one_epoch computes the packed layout from the tokens it still holds on the host (no device-to-host copy, so
    the host does not wait for the previous batch): the same cu_seqlens as the library's neat_pack_plan on the device
    tokens -- rows with a repeated maximum (first one wins), a row that is all zeros, a sharded slice -- and a step run
    with either plan gives bit-identical results."""
    name = "neat-tiny"
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    host = list(synth.synth_caption_sets(11, t["context_length"], t["vocab_size"], 77))
    host[0] = host[0].clone()
    host[0][2, 5] = host[0][2].max()                    # the maximum twice in one row: EOT = its first position
    host[0][2, 1] = host[0][2].max()
    host[1] = host[1].clone()
    host[1][4] = 0                                      # all equal: position 0
    model, _ = _model(name, 0)
    model.pack_tokens = True
    for lo, hi in ((0, 11), (3, 8)):
        h = [x[lo:hi] for x in host]
        d = [x.to(DEV) for x in h]
        p_dev = T.batch_plan(model, d)
        p_host = T.batch_plan(model, d, h)
        assert torch.equal(p_dev[0], p_host[0])
        assert torch.equal(p_dev[1], p_host[1]) and p_host[1].dtype == torch.int32 and p_host[1].is_cuda
        assert torch.equal(p_dev[2], p_host[2]) and not p_host[2].is_cuda
    with pytest.raises(RuntimeError):
        T.batch_plan(model, d, [x[:2] for x in h])
    G = torch.nn.functional.normalize(torch.randn(9, cfg["embed_dim"], generator=torch.Generator().manual_seed(2)), dim=-1).to(DEV)
    res = []
    for use_host in (False, True):
        model, _ = _model(name, 0)
        model.pack_tokens = True
        oc.freeze_parameters(model, False)
        opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
        d = [x.to(DEV) for x in host]
        plan = T.batch_plan(model, d, host if use_host else None)
        losses, idx, S = T.tta_step(model, opt, d, G, 0.03, 0.07, plan=plan)
        res.append((losses.cpu(), idx.cpu(), S.cpu(), _flat_ln(model)))
    for a, b in zip(*res):
        assert torch.equal(torch.as_tensor(a), torch.as_tensor(b))


class _ListLoader:
    def __init__(self, batches, n):
        self.batches, self.dataset = batches, list(range(n))

    def __iter__(self):
        return iter(self.batches)


def _fake_tokenizer(ctx, vocab):
    """Deterministic stand-in for the BPE tokenizer: strings of integers -> token rows (SOT ... EOT, zero padded)."""
    def tok(texts):
        out = torch.zeros(len(texts), ctx, dtype=torch.int64)
        for i, t in enumerate(texts):
            ids = [int(x) % (vocab - 3) + 1 for x in t.split()][: ctx - 2]
            row = [vocab - 2] + ids + [vocab - 1]
            out[i, : len(row)] = torch.tensor(row)
        return out
    return tok


def test_offline_eval_and_mcq_match_oracle():
    """evaluate_model (offline retrieval, reference evaluation/retrieval.py:11-89) and the MCQ
    scorers (evaluation/mcq.py:10-217) on the tiny model against the oracle encoders + the reference formulas."""
    from neat_amd.evaluation import mcq as MQ, retrieval as R
    model, sd = _model("neat-tiny", 0)
    cfg = synth.get_cfg("neat-tiny")
    tok = _fake_tokenizer(cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"])
    gen = torch.Generator().manual_seed(3)
    n_img, per = 12, 3
    imgs = synth.synth_images(n_img, 32, 21)
    caps = [[" ".join(str(int(x)) for x in torch.randint(0, 400, (6,), generator=gen)) for _ in range(per)]
            for _ in range(n_img)]
    batches = [(imgs[i:i + 4], caps[i:i + 4]) for i in range(0, n_img, 4)]
    args = types.SimpleNamespace(device=DEV, batch_size=4, video=False)
    got = R.evaluate_model(model, _ListLoader(batches, n_img), args, tok)
    fi = O.encode_image_fwd(sd, imgs)
    ft = O.encode_text_fwd(sd, tok([c for cs in caps for c in cs]))[0]
    scores = ft @ fi.t()
    pos = torch.zeros_like(scores, dtype=torch.bool)
    pos[torch.arange(n_img * per), torch.arange(n_img * per) // per] = True
    ref = O.recall_metrics(scores, pos)
    for k, v in ref.items():
        assert abs(got[k] - v) <= 1.0 / n_img + 1e-6, k       # a near-tie may flip one row under bf16 encoders
    assert abs(got["image_retrieval_medR"] - O.median_rank(scores, pos)) <= 1
    assert abs(got["text_retrieval_medR"] - O.median_rank(scores.t(), pos.t())) <= 1
    # MCQ: 4 options per image, option lists are [options][batch]
    opts = [[caps[i][0] for i in range(n_img)]] + [
        [" ".join(str(int(x)) for x in torch.randint(0, 400, (6,), generator=gen)) for _ in range(n_img)] for _ in range(3)]
    answers = torch.randint(0, 4, (n_img,), generator=gen)
    types_ = [("positive", "negative", "hybrid")[i % 3] for i in range(n_img)]
    mb = [(imgs[i:i + 6], [o[i:i + 6] for o in opts], answers[i:i + 6], types_[i:i + 6]) for i in range(0, n_img, 6)]
    m = MQ.evaluate_model(model, _ListLoader(mb, n_img), args, tok, is_synthetic=True)   # answers at any index
    ftxt = torch.stack([O.encode_text_fwd(sd, tok(o))[0] for o in opts])
    logits = torch.einsum("bf,nbf->bn", fi, ftxt)
    margin = logits.topk(2, dim=1)[0]
    clear = (margin[:, 0] - margin[:, 1]) > 2e-2
    ref_acc = float((logits.argmax(1) == answers).float().mean())
    assert abs(m["total_accuracy"] - ref_acc) <= float((~clear).sum()) / n_img + 1e-6
    assert sum(m["predictions_by_type"].values()) == n_img
    b2 = [(imgs[i:i + 6], [o[i:i + 6] for o in opts[:2]], answers[i:i + 6] % 2) for i in range(0, n_img, 6)]
    mb2 = MQ.evaluate_binary_mcq_model(model, _ListLoader(b2, n_img), args, tok)
    logits2 = logits[:, :2]
    clear2 = (logits2[:, 0] - logits2[:, 1]).abs() > 2e-2
    ref_acc2 = float((logits2.argmax(1) == answers % 2).float().mean())
    assert abs(mb2["total_accuracy"] - ref_acc2) <= float((~clear2).sum()) / n_img + 1e-6
    assert int(clear2.sum()) >= n_img // 2              # the comparison above is not vacuous
    # the driver the CLI calls after train_one_epoch (evaluation/utils.py:9-66): result keys + one results.jsonl line
    import json as _json
    import tempfile
    from neat_amd.evaluation.utils import evaluate
    with tempfile.TemporaryDirectory() as d:
        args2 = types.SimpleNamespace(device=DEV, batch_size=4, video=False, save_logs=True, checkpoint_path=d, wandb=False)
        online = scores.to(DEV).contiguous()
        data = {"coco-negated-retrieval": types.SimpleNamespace(dataloader=_ListLoader(batches, n_img)),
                "synthetic-mcq": types.SimpleNamespace(dataloader=_ListLoader(mb, n_img))}
        res = evaluate(model, data, 0, args2, score_matrix_t2i=online, tokenizer=tok)
        want = {f"{p}-coco-negated-{dd}-R@{k}" for p in ("online", "offline") for dd in ("t2i", "i2t") for k in (1, 5, 10)}
        assert want <= set(res) and "synthetic-mcq-total_accuracy" in res
        for k, v in ref.items():                                  # online metrics on the exact fp32 scores = the oracle's
            dd, kk = ("t2i" if k.startswith("image") else "i2t"), k.split("@")[1]
            assert abs(res[f"online-coco-negated-{dd}-R@{kk}"] - v) < 1e-6
        lines = open(os.path.join(d, "results.jsonl")).read().strip().split("\n")
        assert len(lines) == 1 and set(_json.loads(lines[0])) == set(res)


def test_config5_at_model_size_vitb16_binary_mcq_vs_oracle():
    """BASELINE config 5 at MODEL size (VERDICT r2 next #1d): ViT-B-16, one batch of 64 images x 2 captions through
    ``mcq_eval``'s CheXpert branch (reference evaluation/mcq.py:199-205, 330-339) against the fp32 oracle on the host:
    per-image logits within 2e-2, the prediction identical wherever the oracle's margin is clear of that bound, the two
    result keys, and the graph-replayed scorer equal to the eager one."""
    from neat_amd.evaluation import mcq as MQ
    name = "ViT-B-16"
    model, sd = _model(name, 0)
    cfg = synth.get_cfg(name)
    t = cfg["text_cfg"]
    tok = _fake_tokenizer(t["context_length"], t["vocab_size"])
    gen = torch.Generator().manual_seed(17)
    B = 64
    imgs = synth.synth_images(B, 224, 31)
    caps = [[" ".join(str(int(x)) for x in torch.randint(0, 40000, (int(torch.randint(4, 12, (1,), generator=gen)),),
                                                         generator=gen)) for _ in range(B)] for _ in range(2)]
    answers = torch.randint(0, 2, (B,), generator=gen)
    fi = O.encode_image_fwd(sd, imgs)
    ft = torch.stack([O.encode_text_fwd(sd, tok(c))[0] for c in caps])
    logits = torch.einsum("bf,nbf->bn", fi, ft)
    # kernel-level: scores and predictions of the batch
    with torch.no_grad():
        gi = model.encode_image(imgs.to(DEV), normalize=True)
        gt = model.encode_text(torch.cat([tok(c) for c in caps]).to(DEV), normalize=True).view(2, B, -1)
    scores, pred = ops.mcq_scores(gi, gt)
    assert min_cosine(gi.cpu(), fi) >= 0.999 and min_cosine(gt.cpu().view(2 * B, -1), ft.view(2 * B, -1)) >= 0.999
    assert (scores.cpu() - logits).abs().max().item() < 2e-2
    clear = (logits[:, 0] - logits[:, 1]).abs() > 2e-2
    assert torch.equal(pred.cpu().long()[clear], logits.argmax(1)[clear])
    # through the caller: the cxr branch of mcq_eval, both CheXpert sets
    args = types.SimpleNamespace(device=DEV, batch_size=B, video=False, cxr_dataset=True)
    data = {k: types.SimpleNamespace(dataloader=_ListLoader([(imgs, caps, answers)], B)) for k in MQ.CXR_DATASETS}
    res = MQ.mcq_eval(model, data, 0, args, tok)
    assert set(res) == {"chexpert-binary-mcq-total_accuracy", "chexpert-affirmation-binary-mcq-total_accuracy"}
    ref_acc = float((logits.argmax(1) == answers).float().mean())
    for v in res.values():
        assert abs(v - ref_acc) <= float((~clear).sum()) / B + 1e-6
    _report("config5_vitb16_mcq", dict(score_max_abs=float((scores.cpu() - logits).abs().max()),
                                       unclear_margins=int((~clear).sum()), accuracy=res, accuracy_oracle=ref_acc))
    # one HIP graph per batch shape replays the same scorer (bench.py --workload mcq measures both)
    gs = MQ.GraphedScorer(model, B, 2, t["context_length"], 224, torch.device(DEV))
    s2, p2 = gs(imgs.to(DEV), torch.cat([tok(c) for c in caps]).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(p2, pred) and torch.equal(s2, scores)


def test_video_pooling_matches_oracle():
    model, sd = _model("neat-tiny", 0)
    clips = synth.synth_images(3 * 4, 32, 5).view(3, 4, 3, 32, 32)
    vm = oc.VideoCLIP(model)
    out = vm.encode_video(clips.to(DEV)).cpu()
    assert min_cosine(out, O.encode_video_fwd(sd, clips)) >= 0.999


def test_model_requires_gpu_and_library():
    m = oc.create_model("neat-tiny", device="cpu")
    with pytest.raises(RuntimeError):
        m.encode_text(torch.zeros(2, 16, dtype=torch.int64), normalize=True)
