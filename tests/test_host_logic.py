"""CPU: host-side logic of the callers either side of the hot path -- the LayerNorm hand-over to a HuggingFace-shaped
text tower (reference tta_utils.py:287-341), the MCQ statistics (reference evaluation/mcq.py:38-146) and the model factory's
activation guard.  No kernels run here: the scoring call is replaced by scripted predictions."""
import types

import pytest
import torch

from neat_amd import open_clip as oc, synth, tta_utils
from neat_amd.evaluation import mcq as MQ


# ------------------------------------------------------------------ LN export / hand-over
class _HFLayer(torch.nn.Module):
    def __init__(self, c, with_ln2=True):
        super().__init__()
        self.layer_norm1 = torch.nn.LayerNorm(c)
        if with_ln2:
            self.layer_norm2 = torch.nn.LayerNorm(c)


class _HFClip(torch.nn.Module):
    """The attribute layout of transformers.CLIPModel that replace_text_layernorm_params touches."""

    def __init__(self, c, n, dtype=torch.float32):
        super().__init__()
        self.text_model = torch.nn.Module()
        self.text_model.encoder = torch.nn.Module()
        self.text_model.encoder.layers = torch.nn.ModuleList([_HFLayer(c, with_ln2=(i != 1)) for i in range(n)])
        self.text_model.final_layer_norm = torch.nn.LayerNorm(c)
        self.to(dtype)


def _adapted_tiny():
    model = oc.create_model("neat-tiny", device="cpu")
    model.load_state_dict(synth.synth_state_dict("neat-tiny", 0))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for blk in model.transformer.resblocks:                     # pretend TTA moved every LN parameter
            for ln in (blk.ln_1, blk.ln_2):
                ln.weight.add_(torch.randn(ln.weight.shape, generator=g) * 0.01)
                ln.bias.add_(torch.randn(ln.bias.shape, generator=g) * 0.01)
    return model


def test_replace_text_layernorm_params_into_hf_shaped_module():
    model = _adapted_tiny()
    n = len(model.transformer.resblocks)
    C = model.transformer.resblocks[0].ln_1.weight.numel()
    hf = _HFClip(C, n, dtype=torch.float64)                        # another dtype: values are cast, storage is kept
    before_final = hf.text_model.final_layer_norm.weight.clone()
    ptr = hf.text_model.encoder.layers[0].layer_norm1.weight.data_ptr()
    assert tta_utils.replace_text_layernorm_params(hf, model) == n
    for i, (hl, blk) in enumerate(zip(hf.text_model.encoder.layers, model.transformer.resblocks)):
        assert torch.equal(hl.layer_norm1.weight, blk.ln_1.weight.double()) and torch.equal(hl.layer_norm1.bias, blk.ln_1.bias.double())
        if i != 1:                                                  # layer 1 has no layer_norm2: skipped, not an error
            assert torch.equal(hl.layer_norm2.weight, blk.ln_2.weight.double())
    assert hf.text_model.encoder.layers[0].layer_norm1.weight.data_ptr() == ptr          # in place
    # CLIP keeps ln_final outside model.transformer (open_clip/model.py:245), so -- exactly as in the reference -- the HF
    # final_layer_norm is left alone
    assert torch.equal(hf.text_model.final_layer_norm.weight, before_final)
    # a tower that does carry transformer.ln_final hands it over too
    model.transformer.ln_final = torch.nn.LayerNorm(C)
    torch.nn.init.normal_(model.transformer.ln_final.weight)
    tta_utils.replace_text_layernorm_params(hf, model)
    assert torch.equal(hf.text_model.final_layer_norm.weight, model.transformer.ln_final.weight.double())


def test_export_text_layernorm_params_round_trip():
    model = _adapted_tiny()
    d = tta_utils.export_text_layernorm_params(model)
    _, names = oc.collect_params(model, False)
    assert sorted(d) == sorted("transformer." + n for n in names)
    fresh = oc.create_model("neat-tiny", device="cpu")
    fresh.load_state_dict(synth.synth_state_dict("neat-tiny", 0))
    missing = fresh.load_state_dict(d, strict=False)
    assert not missing.unexpected_keys
    for a, b in zip(oc.collect_params(fresh, False)[0], oc.collect_params(model, False)[0]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------ MCQ statistics
def _count_item_by_item(preds, answers, qtypes, option_type):
    """Plain restatement of what the reference reports (mcq.py:38-146), one item at a time."""
    T = ("positive", "negative", "hybrid")
    right = {t: 0 for t in T}; asked = {t: 0 for t in T}; pbt = {t: 0 for t in T}
    wac = {k: 0 for k in option_type}; wbq = {t: {u: 0 for u in T} for t in T}
    for p, a, q in zip(preds, answers, qtypes):
        asked[q] += 1
        if p == a:
            right[q] += 1; pbt[q] += 1
        else:
            wac[p] += 1; pbt[option_type[p]] += 1; wbq[q][option_type[p]] += 1
    return right, asked, pbt, wac, wbq


@pytest.mark.parametrize("synthetic", [True, False])
def test_mcq_statistics_equal_item_by_item_count(monkeypatch, synthetic):
    g = torch.Generator().manual_seed(5 + synthetic)
    n, bs = 57, 8
    answers = torch.randint(0, 4, (n,), generator=g) if synthetic else torch.zeros(n, dtype=torch.int64)
    preds = torch.randint(0, 4, (n,), generator=g)
    if not synthetic:
        preds[::3] = 0
    qtypes = [("positive", "negative", "hybrid")[int(i)] for i in torch.randint(0, 3, (n,), generator=g)]
    batches, cursor = [], []
    for i in range(0, n, bs):
        batches.append((torch.zeros(len(preds[i:i + bs]), 1), None, answers[i:i + bs], qtypes[i:i + bs]))
        cursor.append(preds[i:i + bs])
    it = iter(cursor)
    monkeypatch.setattr(MQ, "_score_batch", lambda *a, **k: next(it))

    class L:
        dataset = list(range(n))

        def __iter__(self):
            return iter(batches)

    class M:
        def eval(self):
            pass

    out = MQ.evaluate_model(M(), L(), types.SimpleNamespace(device="cpu"), None, is_synthetic=synthetic)
    option_type = ({0: "positive", 1: "hybrid", 2: "hybrid", 3: "negative"} if synthetic
                   else {1: "hybrid", 2: "positive", 3: "negative"})
    right, asked, pbt, wac, wbq = _count_item_by_item(preds.tolist(), answers.tolist(), qtypes, option_type)
    assert out["total_accuracy"] == sum(right.values()) / n
    for t in right:
        assert out[f"{t}_accuracy"] == right[t] / asked[t]
    assert out["predictions_by_type"] == pbt and out["wrong_answers_by_question_type"] == wbq
    tot = sum(wac.values())
    # the reference keys this dict by caption type, so the two "hybrid" options of the synthetic sets collapse: last wins
    want = {}
    for k, v in wac.items():
        want[option_type[k]] = v / tot * 100
    assert out["wrong_answer_percentages"] == want
    assert out["most_common_wrong_answer_type"] == option_type[max(wac, key=wac.get)]
    assert sum(pbt.values()) == n


def test_mcq_wrong_pick_of_option_zero_outside_synthetic_sets_raises(monkeypatch):
    monkeypatch.setattr(MQ, "_score_batch", lambda *a, **k: torch.tensor([0, 1]))

    class L:
        dataset = [0, 1]

        def __iter__(self):
            return iter([(torch.zeros(2, 1), None, torch.tensor([1, 1]), ["positive", "negative"])])

    class M:
        def eval(self):
            pass

    with pytest.raises(KeyError):
        MQ.evaluate_model(M(), L(), types.SimpleNamespace(device="cpu"), None, is_synthetic=False)


def test_mcq_eval_cxr_branch_reports_the_two_chexpert_keys(monkeypatch):
    """args.cxr_dataset (reference mcq.py:330-339, BASELINE config 5): binary scorer, two result keys, nothing else."""
    preds = {"chexpert-binary-mcq": [torch.tensor([0, 1, 1, 0]), torch.tensor([1, 1])],
             "chexpert-affirmation-binary-mcq": [torch.tensor([1, 1, 1, 1]), torch.tensor([0, 0])]}
    answers = [torch.tensor([0, 1, 0, 0]), torch.tensor([1, 0])]                 # 4 / 6 and 2 / 6 right
    cursor = {}
    monkeypatch.setattr(MQ, "_score_batch", lambda model, image, captions, tok, dev: next(cursor[image]))

    class L:
        dataset = list(range(6))

        def __init__(self, name):
            self.name = name

        def __iter__(self):
            cursor[self.name] = iter(preds[self.name])
            caps = [["no evidence of edema"] * 4, ["edema is present"] * 4]
            return iter([(self.name, caps, answers[0]), (self.name, [c[:2] for c in caps], answers[1])])

    class M:
        def eval(self):
            pass

    data = {n: types.SimpleNamespace(dataloader=L(n)) for n in preds}
    data["coco-mcq"] = None                                                       # ignored on this branch
    out = MQ.mcq_eval(M(), data, 0, types.SimpleNamespace(device="cpu", cxr_dataset=True, video=False), None)
    assert out == {"chexpert-binary-mcq-total_accuracy": 4 / 6, "chexpert-affirmation-binary-mcq-total_accuracy": 2 / 6}
    only = MQ.mcq_eval(M(), {"chexpert-binary-mcq": data["chexpert-binary-mcq"]}, 0,
                       types.SimpleNamespace(device="cpu", cxr_dataset=True, video=False), None)
    assert list(only) == ["chexpert-binary-mcq-total_accuracy"]


# ------------------------------------------------------------------ factory
def test_pretrained_checkpoint_needs_an_explicit_quick_gelu(tmp_path):
    sd = synth.synth_state_dict("neat-tiny", 0)
    ck = tmp_path / "w.pt"
    torch.save(sd, ck)
    with pytest.raises(RuntimeError, match="QuickGELU"):
        oc.create_model("neat-tiny", pretrained=str(ck), device="cpu")
    m = oc.create_model("neat-tiny", pretrained=str(ck), device="cpu", force_quick_gelu=True)
    assert torch.equal(m.text_projection.detach(), sd["text_projection"])
    # an OpenAI-format state dict (the three scalar entries beside the weights) says QuickGELU by itself, as in the
    # reference (open_clip/model.py:450-506): create_model_and_transforms(name, pretrained=<OpenAI .pt>) works unchanged
    sd_openai = dict(sd, input_resolution=torch.tensor(32), context_length=torch.tensor(16), vocab_size=torch.tensor(512))
    ck2 = tmp_path / "openai.pt"
    torch.save(sd_openai, ck2)
    m2, _, _ = oc.create_model_and_transforms("neat-tiny", pretrained=str(ck2), device="cpu")
    assert torch.equal(m2.text_projection.detach(), sd["text_projection"])


# ------------------------------------------------------------------ bench.py starts its own ranks
def _run_bench(*argv, env_extra=None):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=240)


def test_host_cu_seqlens_is_the_prefix_sum_of_first_argmax_plus_one():
    """train_TTA.host_cu_seqlens (the packed layout from host tokens): rows ordered [text, pos, inv, neg], EOT = first
    maximum of a row (torch.argmax on the reference's side, open_clip/transformer.py:555-558)."""
    from neat_amd import train_TTA as T
    g = torch.Generator().manual_seed(3)
    sets = [torch.randint(0, 50, (7, 13), generator=g) for _ in range(4)]
    sets[0][1, 3] = sets[0][1, 9] = 99                  # a tie: the first position counts
    sets[3][0] = 0
    text, pos, neg, inv = sets
    cu = T.host_cu_seqlens((text, pos, neg, inv))
    cat = torch.cat([text, pos, inv, neg])
    want = [0]
    for row in cat.tolist():
        want.append(want[-1] + row.index(max(row)) + 1)
    assert cu.dtype == torch.int32 and cu.tolist() == want
    assert cu[2] - cu[1] == 4 and cu[15] - cu[14] == 1       # inv (the 4th argument) is the 3rd block of rows


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: two ranks under torch.distributed.run, rank 0's one JSON
    line relayed, the child's status returned (reference intent: main_TTA.py:245-256, one process per GPU)."""
    import json
    r = _run_bench("--gpus", "2", "--launcher-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == dict(selftest=True, n_gpus=2, rank_sum=1.0)


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_gpus_n_refuses_without_n_gpus():
    """Fewer GPUs than ranks: a clear message and a non-zero status, never an `n_gpus: 1` line."""
    r = _run_bench("--gpus", "2")
    assert r.returncode == 2
    assert "needs 2 visible GPUs" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
