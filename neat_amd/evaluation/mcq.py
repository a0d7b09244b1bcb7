"""Multiple-choice evaluation on HIP kernels (reference src/evaluation/mcq.py:10-146, 149-217, 296-389).

Per batch: images and all answer options are encoded by the HIP towers, then one kernel computes
``einsum('bf,nbf->bn')`` and the arg-max (mcq.py:92-97).  The bookkeeping of answer types is the reference's.
"""
from __future__ import annotations

import torch

from .. import ops


def _score_batch(model, image_tensor, captions, tokenizer, device):
    """-> predicted answer int64 [B] on the CPU.  ``captions``: list (options) of lists (batch) of strings."""
    batch_size, num_options = image_tensor.size(0), len(captions)
    toks = torch.cat([tokenizer(caption) for caption in captions]).to(device)
    if toks.dtype != torch.int64:
        toks = toks.long()
    with torch.no_grad():
        image_features = model.encode_image(image_tensor.to(device), normalize=True)
        text_features = model.encode_text(toks, normalize=True).view(num_options, batch_size, -1)
    _, pred = ops.mcq_scores(image_features, text_features)
    return pred.long().cpu()


class GraphedScorer:
    """The scorer of one batch shape as ONE HIP graph: encode_image + encode_text + normalise + einsum + arg-max are
    ~190 small launches at BASELINE config 5's batch (64 images x 2 captions), and the host's launch path, not the GPU, sets
    the latency.  Inputs are copied into static buffers, the graph is replayed, outputs live in static buffers (valid
    until the next call).  Weights are read through their pointers at replay time, so LayerNorm parameters adapted IN
    PLACE after the capture are honoured; a model whose parameter tensors are REPLACED must be captured again."""

    def __init__(self, model, batch: int, num_options: int, context_length: int, image_size: int, device):
        self.model, self.B, self.n = model, batch, num_options
        self.img = torch.zeros(batch, 3, image_size, image_size, device=device)
        self.tok = torch.zeros(num_options * batch, context_length, dtype=torch.int64, device=device)
        self.tok[:, 0] = 1                                        # any valid row for the warm-up passes
        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                             # first use of every kernel / packed weight: not captured
            for _ in range(2):
                self._run()
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.scores, self.pred = self._run()

    def _run(self):
        clip = self.model.clip if hasattr(self.model, "clip") else self.model
        packed = getattr(clip, "pack_tokens", False)
        clip.pack_tokens = False              # packed text batches size their buffers from a device-to-host copy: not capturable
        try:
            with torch.no_grad():
                fi = self.model.encode_image(self.img, normalize=True)
                ft = self.model.encode_text(self.tok, normalize=True).view(self.n, self.B, -1)
                return ops.mcq_scores(fi, ft)
        finally:
            clip.pack_tokens = packed

    def __call__(self, images, tokens):
        self.img.copy_(images, non_blocking=True)
        self.tok.copy_(tokens, non_blocking=True)
        self.graph.replay()
        return self.scores, self.pred


TYPES = ("positive", "negative", "hybrid")


def evaluate_model(model, dataloader, args, tokenizer, is_synthetic=False):
    """mcq.py:10-146: accuracy overall / per question type and the wrong-answer statistics.

    The reference counts item by item in Python; here every batch contributes (question type, predicted option, hit)
    triples and ONE bincount over a 3 x 4 x 2 table yields all the statistics it reports.  Option index -> caption type
    of a WRONG pick: synthetic sets {0: positive, 1: hybrid, 2: hybrid, 3: negative}, others {1: hybrid, 2: positive,
    3: negative} (option 0 is the correct caption there)."""
    device = getattr(args, "device", "cuda")
    model.eval()
    option_type = ({0: "positive", 1: "hybrid", 2: "hybrid", 3: "negative"} if is_synthetic
                   else {1: "hybrid", 2: "positive", 3: "negative"})
    n_opt = 4
    table = torch.zeros(len(TYPES) * n_opt * 2, dtype=torch.int64)
    for image_tensor, captions, correct_answer, correct_answer_type in dataloader:
        pred = _score_batch(model, image_tensor, captions, tokenizer, device)
        hit = (pred == torch.as_tensor(correct_answer).cpu()).long()
        qtype = torch.tensor([TYPES.index(t) for t in correct_answer_type], dtype=torch.int64)
        table += torch.bincount((qtype * n_opt + pred) * 2 + hit, minlength=table.numel())
    table = table.view(len(TYPES), n_opt, 2)                  # [question type, predicted option, hit]
    asked = table.sum(dim=(1, 2))
    right = table[:, :, 1].sum(dim=1)
    wrong_by_option = table[:, :, 0].sum(dim=0)               # how often each option was picked wrongly
    wrong_answer_counts = {k: int(wrong_by_option[k]) for k in option_type}
    predictions_by_type = {t: int(right[i]) for i, t in enumerate(TYPES)}
    wrong_answers_by_question_type = {t: {u: 0 for u in TYPES} for t in TYPES}
    for k, wrong_type in option_type.items():
        predictions_by_type[wrong_type] += int(wrong_by_option[k])
        for i, t in enumerate(TYPES):
            wrong_answers_by_question_type[t][wrong_type] += int(table[i, k, 0])
    stray = int(wrong_by_option.sum()) - sum(wrong_answer_counts.values())
    if stray:
        raise KeyError(f"mcq.evaluate_model: {stray} wrong prediction(s) on an option without a caption type "
                       f"(option 0 is the correct caption outside the synthetic sets)")
    total_wrong = sum(wrong_answer_counts.values())
    most_common = max(wrong_answer_counts, key=wrong_answer_counts.get)
    result = {"total_accuracy": int(right.sum()) / len(dataloader.dataset)}
    for i, t in enumerate(TYPES):
        result[f"{t}_accuracy"] = int(right[i]) / int(asked[i]) if int(asked[i]) > 0 else float("nan")
    result.update({
        "most_common_wrong_answer_type": option_type[most_common],
        # the reference divides by zero when nothing is wrong (mcq.py:134); report NaN instead of raising
        "wrong_answer_percentages": {option_type[k]: (v / total_wrong * 100 if total_wrong else float("nan"))
                                     for k, v in wrong_answer_counts.items()},
        "predictions_by_type": predictions_by_type,
        "wrong_answers_by_question_type": wrong_answers_by_question_type,
    })
    return result


def evaluate_binary_mcq_model(model, dataloader, args, tokenizer):
    """mcq.py:149-217: two options per image, total accuracy."""
    device = getattr(args, "device", "cuda")
    model.eval()
    correct = 0
    for image_tensor, captions, correct_answer in dataloader:
        pred = _score_batch(model, image_tensor, captions, tokenizer, device)
        correct += int((pred == torch.as_tensor(correct_answer).cpu()).sum())
    return {"total_accuracy": correct / len(dataloader.dataset)}


CXR_DATASETS = ("chexpert-binary-mcq", "chexpert-affirmation-binary-mcq")


def mcq_eval(model, data, epoch, args, tokenizer=None):
    """mcq.py:296-389: result keys per dataset present in ``data``.

    ``args.cxr_dataset`` (mcq.py:330-339, BASELINE config 5): the two CheXpert binary sets, one ``-total_accuracy`` key
    each.  The reference scores them with a HuggingFace PLIP model (``evaluate_binary_mcq_model_hf`` :220-294: per image
    ``einsum('bd,bnd->bn')`` over its two captions + arg-max); with an open_clip-layout model -- what this package
    adapts, and what ``replace_text_layernorm_params`` hands over to PLIP -- the same scoring is
    ``evaluate_binary_mcq_model`` (:149-217), which is what runs here."""
    results = {}
    if getattr(args, "cxr_dataset", False) and not getattr(args, "video", False):
        for name in CXR_DATASETS:
            if name in data:
                m = evaluate_binary_mcq_model(model, data[name].dataloader, args, tokenizer)
                results[f"{name}-total_accuracy"] = m["total_accuracy"]
        return results
    names = ["msrvtt-mcq"] if getattr(args, "video", False) else ["synthetic-mcq", "coco-mcq", "voc2007-mcq"]
    for name in names:
        if name not in data:
            continue
        m = evaluate_model(model, data[name].dataloader, args, tokenizer, is_synthetic=(name == "synthetic-mcq"))
        for key in ("total_accuracy", "positive_accuracy", "negative_accuracy", "hybrid_accuracy",
                    "most_common_wrong_answer_type", "predictions_by_type", "wrong_answers_by_question_type"):
            results[f"{name}-{key}"] = m[key]
        results[f"{name}-wrong_answer_percentages"] = list(m["wrong_answer_percentages"].items())
    return results
