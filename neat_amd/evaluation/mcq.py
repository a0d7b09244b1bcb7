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


def evaluate_model(model, dataloader, args, tokenizer, is_synthetic=False):
    """mcq.py:10-146: accuracy overall / per question type and the wrong-answer statistics."""
    device = getattr(args, "device", "cuda")
    model.eval()
    total_questions = len(dataloader.dataset)
    types = ("positive", "negative", "hybrid")
    correct_answers_by_type = {t: 0 for t in types}
    total_questions_by_type = {t: 0 for t in types}
    if is_synthetic:
        wrong_answer_to_type = {0: "positive", 1: "hybrid", 2: "hybrid", 3: "negative"}
    else:
        wrong_answer_to_type = {1: "hybrid", 2: "positive", 3: "negative"}
    wrong_answer_counts = {k: 0 for k in wrong_answer_to_type}
    predictions_by_type = {t: 0 for t in types}
    wrong_answers_by_question_type = {t: {u: 0 for u in types} for t in types}
    correct_answers_sum = 0
    for image_tensor, captions, correct_answer, correct_answer_type in dataloader:
        pred = _score_batch(model, image_tensor, captions, tokenizer, device)
        correct_answer = torch.as_tensor(correct_answer).cpu()
        correct_answers_sum += int((pred == correct_answer).sum())
        for i in range(image_tensor.size(0)):
            answer_type = correct_answer_type[i]
            total_questions_by_type[answer_type] += 1
            if pred[i] == correct_answer[i]:
                correct_answers_by_type[answer_type] += 1
                predictions_by_type[answer_type] += 1
            else:
                wrong_type = wrong_answer_to_type[int(pred[i])]
                wrong_answer_counts[int(pred[i])] += 1
                predictions_by_type[wrong_type] += 1
                wrong_answers_by_question_type[answer_type][wrong_type] += 1

    def acc(t):
        return correct_answers_by_type[t] / total_questions_by_type[t] if total_questions_by_type[t] > 0 else float("nan")

    most_common = max(wrong_answer_counts, key=wrong_answer_counts.get)
    total_wrong = sum(wrong_answer_counts.values())
    return {
        "total_accuracy": correct_answers_sum / total_questions,
        "positive_accuracy": acc("positive"),
        "negative_accuracy": acc("negative"),
        "hybrid_accuracy": acc("hybrid"),
        "most_common_wrong_answer_type": wrong_answer_to_type[most_common],
        # the reference divides by zero when nothing is wrong (mcq.py:134); report NaN instead of raising
        "wrong_answer_percentages": {wrong_answer_to_type[k]: (v / total_wrong * 100 if total_wrong else float("nan"))
                                     for k, v in wrong_answer_counts.items()},
        "predictions_by_type": predictions_by_type,
        "wrong_answers_by_question_type": wrong_answers_by_question_type,
    }


def evaluate_binary_mcq_model(model, dataloader, args, tokenizer):
    """mcq.py:149-217: two options per image, total accuracy."""
    device = getattr(args, "device", "cuda")
    model.eval()
    correct = 0
    for image_tensor, captions, correct_answer in dataloader:
        pred = _score_batch(model, image_tensor, captions, tokenizer, device)
        correct += int((pred == torch.as_tensor(correct_answer).cpu()).sum())
    return {"total_accuracy": correct / len(dataloader.dataset)}


def mcq_eval(model, data, epoch, args, tokenizer=None):
    """mcq.py:296-389: result keys per dataset present in ``data`` (the HF-PLIP CheXpert branch is out of scope)."""
    results = {}
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
