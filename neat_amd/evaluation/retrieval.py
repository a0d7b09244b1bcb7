"""Retrieval scoring on HIP kernels, with the reference's function names and semantics
(reference src/evaluation/retrieval.py:186-214, 219-257, 280-333)."""
from __future__ import annotations

import torch

from .. import ops


def recall_at_k(scores, positive_pairs, k):
    """retrieval.py:280-301: |topk(scores_i, k) ∩ positives_i| / |positives_i| per row.

    ``scores`` may be a transposed view (the reference passes ``scores.T``, :248-250): it is read in place.
    One wavefront selects the k winners per row in registers; no [b,k,N] one-hot is built."""
    _, idx = ops.topk_rows(scores, k)
    return ops.recall_from_topk(idx, positive_pairs)


def batchify(func, X, Y, batch_size, device, *args, **kwargs):
    """retrieval.py:304-333.  Kept for API compatibility; the kernels take the whole matrix at once, so
    the batches are only a view split (results are identical for any batch_size)."""
    results = []
    for start in range(0, len(X), batch_size):
        end = start + batch_size
        x = X[start:end].to(device)
        y = Y[start:end].to(device)
        results.append(func(x, y, *args, **kwargs).cpu())
    return torch.cat(results)


def recall_metrics(scores, positive_pairs, recall_k_list=(1, 5, 10)):
    """The six recall numbers of evaluate_model_with_similarity (retrieval.py:243-250) in one pass per
    direction and k, whole matrix at once."""
    metrics = {}
    for k in recall_k_list:
        metrics[f"image_retrieval_recall@{k}"] = (recall_at_k(scores, positive_pairs, k) > 0).float().mean().item()
        metrics[f"text_retrieval_recall@{k}"] = (recall_at_k(scores.T, positive_pairs.T, k) > 0).float().mean().item()
    return metrics


def evaluate_model_with_similarity(dataloader, args, scores, recall_k_list=(1, 5, 10), texts_image_index=None):
    """retrieval.py:219-257 ("online" evaluation on the TTA score matrix).  ``texts_image_index`` may be
    given directly; otherwise it is derived from the loader order like the reference does (:229-236),
    which iterates the loader only to count captions per image."""
    if texts_image_index is None:
        texts_image_index, start = [], 0
        for batch in dataloader:
            batch_texts = batch[1]
            for off, texts in enumerate(batch_texts):
                texts_image_index.extend([start + off] * len(texts))
            start += len(batch[0])
    positive_pairs = torch.zeros_like(scores, dtype=torch.bool)
    positive_pairs[torch.arange(len(scores), device=scores.device),
                   torch.as_tensor(texts_image_index, device=scores.device)] = True
    return recall_metrics(scores, positive_pairs, recall_k_list)
