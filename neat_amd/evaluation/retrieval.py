"""Retrieval scoring on HIP kernels, with the reference's function names and semantics
(reference src/evaluation/retrieval.py:11-89, 186-214, 219-257, 259-278, 280-333)."""
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


def median_rank(scores, positive_pairs):
    """retrieval.py:186-214: median over rows of the (1-based) rank of the best-ranked positive; rows without a
    positive are dropped; ``torch.median`` semantics (lower middle element).  The reference sorts every row and
    scatters ranks with a Python loop; here one kernel counts, per row, the scores that beat the best positive."""
    ranks = ops.best_positive_rank(scores, positive_pairs).cpu().tolist()
    ranks = sorted(r for r in ranks if r > 0)
    if not ranks:
        return float("nan")
    return float(ranks[(len(ranks) - 1) // 2])


def _median_of_ranks(ranks):
    ranks = sorted(r for r in ranks.cpu().tolist() if r > 0)
    return float(ranks[(len(ranks) - 1) // 2]) if ranks else float("nan")


def recall_metrics(scores, positive_pairs=None, recall_k_list=(1, 5, 10), with_median_rank=False, texts_image_index=None):
    """The recall numbers of evaluate_model_with_similarity (retrieval.py:243-250) and (optionally) the two medR entries
    (:254-255).  The reference makes one pass over the score matrix per direction AND per k (six passes, each building a
    [b, k, N] one-hot); here each direction is ONE top-max(k) pass and every smaller k is a prefix of that list.

    ``texts_image_index`` (caption -> image, what the reference builds ``positive_pairs`` from, :229-240): the positives
    are then taken from this list and no dense bool matrix is read at all; otherwise ``positive_pairs`` is used."""
    ks = sorted(int(k) for k in recall_k_list)
    kmax = ks[-1]
    metrics = {}
    n_txt, n_img = scores.shape
    _, idx_t2i = ops.topk_rows(scores, min(kmax, n_img))
    _, idx_i2t = ops.topk_rows(scores.T, min(kmax, n_txt))
    if texts_image_index is not None:
        owner = torch.as_tensor(texts_image_index, device=scores.device).to(torch.int32).contiguous()
        per_image = torch.bincount(owner.long(), minlength=n_img).to(torch.int32)
        r_t2i = ops.recall_owner_topk(idx_t2i, ks, row_owner=owner)
        r_i2t = ops.recall_owner_topk(idx_i2t, ks, col_owner=owner, npos=per_image)
        for q, k in enumerate(ks):
            metrics[f"image_retrieval_recall@{k}"] = (r_t2i[q] > 0).float().mean().item()
            metrics[f"text_retrieval_recall@{k}"] = (r_i2t[q] > 0).float().mean().item()
        if with_median_rank:
            metrics["image_retrieval_medR"] = _median_of_ranks(ops.best_positive_rank_owner(scores, row_owner=owner))
            metrics["text_retrieval_medR"] = _median_of_ranks(ops.best_positive_rank_owner(scores.T, col_owner=owner))
        return metrics
    if positive_pairs is None:
        raise RuntimeError("recall_metrics: pass positive_pairs or texts_image_index")
    for k in ks:
        r_t = ops.recall_from_topk(idx_t2i[:, :k].contiguous(), positive_pairs)
        r_i = ops.recall_from_topk(idx_i2t[:, :k].contiguous(), positive_pairs.T)
        metrics[f"image_retrieval_recall@{k}"] = (r_t > 0).float().mean().item()
        metrics[f"text_retrieval_recall@{k}"] = (r_i > 0).float().mean().item()
    if with_median_rank:
        metrics["image_retrieval_medR"] = median_rank(scores, positive_pairs)
        metrics["text_retrieval_medR"] = median_rank(scores.T, positive_pairs.T)
    return metrics


def recall_from_embeddings(texts_emb, images_emb, texts_image_index, recall_k_list=(1, 5, 10)):
    """The recall entries of ``evaluate_model`` (retrieval.py:70 + :243-250) straight from the embeddings, without the
    [N_text, N_image] score matrix in HBM: one fused similarity + top-k call per direction (``neat_sim_topk``; the scores
    only exist in a cache-sized chunk) and the owner-vector recall.  Same numbers as ``recall_metrics`` on the
    materialised matrix (same kernels, bit-identical lists); medR needs whole score rows and is not produced here."""
    ks = sorted(int(k) for k in recall_k_list)
    n_txt, n_img = texts_emb.shape[0], images_emb.shape[0]
    owner = torch.as_tensor(texts_image_index, device=texts_emb.device).to(torch.int32).contiguous()
    per_image = torch.bincount(owner.long(), minlength=n_img).to(torch.int32)
    _, idx_t2i = ops.sim_topk(texts_emb, images_emb, min(ks[-1], n_img))
    _, idx_i2t = ops.sim_topk(images_emb, texts_emb, min(ks[-1], n_txt))
    r_t2i = ops.recall_owner_topk(idx_t2i, ks, row_owner=owner)
    r_i2t = ops.recall_owner_topk(idx_i2t, ks, col_owner=owner, npos=per_image)
    metrics = {}
    for q, k in enumerate(ks):
        metrics[f"image_retrieval_recall@{k}"] = (r_t2i[q] > 0).float().mean().item()
        metrics[f"text_retrieval_recall@{k}"] = (r_i2t[q] > 0).float().mean().item()
    return metrics


def dataloader_with_indices(dataloader):
    """retrieval.py:259-278"""
    start = 0
    for batch in dataloader:
        end = start + len(batch[0])
        yield (*batch, torch.arange(start, end))
        start = end


def evaluate_model(model, dataloader, args, tokenizer, recall_k_list=(1, 5, 10)):
    """retrieval.py:11-89 ("offline" evaluation): re-encode every image and caption with the current (adapted)
    parameters, score all pairs, recall@k both ways and medR.  Embeddings stay on the GPU (the reference moves
    them to the CPU and multiplies there); the similarity is the exact-fp32 GEMM that also feeds top-k."""
    device = getattr(args, "device", "cuda")
    img_emb, txt_emb, texts_image_index = [], [], []
    with torch.no_grad():
        for batch_images, batch_texts, inds in dataloader_with_indices(dataloader):
            toks = tokenizer([text for texts in batch_texts for text in texts]).to(device)
            if toks.dtype != torch.int64:
                toks = toks.long()
            img_emb.append(model.encode_image(batch_images.to(device), normalize=True))
            txt_emb.append(model.encode_text(toks, normalize=True))
            texts_image_index.extend(int(ind) for ind, texts in zip(inds, batch_texts) for _ in texts)
    images_emb = torch.cat(img_emb).contiguous()
    texts_emb = torch.cat(txt_emb).contiguous()
    scores = ops.gemm_f32_nt(texts_emb, images_emb)
    return recall_metrics(scores, None, recall_k_list, with_median_rank=True, texts_image_index=texts_image_index)


def retrieval_eval(model, data, args, scores, tokenizer=None):
    """retrieval.py:91-183: the result keys the reference's CLI logs, for the datasets present in ``data``
    (objects with a ``.dataloader``)."""
    results = {}

    def put(prefix, m):
        for d, key in (("t2i", "image_retrieval"), ("i2t", "text_retrieval")):
            for k in (1, 5, 10):
                # the reference spells exactly one key with an underscore (retrieval.py:117): 'msrvtt-i2t_R@1'
                sep = "_" if (prefix, d, k) == ("msrvtt", "i2t", 1) else "-"
                results[f"{prefix}-{d}{sep}R@{k}"] = m[f"{key}_recall@{k}"]

    video = getattr(args, "video", False)
    plain, negated = ("msrvtt-retrieval", "msrvtt-negated-retrieval") if video else ("coco-retrieval", "coco-negated-retrieval")
    stem = "msrvtt" if video else "coco"
    if plain in data:
        put(stem, evaluate_model(model, data[plain].dataloader, args, tokenizer))
    if negated in data:
        put(f"online-{stem}-negated", evaluate_model_with_similarity(data[negated].dataloader, args, scores))
        put(f"offline-{stem}-negated", evaluate_model(model, data[negated].dataloader, args, tokenizer))
    return results


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
    return recall_metrics(scores, None, recall_k_list, with_median_rank=True, texts_image_index=texts_image_index)
