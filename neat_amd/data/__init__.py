"""Datasets and loaders of the TTA path (reference src/data/__init__.py:12-73), MI355X-first.

Same entry points (``create_dataset``, ``create_sampler``, ``create_loader``) and the same dataset attributes the TTA
loop reads (``.dataset.text`` / ``.dataset.image``).  Differences, on purpose:
  * the image transform does not run in the dataset: the dataset yields decoded uint8 [H, W, 3] images and
    ``GpuImageLoader`` resizes / normalises a whole batch of mixed-size images on the GPU in one launch set
    (``neat_image_preprocess_batch``, Pillow-exact bicubic),
    so a gallery batch arrives on the device as fp32 [b, 3, S, S]; the MSR-VTT gallery dataset yields the sampled frames
    as uint8 [T, H, W, 3] and ``GpuVideoLoader`` runs the video transform (/ 255, antialiased bicubic as torch does it on
    float tensors, Normalize: ``neat_frames_preprocess``) -> fp32 [b, T, 3, S, S];
  * ``create_loader`` passes ``shuffle`` only when there is no sampler (the reference passes both, which
    ``DataLoader`` rejects -- SURVEY.md section 0).
"""
import torch
from torch.utils.data import DataLoader

from .. import ops
from .coco_negated_dataset import coco_negated_retrieval_eval_image, coco_negated_retrieval_eval_text
from .msrvtt_negated_dataset import msrvtt_negated_retrieval_eval_image, msrvtt_negated_retrieval_eval_text
from .utils import pre_caption

__all__ = ["create_dataset", "create_sampler", "create_loader", "GpuImageLoader", "GpuVideoLoader", "pre_caption",
           "coco_negated_retrieval_eval_image", "coco_negated_retrieval_eval_text",
           "msrvtt_negated_retrieval_eval_image", "msrvtt_negated_retrieval_eval_text"]


def create_dataset(dataset_name, input_filename, resize=384):
    kw = dict(csv_file=input_filename, sep=',', img_key='filepath', caption_key_pos='pos_captions',
              caption_key_neg='neg_captions')
    if dataset_name == 'coco_negated':
        text = coco_negated_retrieval_eval_text(None, **kw)
        image = coco_negated_retrieval_eval_image(None, **kw)
    elif dataset_name == 'msrvtt_negated':
        text = msrvtt_negated_retrieval_eval_text(None, **kw)
        image = msrvtt_negated_retrieval_eval_image(None, **kw)
    else:
        raise NotImplementedError(f"create_dataset: unknown dataset '{dataset_name}' (coco_negated, msrvtt_negated)")
    image.resize = text.resize = resize
    return text, image


def create_sampler(datasets, shuffles, num_tasks, global_rank):
    return [torch.utils.data.DistributedSampler(d, num_replicas=num_tasks, rank=global_rank, shuffle=s)
            for d, s in zip(datasets, shuffles)]


def _list_collate(batch):
    return [b[0] for b in batch], torch.tensor([b[1] for b in batch], dtype=torch.int64)


class GpuImageLoader:
    """Iterates a DataLoader of (uint8 [H, W, 3] image, index) and yields (fp32 [b, 3, S, S] on ``device``, index)."""

    def __init__(self, loader, size, device=None):
        self.loader, self.size = loader, size
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, index in self.loader:
            on_dev = [im.to(self.device, non_blocking=True).contiguous() for im in images]
            yield ops.image_preprocess_batch(on_dev, self.size), index


class GpuVideoLoader(GpuImageLoader):
    """Iterates a DataLoader of (uint8 [T, H, W, 3] frames, index) and yields (fp32 [b, T, 3, S, S] on ``device``, index).
    Clips of one size (the MSR-VTT case) go through ONE call of the frame kernel as a [b*T, H, W, 3] stack."""

    def __iter__(self):
        for clips, index in self.loader:
            shapes = {tuple(c.shape) for c in clips}
            if len(shapes) == 1:
                T = clips[0].shape[0]
                stack = torch.stack(clips).to(self.device, non_blocking=True)
                out = ops.frames_preprocess(stack.view(-1, *stack.shape[2:]), self.size)
                yield out.view(len(clips), T, 3, self.size, self.size), index
            else:
                outs = [ops.frames_preprocess(c.to(self.device, non_blocking=True).contiguous(), self.size) for c in clips]
                yield torch.stack(outs), index


def create_loader(datasets, samplers, batch_size, num_workers, is_trains, collate_fns, device=None):
    loaders = []
    for dataset, sampler, bs, n_worker, is_train, collate_fn in zip(datasets, samplers, batch_size, num_workers,
                                                                    is_trains, collate_fns):
        is_image = isinstance(dataset, coco_negated_retrieval_eval_image)
        is_video = isinstance(dataset, msrvtt_negated_retrieval_eval_image)
        loader = DataLoader(dataset, batch_size=bs, num_workers=n_worker, pin_memory=True, sampler=sampler,
                            shuffle=(sampler is None), collate_fn=_list_collate if (is_image and collate_fn is None)
                            else collate_fn, drop_last=False)
        if is_image:
            loader = (GpuVideoLoader if is_video else GpuImageLoader)(loader, getattr(dataset, "resize", 224), device)
        loaders.append(loader)
    return loaders
