"""The two image transforms ``create_model_and_transforms`` hands back (reference open_clip/factory.py:445-510 returns
``preprocess_train, preprocess_val`` built by open_clip/transform.py:274-407 and the driver passes them on to its datasets).

MI355X-first split of that work: the HOST side of a transform only turns whatever the dataset decoded (a PIL image, a numpy
array, a uint8 tensor) into a contiguous uint8 [H, W, 3] tensor; resize + ToTensor + Normalize run on the GPU for a whole
batch at once (``neat_amd.data.GpuImageLoader`` -> ``neat_image_preprocess_batch``, Pillow-exact bicubic).  The transform
carries the parameters the loader needs (``size``, ``mean``, ``std``), so a dataset written for the reference's transforms
works unchanged once its DataLoader is wrapped: ``GpuImageLoader(loader, transform.size)``.
"""
import numpy as np
import torch

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


class GpuImageTransform:
    """Host half of the image transform: anything image-like -> uint8 [H, W, 3] tensor (RGB).  ``resize_mode`` is
    'squash' (Resize((S, S)), what the TTA gallery datasets use, reference data/__init__.py:14-20)."""

    def __init__(self, size=224, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD, is_train=False):
        self.size = int(size[0] if isinstance(size, (tuple, list)) else size)
        self.mean, self.std = tuple(mean), tuple(std)
        self.is_train = is_train
        self.interpolation, self.resize_mode = "bicubic", "squash"

    def __call__(self, image):
        if isinstance(image, torch.Tensor):
            t = image
        elif isinstance(image, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(image))
        else:                                              # PIL.Image
            t = torch.from_numpy(np.asarray(image.convert("RGB")).copy())
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise TypeError(f"GpuImageTransform: expected an RGB image as uint8 [H, W, 3], got {t.dtype} {tuple(t.shape)}")
        return t.contiguous()

    def __repr__(self):
        return (f"GpuImageTransform(size={self.size}, mean={self.mean}, std={self.std}, interpolation='bicubic', "
                f"resize_mode='squash'; resize/normalise run in neat_amd.data.GpuImageLoader)")


def image_transform(image_size, is_train, mean=None, std=None, **_unused):
    """reference open_clip/transform.py:274 (same leading arguments)."""
    return GpuImageTransform(image_size, mean or OPENAI_DATASET_MEAN, std or OPENAI_DATASET_STD, is_train)
