"""TEST INFRASTRUCTURE (oracle): CPU restatement of the image preprocessing of the reference's TTA datasets.

Reference: src/data/__init__.py:14-20 -- torchvision ``Resize((S, S), BICUBIC)`` on a PIL image (= Pillow's
``Image.resize``: two-pass separable convolution on 8-bit channels with antialiasing support scaling and 22-bit
fixed-point coefficients, libImaging/Resample.c), ``ToTensor`` (uint8 / 255) and ``Normalize(mean, std)``.
The arithmetic lives in Pillow, a third-party dependency of the reference (present in this image, 12.2.0);
tests/test_oracle_image.py pins this restatement against Pillow itself, bit for bit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the whole input range.
    Returns (ksize, bounds int32 [out, 2] = (first input index, count), kk int32 [out, ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0                                                    # plain left-to-right sum, as in C
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img: np.ndarray, out_size: int) -> np.ndarray:
    """One separable pass along axis 0 of img (uint8 [n, ...]) -> uint8 [out_size, ...]."""
    _, bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img: np.ndarray, size: int) -> np.ndarray:
    """img uint8 [H, W, 3] -> uint8 [size, size, 3]; Pillow order: horizontal pass, then vertical pass
    (a pass whose input and output sizes are equal is skipped, as ImagingResample does)."""
    h, w, _ = img.shape
    tmp = img
    if w != size:
        tmp = _pass(np.ascontiguousarray(img.transpose(1, 0, 2)), size).transpose(1, 0, 2)
    if h != size:
        tmp = _pass(np.ascontiguousarray(tmp), size)
    return np.ascontiguousarray(tmp)


def to_tensor_normalize(img_u8: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """uint8 [S, S, 3] -> float32 [3, S, S]: ToTensor (x / 255 in fp32) then Normalize ((x - mean) / std in fp32)."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return ((x - m) / s).astype(np.float32)


def preprocess(img_u8: np.ndarray, size: int = 224) -> np.ndarray:
    return to_tensor_normalize(resize_bicubic_u8(img_u8, size))
