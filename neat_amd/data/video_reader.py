"""Clip readers for the MSR-VTT gallery (reference src/data/video_reader.py:17-83: ``VideoReader.from_path(path)``,
``len(reader)``, ``reader(indices) -> uint8 [n, H, W, 3]``, ``get_avg_fps()``).

Decoding stays on the host and is not part of the built path.  Readers here:
  * ``NpyVideoReader``     -- a ``.npy`` file holding the decoded clip, uint8 [T, H, W, 3] (memory-mapped);
  * ``ImageVideoReader``   -- a still image = a clip of one frame (PIL);
  * ``DecordVideoReader``  -- any container ``decord`` can open, when decord is installed (it is not in this image;
                              like the reference, an unreadable file yields zero frames of 256 x 256).
"""
import os
from typing import Sequence

import numpy as np
import torch

_IMAGE_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


class VideoReader:
    def __call__(self, indices: Sequence[int]) -> torch.Tensor:
        raise NotImplementedError

    def __len__(self) -> int:
        raise NotImplementedError

    def get_avg_fps(self) -> float:
        raise NotImplementedError

    @staticmethod
    def from_path(path) -> "VideoReader":
        low = os.fspath(path).lower()
        if low.endswith('.npy'):
            return NpyVideoReader(path)
        if low.endswith(_IMAGE_EXT):
            return ImageVideoReader(path)
        return DecordVideoReader(path)


class NpyVideoReader(VideoReader):
    def __init__(self, path, fps: float = 1.0) -> None:
        self.frames = np.load(path, mmap_mode='r', allow_pickle=False)
        if self.frames.dtype != np.uint8 or self.frames.ndim != 4 or self.frames.shape[3] != 3:
            raise RuntimeError(f"{path}: expected a uint8 [T, H, W, 3] clip, got {self.frames.dtype} {self.frames.shape}")
        self.fps = fps

    def __call__(self, indices):
        return torch.from_numpy(np.stack([self.frames[int(i)] for i in indices]))

    def __len__(self):
        return int(self.frames.shape[0])

    def get_avg_fps(self):
        return self.fps


class ImageVideoReader(VideoReader):
    def __init__(self, path) -> None:
        from .coco_negated_dataset import decode_rgb_u8
        self.frame = decode_rgb_u8(path)

    def __call__(self, indices):
        return self.frame.unsqueeze(0).expand(len(indices), -1, -1, -1).contiguous()

    def __len__(self):
        return 1

    def get_avg_fps(self):
        return 1.0


class DecordVideoReader(VideoReader):
    def __init__(self, path) -> None:
        try:
            import decord
        except ImportError as e:
            raise RuntimeError(f"{path}: decoding video containers needs 'decord', which is not installed; decode the "
                               "clip to a uint8 [T, H, W, 3] .npy file instead") from e
        try:
            self.video_reader = decord.VideoReader(os.fspath(path), num_threads=1)
        except decord.DECORDError:
            self.video_reader = None

    def __call__(self, indices):
        if self.video_reader is not None:
            return torch.from_numpy(self.video_reader.get_batch(list(indices)).asnumpy())
        return torch.zeros(len(indices), 256, 256, 3, dtype=torch.uint8)

    def __len__(self):
        return len(self.video_reader) if self.video_reader is not None else 1

    def get_avg_fps(self):
        return self.video_reader.get_avg_fps() if self.video_reader is not None else 1
