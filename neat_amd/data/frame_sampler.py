"""Frame index choice for a clip (reference src/data/frame_sampler.py:31-40, the sampler the MSR-VTT datasets use).

``UniformFrameSampler(max_frames)(start, end, fps)``: cut [start, end] into n = min(max_frames, end - start + 1) intervals
with integer ticks (``torch.linspace(..., dtype=torch.int)``: fp32 steps, truncated) and take the midpoint of every
interval, halves rounded to even (``torch.round``).  Returns plain ints.
"""
from typing import List

import torch


class UniformFrameSampler:
    def __init__(self, max_frames: int) -> None:
        self.max_frames = max_frames

    def __call__(self, start_frame: int, end_frame: int, fps: float = 0.0) -> List[int]:
        n = min(self.max_frames, end_frame - start_frame + 1)
        ticks = torch.linspace(start_frame, end_frame, n + 1, dtype=torch.int).tolist()
        # (a + b) / 2 is exact in binary floating point; Python's round() is round-half-to-even like torch.round
        return [int(round((a + b) / 2)) for a, b in zip(ticks[:-1], ticks[1:])]
