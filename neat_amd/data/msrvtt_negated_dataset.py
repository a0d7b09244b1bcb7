"""TTA datasets over a NegBench MSR-VTT-negated CSV (reference src/data/msrvtt_negated_dataset.py:18-160).

Same CSV schema and bookkeeping as the COCO-negated datasets (``filepath`` is a video path).  The gallery dataset samples
``max_frames`` (4) frames uniformly (``UniformFrameSampler`` over frames 0 .. len-1) and hands out the DECODED frames as a
uint8 [T, H, W, 3] tensor; the transform (/ 255, antialiased bicubic resize, Normalize) runs on the GPU in the loader
(neat_amd.data.GpuVideoLoader -> neat_frames_preprocess), not in the dataset.
"""
import torch

from .coco_negated_dataset import coco_negated_retrieval_eval_image, coco_negated_retrieval_eval_text
from .frame_sampler import UniformFrameSampler
from .video_reader import VideoReader


class msrvtt_negated_retrieval_eval_image(coco_negated_retrieval_eval_image):
    def __init__(self, transform, csv_file, sep=',', img_key='filepath', caption_key='captions',
                 caption_key_pos='pos_captions', caption_key_neg='neg_captions', max_words=77, max_frames=4):
        super().__init__(transform, csv_file, sep=sep, img_key=img_key, caption_key=caption_key,
                         caption_key_pos=caption_key_pos, caption_key_neg=caption_key_neg, max_words=max_words)
        self.frame_sampler = UniformFrameSampler(max_frames=max_frames)

    def _load_and_sample_video(self, video_path) -> torch.Tensor:
        reader = VideoReader.from_path(video_path)
        indices = self.frame_sampler(0, len(reader) - 1, reader.get_avg_fps())
        return reader(indices)

    def __getitem__(self, index):
        frames = self._load_and_sample_video(self.annotation[index]['image'])       # uint8 [T, H, W, 3]
        if self.transform is not None:
            frames = self.transform(frames.permute(0, 3, 1, 2))
        return frames, index


class msrvtt_negated_retrieval_eval_text(coco_negated_retrieval_eval_text):
    def __init__(self, transform, csv_file, sep=',', img_key='filepath', caption_key='captions',
                 caption_key_pos='pos_captions', caption_key_neg='neg_captions', caption_key_inv='inverted_captions',
                 max_words=77):
        super().__init__(transform, csv_file, sep=sep, img_key=img_key, caption_key=caption_key,
                         caption_key_pos=caption_key_pos, caption_key_neg=caption_key_neg,
                         caption_key_inv=caption_key_inv, max_words=max_words)
