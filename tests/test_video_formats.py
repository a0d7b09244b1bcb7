"""CPU tests of the MSR-VTT side of the data path.

* UniformFrameSampler and the two msrvtt_negated datasets against tests/golden/video.json, which holds the outputs of the
  REFERENCE's own modules on the same inputs (tests/golden/make_golden_video.py);
* the video-transform oracle (oracle/video_oracle.py) against torch's own CPU operator -- the third-party code the
  reference's tensor Resize(antialias=True) calls.  Float arithmetic: tolerance 2e-6 on the [0, 1] scale (ATen sums the
  taps in its own vector order), 1e-5 after Normalize (division by std ~ 0.27).
"""
import json
import os

import numpy as np
import pytest
import torch

from neat_amd.data.frame_sampler import UniformFrameSampler
from neat_amd.data.msrvtt_negated_dataset import msrvtt_negated_retrieval_eval_image, msrvtt_negated_retrieval_eval_text
from neat_amd.data.video_reader import NpyVideoReader, VideoReader
from oracle import video_oracle as VO

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "video.json")))
CSV = os.path.join(HERE, "golden", "coco_negated_synth.csv")
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def test_uniform_frame_sampler_matches_reference():
    assert len(GOLD["uniform_frame_sampler"]) >= 100
    for c in GOLD["uniform_frame_sampler"]:
        got = UniformFrameSampler(c["max_frames"])(c["start"], c["end"], 25.0)
        assert got == c["indices"], c
        assert all(type(v) is int for v in got)


def test_msrvtt_text_dataset_matches_reference():
    g = GOLD["text_dataset"]
    ds = msrvtt_negated_retrieval_eval_text(None, csv_file=CSV)
    assert len(ds) == g["len"]
    for k in ("text", "pos_text", "neg_text", "inv_text", "image"):
        assert getattr(ds, k) == g[k], k
    assert {str(k): v for k, v in ds.txt2img.items()} == g["txt2img"]
    assert {str(k): v for k, v in ds.img2txt.items()} == g["img2txt"]
    assert [list(ds[i]) for i in range(len(ds))] == g["items"]


def test_msrvtt_image_dataset_matches_reference():
    g = GOLD["image_dataset"]
    ds = msrvtt_negated_retrieval_eval_image(None, csv_file=CSV)
    assert len(ds) == g["len"] and ds.text == g["text"] and ds.image == g["image"]
    assert {str(k): v for k, v in ds.txt2img.items()} == g["txt2img"]
    assert {str(k): v for k, v in ds.img2txt.items()} == g["img2txt"]
    assert ds.frame_sampler.max_frames == g["max_frames"] == 4


def test_msrvtt_image_dataset_samples_frames_from_npy(tmp_path):
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 256, (30, 12, 16, 3), dtype=np.uint8)
    np.save(tmp_path / "v0.npy", clip)
    np.save(tmp_path / "v1.npy", clip[:3])
    csv_file = tmp_path / "v.csv"
    row = '"[\'a man is not singing\']","[\'a man\']","[\'singing\']","[\'a man is singing\']"'
    csv_file.write_text("filepath,captions,pos_captions,neg_captions,inverted_captions\n"
                        f"{tmp_path / 'v0.npy'},{row}\n{tmp_path / 'v1.npy'},{row}\n")
    ds = msrvtt_negated_retrieval_eval_image(None, csv_file=str(csv_file))
    frames, idx = ds[0]
    assert idx == 0 and frames.dtype == torch.uint8 and tuple(frames.shape) == (4, 12, 16, 3)
    want = UniformFrameSampler(4)(0, 29)
    assert np.array_equal(frames.numpy(), clip[want])
    frames, _ = ds[1]                                             # 3 frames, integer ticks 0 0 1 2: midpoints 0, 0, 2
    assert UniformFrameSampler(4)(0, 2) == [0, 0, 2] and np.array_equal(frames.numpy(), clip[[0, 0, 2]])
    assert isinstance(VideoReader.from_path(tmp_path / "v0.npy"), NpyVideoReader)


def test_video_reader_rejects_containers_without_decord(tmp_path):
    try:
        import decord  # noqa: F401
        pytest.skip("decord is installed")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="decord"):
        VideoReader.from_path(tmp_path / "clip.mp4")


@pytest.mark.parametrize("h,w,s", [(240, 320, 224), (120, 160, 64), (224, 224, 224), (50, 70, 96), (224, 300, 224),
                                   (448, 448, 224), (37, 53, 32)])
def test_video_oracle_matches_torch_operator(h, w, s):
    rng = np.random.default_rng(h + w)
    fr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    fr[0, : h // 4] = 255                                         # saturated band: the float result overshoots, unclamped
    fr[0, h // 4: h // 2, ::2] = 0
    x = torch.from_numpy(fr).permute(0, 3, 1, 2).to(torch.float32) / 255          # ConvertImageDtype on the permuted view
    ref = torch.nn.functional.interpolate(x, size=(s, s), mode="bicubic", align_corners=False, antialias=True)
    got = VO.resize_bicubic_aa(np.transpose(fr, (0, 3, 1, 2)).astype(np.float32) / np.float32(255), s)
    assert np.abs(got - ref.numpy()).max() <= 2e-6
    if (h, w) != (s, s):
        assert ref.max() > 1.0 or ref.min() < 0.0                 # overshoot really is kept
    m = torch.tensor(MEAN).view(1, 3, 1, 1)
    d = torch.tensor(STD).view(1, 3, 1, 1)
    assert np.abs(VO.preprocess_frames(fr, s, MEAN, STD) - ((ref - m) / d).numpy()).max() <= 1e-5


def test_video_oracle_coefficients_sum_to_one_and_identity():
    for n_in, n_out in [(320, 224), (240, 224), (100, 224), (224, 224), (1280, 224)]:
        xmin, xsize, w = VO.aa_coeffs(n_in, n_out)
        assert np.all(xmin >= 0) and np.all(xmin + xsize <= n_in) and np.all(xsize >= 1)
        assert np.abs(w.sum(1) - 1).max() < 1e-6
    x = np.random.default_rng(1).random((1, 3, 24, 24), dtype=np.float32)
    assert np.array_equal(VO.resize_bicubic_aa(x, 24), x)
