"""Golden fixtures for the MSR-VTT side of the data path, made by RUNNING THE REFERENCE's own modules here.

Run from the repo root:  python tests/golden/make_golden_video.py

Imports (never copies) /root/reference/src/data/{utils,utils_video,frame_sampler,video_reader,msrvtt_negated_dataset}.py by
file path under a synthetic package.  Third-party modules those files import but never use on this path are absent from
the image and are given empty placeholders (``overrides`` -- a no-op decorator, ``decord``, torchvision's ``VideoClips`` /
``download_url``); none of them takes part in the values stored.  OUTPUTS only go to tests/golden/video.json:
  * UniformFrameSampler(max_frames)(start, end, fps) for a grid of clip lengths,
  * every attribute the TTA loop reads from the two msrvtt_negated datasets on tests/golden/coco_negated_synth.csv
    (same CSV schema; the file paths are never opened by these attributes).
Cannot run on the GPU box (no /root/reference there).
"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/src/data"


def load_reference_modules():
    import torch  # noqa: F401  (the sampler is torch code)
    pkg = types.ModuleType("data")
    pkg.__path__ = [REF_DATA]
    sys.modules["data"] = pkg
    ov = types.ModuleType("overrides")
    ov.overrides = lambda f: f
    sys.modules["overrides"] = ov
    dec = types.ModuleType("decord")
    dec.bridge = types.SimpleNamespace(set_bridge=lambda *_: None)
    dec.DECORDError = RuntimeError
    sys.modules["decord"] = dec
    for name in ("torchvision", "torchvision.datasets", "torchvision.datasets.utils", "torchvision.datasets.video_utils",
                 "torchvision.datasets.folder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision.datasets"].folder = sys.modules["torchvision.datasets.folder"]
    sys.modules["torchvision.datasets.utils"].download_url = lambda *a, **k: None
    sys.modules["torchvision.datasets.video_utils"].VideoClips = object
    mods = {}
    for name in ("utils", "utils_video", "frame_sampler", "video_reader", "msrvtt_negated_dataset"):
        spec = importlib.util.spec_from_file_location(f"data.{name}", os.path.join(REF_DATA, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"data.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def main():
    mods = load_reference_modules()
    fs = mods["frame_sampler"]
    cases = []
    for max_frames in (1, 2, 4, 8):
        sampler = fs.UniformFrameSampler(max_frames=max_frames)
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 30, 100, 299, 300, 301, 1000, 4097):
            for start in (0, 3):
                idx = [int(v) for v in sampler(start, start + n - 1, 25.0)]
                cases.append(dict(max_frames=max_frames, start=start, end=start + n - 1, indices=idx))
    out = {"uniform_frame_sampler": cases}
    csv_path = os.path.join(HERE, "coco_negated_synth.csv")
    ds = mods["msrvtt_negated_dataset"]
    text = ds.msrvtt_negated_retrieval_eval_text(None, csv_file=csv_path, sep=',', img_key='filepath',
                                                 caption_key_pos='pos_captions', caption_key_neg='neg_captions')
    image = ds.msrvtt_negated_retrieval_eval_image(None, csv_file=csv_path, sep=',', img_key='filepath',
                                                   caption_key_pos='pos_captions', caption_key_neg='neg_captions')
    out["text_dataset"] = dict(len=len(text), text=text.text, pos_text=text.pos_text, neg_text=text.neg_text,
                               inv_text=text.inv_text, image=text.image,
                               txt2img={str(k): v for k, v in text.txt2img.items()},
                               img2txt={str(k): v for k, v in text.img2txt.items()},
                               items=[list(text[i]) for i in range(len(text))])
    out["image_dataset"] = dict(len=len(image), text=image.text, image=image.image,
                                txt2img={str(k): v for k, v in image.txt2img.items()},
                                img2txt={str(k): v for k, v in image.img2txt.items()},
                                max_frames=image.frame_sampler.max_frames)
    with open(os.path.join(HERE, "video.json"), "w") as f:
        json.dump(out, f, indent=None, ensure_ascii=True)
    print("wrote video.json:", len(cases), "sampler cases,", len(text), "caption rows,", len(image), "videos")


if __name__ == "__main__":
    main()
