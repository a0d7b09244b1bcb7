"""Golden fixtures for the data formats either side of the TTA path, made by RUNNING THE REFERENCE's own data code here.

Run from the repo root:  python tests/golden/make_golden_data.py

Imports (never copies) /root/reference/src/data/utils.py and coco_negated_dataset.py by file path under a synthetic
package (the reference's data/__init__.py pulls in torchvision and decord, which are absent and not needed), feeds them
the committed synthetic CSV tests/golden/coco_negated_synth.csv, and stores OUTPUTS only in tests/golden/data.json:
  * pre_caption on a list of awkward strings / lists,
  * every attribute the TTA loop and the online evaluation read from the two datasets.
Cannot run on the GPU box (no /root/reference there).
"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/src/data"

CAPTION_CASES = [
    "A photo of a Dog, but NOT a cat.",
    "  Two   people (smiling) ; no  umbrella!  ",
    "line one\nline two\n",
    "tabs\tand nbsp ~ tilde * star # hash : colon \"quoted\"",
    "",
    "...",
    " ".join(f"w{i}" for i in range(80)),
    "ends with newline\n\n",
    "UPPER.lower.MiXeD",
    ["a list", "Of Strings.", "joined!"],
    ["x"] * 90,
    "trailing spaces and dots . . .   ",
    "multiple\n\nnewlines  inside",
    "unicode éè 中文 caption; with (parens)",
]


def load_reference_data_modules():
    pkg = types.ModuleType("data")
    pkg.__path__ = [REF_DATA]
    sys.modules["data"] = pkg
    tvd = types.ModuleType("torchvision.datasets.utils")
    tvd.download_url = lambda *a, **k: None
    for name in ("torchvision", "torchvision.datasets"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.datasets.utils"] = tvd
    mods = {}
    for name in ("utils", "coco_negated_dataset"):
        spec = importlib.util.spec_from_file_location(f"data.{name}", os.path.join(REF_DATA, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"data.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def main():
    mods = load_reference_data_modules()
    pre_caption = mods["utils"].pre_caption
    out = {"pre_caption": [dict(input=c, max_words=mw, output=pre_caption(c, mw))
                           for c in CAPTION_CASES for mw in (50, 77, 5)]}
    csv_path = os.path.join(HERE, "coco_negated_synth.csv")
    ds = mods["coco_negated_dataset"]
    text = ds.coco_negated_retrieval_eval_text(None, csv_file=csv_path, sep=',', img_key='filepath',
                                               caption_key_pos='pos_captions', caption_key_neg='neg_captions')
    image = ds.coco_negated_retrieval_eval_image(None, csv_file=csv_path, sep=',', img_key='filepath',
                                                 caption_key_pos='pos_captions', caption_key_neg='neg_captions')
    out["text_dataset"] = dict(len=len(text), text=text.text, pos_text=text.pos_text, neg_text=text.neg_text,
                               inv_text=text.inv_text, image=text.image,
                               txt2img={str(k): v for k, v in text.txt2img.items()},
                               img2txt={str(k): v for k, v in text.img2txt.items()},
                               items=[list(text[i]) for i in range(len(text))])
    out["image_dataset"] = dict(len=len(image), text=image.text, image=image.image,
                                txt2img={str(k): v for k, v in image.txt2img.items()},
                                img2txt={str(k): v for k, v in image.img2txt.items()})
    with open(os.path.join(HERE, "data.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=True)
    print("wrote data.json:", len(out["pre_caption"]), "pre_caption cases,", len(text), "caption rows,", len(image), "images")


if __name__ == "__main__":
    main()
