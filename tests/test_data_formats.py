"""CPU tests of the data formats either side of the TTA path.

* pre_caption and the two coco_negated datasets against tests/golden/data.json, which holds the outputs of the
  REFERENCE's own data code on the same inputs (tests/golden/make_golden_data.py);
* the image oracle (oracle/image_oracle.py) against Pillow itself, bit for bit -- Pillow is the third-party library
  whose Image.resize the reference's transform calls.
"""
import json
import os

import numpy as np
import pytest

from neat_amd.data.coco_negated_dataset import coco_negated_retrieval_eval_image, coco_negated_retrieval_eval_text
from neat_amd.data.utils import pre_caption
from oracle import image_oracle as IO

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "data.json")))
CSV = os.path.join(HERE, "golden", "coco_negated_synth.csv")


def test_pre_caption_matches_reference():
    assert len(GOLD["pre_caption"]) >= 40
    for case in GOLD["pre_caption"]:
        assert pre_caption(case["input"], case["max_words"]) == case["output"], case


def test_pre_caption_passes_other_types_through():
    assert pre_caption(None) is None and pre_caption(3.5) == 3.5      # the reference returns them unchanged


def test_text_dataset_matches_reference():
    g = GOLD["text_dataset"]
    ds = coco_negated_retrieval_eval_text(None, csv_file=CSV)
    assert len(ds) == g["len"]
    for k in ("text", "pos_text", "neg_text", "inv_text", "image"):
        assert getattr(ds, k) == g[k], k
    assert {str(k): v for k, v in ds.txt2img.items()} == g["txt2img"]
    assert {str(k): v for k, v in ds.img2txt.items()} == g["img2txt"]
    assert [list(ds[i]) for i in range(len(ds))] == g["items"]


def test_image_dataset_matches_reference():
    g = GOLD["image_dataset"]
    ds = coco_negated_retrieval_eval_image(None, csv_file=CSV)
    assert len(ds) == g["len"] and ds.text == g["text"] and ds.image == g["image"]
    assert {str(k): v for k, v in ds.txt2img.items()} == g["txt2img"]
    assert {str(k): v for k, v in ds.img2txt.items()} == g["img2txt"]


def test_image_dataset_decodes_to_u8_hwc(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (31, 45, 3), dtype=np.uint8)
    p = tmp_path / "im.png"
    Image.fromarray(a).save(p)
    csv_file = tmp_path / "one.csv"
    csv_file.write_text('filepath,captions,pos_captions,neg_captions,inverted_captions\n'
                        f'{p},"[\'a dog\']","[\'dog\']","[\'cat\']","[\'a cat\']"\n')
    ds = coco_negated_retrieval_eval_image(None, csv_file=str(csv_file))
    img, idx = ds[0]
    assert idx == 0 and img.dtype.is_floating_point is False and tuple(img.shape) == (31, 45, 3)
    assert np.array_equal(img.numpy(), a)


SIZES = [(37, 53, 24), (480, 640, 224), (224, 224, 224), (100, 300, 224), (500, 375, 224), (17, 19, 224),
         (224, 500, 224), (1, 1, 8), (640, 427, 384)]


@pytest.mark.parametrize("h,w,s", SIZES)
def test_image_oracle_equals_pillow(h, w, s):
    from PIL import Image
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if h > 30:
        a[: h // 3] = 255                                   # saturated regions: overshoot must clip, not wrap
        a[h // 3: h // 2, ::2] = 0
    ref = np.asarray(Image.fromarray(a).resize((s, s), Image.BICUBIC))
    assert np.array_equal(IO.resize_bicubic_u8(a, s), ref)


def test_normalize_oracle_equals_torch():
    import torch
    rng = np.random.default_rng(1)
    u = rng.integers(0, 256, (24, 24, 3), dtype=np.uint8)
    t = torch.from_numpy(u).permute(2, 0, 1).to(torch.float32).div(255)          # ToTensor
    t = t.sub_(torch.tensor(IO.CLIP_MEAN).view(3, 1, 1)).div_(torch.tensor(IO.CLIP_STD).view(3, 1, 1))   # Normalize
    assert np.array_equal(IO.to_tensor_normalize(u), t.numpy())


def test_tokenizer_matches_reference():
    """Token ids of our BPE implementation == the reference tokenizer's (tests/golden/tokenizer.json).  Needs the
    external merge table; skipped where it is not installed (the GPU box has no reference checkout)."""
    from neat_amd.open_clip import tokenizer as T
    try:
        path = T.default_bpe()
    except FileNotFoundError:
        pytest.skip("CLIP BPE merge table not available")
    g = json.load(open(os.path.join(HERE, "golden", "tokenizer.json")))
    tok = T.SimpleTokenizer(path)
    assert (tok.vocab_size, tok.sot_token_id, tok.eot_token_id) == (g["vocab_size"], g["sot"], g["eot"])
    for c in g["cases"]:
        assert tok.encode(c["text"]) == c["ids"], c["text"]
        assert tok(c["text"])[0].tolist() == c["ctx77"]
        assert tok(c["text"], 16)[0].tolist() == c["ctx16"]
        assert tok.decode(c["ids"]) == c["decoded"]
    batch = tok([c["text"] for c in g["cases"]])
    assert batch.shape == (len(g["cases"]), 77) and batch.dtype.is_floating_point is False
