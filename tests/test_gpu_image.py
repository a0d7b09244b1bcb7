"""GPU parity of neat_image_preprocess (through the C ABI) against the oracle and against Pillow: bit-exact."""
import numpy as np
import pytest
import torch

from neat_amd import ops
from oracle import image_oracle as IO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SIZES = [(37, 53, 24), (480, 640, 224), (224, 224, 224), (100, 300, 224), (500, 375, 224), (17, 19, 224),
         (224, 500, 224), (300, 224, 224), (1, 1, 8), (640, 427, 384), (2000, 3000, 224)]


@pytest.mark.parametrize("h,w,s", SIZES)
def test_image_preprocess_bit_exact(h, w, s):
    from PIL import Image
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if h > 30:
        a[: h // 3] = 255
        a[h // 3: h // 2, ::2] = 0
    out = ops.image_preprocess(torch.from_numpy(a).to(DEV), s).cpu().numpy()
    resized = np.asarray(Image.fromarray(a).resize((s, s), Image.BICUBIC))       # the library the reference calls
    if h * w <= 640 * 480:                                                       # the numpy oracle is slow on huge inputs
        assert np.array_equal(IO.resize_bicubic_u8(a, s), resized)
    assert np.array_equal(out, IO.to_tensor_normalize(resized))


def test_gpu_loader_matches_host_transform(tmp_path):
    """create_dataset / create_loader end to end: CSV -> decoded images -> GPU preprocessing == Pillow + torch ops."""
    from PIL import Image
    from neat_amd import data as D
    rng = np.random.default_rng(9)
    rows, arrays = [], []
    for i, (h, w) in enumerate([(48, 64), (100, 80), (224, 224), (30, 300), (333, 500)]):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = tmp_path / f"{i}.png"
        Image.fromarray(a).save(p)
        arrays.append(a)
        rows.append(f'{p},"[\'a dog but not a cat\']","[\'a dog\']","[\'a cat\']","[\'a cat but not a dog\']"')
    csv_file = tmp_path / "t.csv"
    csv_file.write_text("filepath,captions,pos_captions,neg_captions,inverted_captions\n" + "\n".join(rows) + "\n")
    text_ds, image_ds = D.create_dataset("coco_negated", str(csv_file), 224)
    assert len(text_ds) == 5 and len(image_ds) == 5
    text_loader, image_loader = D.create_loader([text_ds, image_ds], [None, None], [4, 2], [0, 0], [False, False],
                                                [None, None], device=DEV)
    seen = {}
    for batch, index in image_loader:
        assert batch.is_cuda and batch.dtype == torch.float32 and tuple(batch.shape[1:]) == (3, 224, 224)
        for b, i in zip(batch.cpu().numpy(), index.tolist()):
            seen[i] = b
    assert sorted(seen) == list(range(5))
    for i, a in enumerate(arrays):
        ref = IO.to_tensor_normalize(np.asarray(Image.fromarray(a).resize((224, 224), Image.BICUBIC)))
        assert np.array_equal(seen[i], ref)
    n = 0
    for cap, pos, neg, inv, idx in text_loader:
        n += len(idx)
        assert all(isinstance(c, str) for c in cap)
    assert n == 5
