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


def test_image_preprocess_batch_equals_single_calls():
    """Mixed sizes in one launch set (up- and down-scaling, one-axis passes, H == W == S) == per-image calls, bit for bit."""
    rng = np.random.default_rng(2)
    shapes = [(480, 640), (224, 224), (100, 300), (224, 500), (300, 224), (17, 19), (640, 427), (1, 1), (500, 375)]
    ims = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).to(DEV) for h, w in shapes]
    for s in (224, 32):
        got = ops.image_preprocess_batch(ims, s)
        for i, im in enumerate(ims):
            assert torch.equal(got[i], ops.image_preprocess(im, s)), (shapes[i], s)


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


def test_csv_to_scores_flow_equals_pretokenized_run(tmp_path):
    """The whole drop-in flow -- CSV -> datasets -> loaders (GPU preprocessing) -> tokenizer -> train_one_epoch ->
    online recall -- gives bit-identical scores to the same epoch fed with pre-tokenised tensors and host-preprocessed
    (Pillow + torch-op) images: the data path adds nothing but the reference's own transforms."""
    import types
    from PIL import Image
    from torch.utils.data import SequentialSampler
    from neat_amd import data as D, open_clip as oc, synth, train_TTA as T
    from neat_amd.evaluation import retrieval as R
    from neat_amd.optim import AdamW

    cfg = synth.get_cfg("neat-tiny")
    ctx, vocab, S = cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"], cfg["vision_cfg"]["image_size"]
    rng = np.random.default_rng(4)
    words = ["dog", "cat", "kite", "bench", "pizza", "zebra", "clock", "boat"]
    rows, arrays = [], []
    for i in range(6):
        a = rng.integers(0, 256, (int(rng.integers(20, 90)), int(rng.integers(20, 90)), 3), dtype=np.uint8)
        p = tmp_path / f"{i}.png"
        Image.fromarray(a).save(p)
        arrays.append(a)
        caps, pos, neg, inv = [], [], [], []
        for j in range(3):
            x, y = words[(i + j) % 8], words[(i + 3 * j + 1) % 8]
            caps.append(f"A photo of a {x}, but not a {y}.")
            pos.append(f"a {x}")
            neg.append(f"a {y}")
            inv.append(f"A photo of a {y}, but not a {x}.")
        rows.append(f'{p},"{caps!r}","{pos!r}","{neg!r}","{inv!r}"')
    csv_file = tmp_path / "t.csv"
    csv_file.write_text("filepath,captions,pos_captions,neg_captions,inverted_captions\n" + "\n".join(rows) + "\n")

    def toy_tokenizer(texts):                                   # the real BPE table is an external asset (not on this box)
        out = torch.zeros(len(texts), ctx, dtype=torch.long)
        for r, t in enumerate(texts):
            ids = [vocab - 2] + [1 + (sum(map(ord, w)) * 7 + len(w)) % (vocab - 3) for w in t.split(" ")][: ctx - 2] + [vocab - 1]
            out[r, :len(ids)] = torch.tensor(ids)
        return out

    def run(text_loader, image_loader, tokenizer):
        model = oc.create_model("neat-tiny", device=DEV)
        model.load_state_dict(synth.synth_state_dict("neat-tiny", 3), strict=True)
        oc.freeze_parameters(model, False)
        opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
        args = types.SimpleNamespace(distributed=False, tta_only_visual=False, name="neat", tta_steps=2,
                                     tta_temperature=0.03, tta_temperature2=0.07)
        return T.train_one_epoch(model, tokenizer, text_loader, image_loader, opt, torch.device(DEV), args)[1]

    text_ds, image_ds = D.create_dataset("coco_negated", str(csv_file), S)
    loaders = D.create_loader([text_ds, image_ds], [SequentialSampler(text_ds), SequentialSampler(image_ds)], [8, 4],
                              [0, 0], [False, False], [None, None], device=DEV)
    t2i = run(loaders[0], loaders[1], toy_tokenizer)

    class Fake:
        def __init__(self, batches, n, attr):
            self.batches, self.dataset = batches, types.SimpleNamespace(**{attr: list(range(n))})

        def __iter__(self):
            return iter(self.batches)

    imgs = torch.stack([torch.from_numpy(IO.to_tensor_normalize(np.asarray(Image.fromarray(a).resize((S, S), Image.BICUBIC))))
                        for a in arrays])
    ib = [(imgs[i:i + 4], torch.arange(i, min(i + 4, 6))) for i in range(0, 6, 4)]
    n = len(text_ds)
    tb = []
    for i in range(0, n, 8):
        sl = slice(i, min(i + 8, n))
        tb.append((toy_tokenizer(text_ds.text[sl]), toy_tokenizer(text_ds.pos_text[sl]), toy_tokenizer(text_ds.neg_text[sl]),
                   toy_tokenizer(text_ds.inv_text[sl]), torch.arange(sl.start, sl.stop)))
    ref = run(Fake(tb, n, "text"), Fake(ib, 6, "image"), lambda x: x)
    assert torch.equal(t2i, ref) and (t2i > -100).all()

    # online evaluation on the produced score matrix: caption r belongs to image txt2img[r]
    texts_image_index = [text_ds.txt2img[r][0] for r in range(n)]
    args = types.SimpleNamespace(device=DEV, batch_size=4, video=False)
    metrics = R.evaluate_model_with_similarity(loaders[1], args, t2i, recall_k_list=(1, 5),
                                               texts_image_index=texts_image_index)
    assert any("recall" in k or "R@" in k for k in metrics) and all(0.0 <= float(v) <= 1e6 for v in metrics.values())
