"""End-to-end throughput of train_one_epoch (strings in, scores out) against the device-only figure of bench.py: does the
host side (tokenising 4 x 256 captions per batch, copies, logging) hide under the GPU time of the step?

No BPE table travels to the GPU box, so the tokenizer here is a stand-in with the same structure and host cost as
neat_amd.open_clip.SimpleTokenizer (regex pre-tokenisation, per-word cache, one int64 array per call); its cost per
batch is printed next to the epoch's time per batch."""
import os, random, re, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import open_clip as oc, synth, train_TTA as T
from neat_amd.optim import AdamW

DEV = torch.device("cuda", 0)
NB, B, NIMG = int(os.environ.get("NB", 24)), 256, 1000
cfg = synth.get_cfg("ViT-B-32")
L, V = cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"]
WORDS = ("a photo of the dog cat but not with and no person riding bicycle street umbrella table pizza there is without "
         "any people standing near water giraffe").split()
PAT = re.compile(r"[a-z]+|[0-9]|[^\sa-z0-9]+")
_ids = {}


# the real SimpleTokenizer measures 5.4 ms per 4 x 256 captions on this host class since round 3 (word cache: 16.7-22 ms in
# round 2, 88 ms before that); TOK_MS pads every call of the stand-in to that cost (per caption set) with a busy wait
TOK_MS = float(os.environ.get("TOK_MS", 1.4))


def tokenizer(texts):
    t_in = time.perf_counter()
    out = _tokenize(texts)
    while (time.perf_counter() - t_in) * 1e3 < TOK_MS:
        pass
    return out


def _tokenize(texts):
    out = np.zeros((len(texts), L), dtype=np.int64)
    for i, t in enumerate(texts):
        ids = [V - 2]
        for w in PAT.findall(re.sub(r"\s+", " ", t).strip().lower()):
            hit = _ids.get(w)
            if hit is None:
                hit = _ids[w] = [1 + (sum(map(ord, w)) * 7 + len(w)) % (V - 3)]
            ids.extend(hit)
        ids = ids[:L - 1] + [V - 1]
        out[i, :len(ids)] = ids
    return torch.from_numpy(out)


class Loader:
    def __init__(self, batches, n, attr):
        self.batches, self.dataset = batches, types.SimpleNamespace(**{attr: list(range(n))})

    def __iter__(self):
        return iter(self.batches)


random.seed(0)
cap = lambda: " ".join(random.choice(WORDS) for _ in range(random.randint(6, 14))) + "."
text_batches = [([cap() for _ in range(B)], [cap() for _ in range(B)], [cap() for _ in range(B)], [cap() for _ in range(B)],
                 torch.arange(i * B, (i + 1) * B)) for i in range(NB)]
g = torch.Generator().manual_seed(3)
image_batches = [(torch.randn(250, 3, 224, 224, generator=g), torch.arange(i * 250, (i + 1) * 250)) for i in range(NIMG // 250)]

t0 = time.perf_counter()
for tb in text_batches[:4]:
    for x in tb[:4]:
        tokenizer(x)
print(f"tokenizer stand-in: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per batch of 4 x {B} captions (host)", flush=True)

model = oc.create_model("ViT-B-32", device=DEV)
model.load_state_dict(synth.synth_state_dict("ViT-B-32", 0))
if os.environ.get("PACK") is not None:
    model.pack_tokens = os.environ["PACK"] != "0"
print(f"pack_tokens = {model.pack_tokens}", flush=True)
oc.freeze_parameters(model, False)
opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
args = types.SimpleNamespace(distributed=False, tta_only_visual=False, name="neat", tta_steps=3, tta_temperature=0.03,
                             tta_temperature2=0.07)
T.train_one_epoch(model, tokenizer, Loader(text_batches[:2], 2 * B, "text"), Loader(image_batches, NIMG, "image"), opt, DEV, args)
torch.cuda.synchronize()
t0 = time.perf_counter()
T.train_one_epoch(model, tokenizer, Loader(text_batches, NB * B, "text"), Loader(image_batches[:1], NIMG, "image"), opt, DEV, args)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"train_one_epoch: {NB} batches of {B} captions x 3 steps in {dt:.3f} s (incl. one gallery batch) = "
      f"{dt / NB * 1e3:.1f} ms per batch, {NB * B / dt:.0f} pairs/s end to end", flush=True)
