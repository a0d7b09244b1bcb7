"""Per-batch wall time of the TTA step from a cold process on a fresh box (one synchronisation per batch): how long the
first batches of a process run slow, which decides what the bench's set-up must prime."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import open_clip as oc, synth, train_TTA as T
from neat_amd.optim import AdamW

dev = torch.device("cuda", 0)
model = oc.create_model("ViT-B-32", device=dev)
model.load_state_dict(synth.synth_state_dict("ViT-B-32", 0))
oc.freeze_parameters(model, False)
opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
cfg = synth.get_cfg("ViT-B-32")["text_cfg"]
G = torch.nn.functional.normalize(torch.randn(5000, 512, device=dev), dim=-1)
if "--gallery" in sys.argv:                     # bench.py's Phase A first: 5 000 images through the vision tower
    g = torch.Generator(device=dev).manual_seed(1234)
    t0 = time.perf_counter()
    with torch.no_grad():
        for s0 in range(0, 5000, 250):
            G[s0:s0 + 250] = model.encode_image(torch.randn(250, 3, 224, 224, generator=g, device=dev), normalize=True)
    torch.cuda.synchronize()
    print(f"batch -1 gallery encode {time.perf_counter() - t0:.3f} s", flush=True)
if "--score" in sys.argv:
    score = torch.full((256 * 8, 5000), -100.0, device=dev)
batches = [[s.to(dev) for s in synth.synth_caption_sets(256, 77, cfg["vocab_size"], 1000 + i)] for i in range(8)]
torch.cuda.synchronize()
t00 = time.perf_counter()
for i in range(24):
    t0 = time.perf_counter()
    for _ in range(3):
        T.tta_step(model, opt, batches[i % 8], G, 0.03, 0.07)
    torch.cuda.synchronize()
    print(f"batch {i:2d} at {time.perf_counter() - t00:6.2f} s: {(time.perf_counter() - t0) * 1e3:7.2f} ms", flush=True)
