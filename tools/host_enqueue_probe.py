"""How much of a TTA step is host time?  Enqueues adaptation batches without synchronising and reports the host's
enqueue time next to the wall time of the same batches (GPU drained at the end), full-length and packed text batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, open_clip as oc, ops, synth
from neat_amd import train_TTA as T
from neat_amd.optim import AdamW

dev = torch.device("cuda", 0)
name = os.environ.get("MODEL", "ViT-B-32")
cfg = synth.get_cfg(name)
t = cfg["text_cfg"]
model = oc.create_model(name, device=dev)
model.load_state_dict(synth.synth_state_dict(name, 0))
oc.freeze_parameters(model, False)
opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
B, nb = 256, 6
G = torch.nn.functional.normalize(torch.randn(5000, cfg["embed_dim"], device=dev), dim=-1)
batches = [[s.to(dev) for s in synth.synth_caption_sets(B, t["context_length"], t["vocab_size"], 1000 + i)] for i in range(nb)]


def run(i):
    plan = T.batch_plan(model, batches[i])
    for _ in range(3):
        T.tta_step(model, opt, batches[i], G, 0.03, 0.07, plan=plan)


for packed in (False, True):
    model.pack_tokens = packed
    run(0); run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nb):
        run(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"packed={packed}: host enqueue {(t1 - t0) / nb * 1e3:.2f} ms per batch, wall {(t2 - t0) / nb * 1e3:.2f} ms per batch "
          f"({(t2 - t0) / nb / 3 * 1e3:.2f} ms per step)", flush=True)

if os.environ.get("PROFILE"):
    # where the host time of the packed step goes (cProfile inflates everything ~2x; read the ranking, not the numbers)
    import cProfile, pstats
    model.pack_tokens = True
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(nb):
        run(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
