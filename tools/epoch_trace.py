"""Where the host blocks inside one train_one_epoch iteration: host time of every phase of the loop body, no added syncs."""
import os, sys, time, types
sys.argv = [sys.argv[0]]
os.environ.setdefault("NB", "12")
import importlib.util
spec = importlib.util.spec_from_file_location("eb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "epoch_bench.py"))
eb = importlib.util.module_from_spec(spec); spec.loader.exec_module(eb)
import torch
from neat_amd import train_TTA as T
dev = eb.DEV
G = torch.nn.functional.normalize(torch.randn(1000, 512, device=dev), dim=-1)
score = torch.full((eb.NB * eb.B, 1000), -100.0, device=dev)
pending = None
for bi, (text, pos, neg, inv, idx) in enumerate(eb.text_batches):
    t = [time.perf_counter()]
    toks = [eb.tokenizer(x) for x in (text, pos, neg, inv)]; t.append(time.perf_counter())
    st = torch.stack(toks); t.append(time.perf_counter())
    pn = st.pin_memory(); t.append(time.perf_counter())
    d = pn.to(dev, non_blocking=True); t.append(time.perf_counter())
    local = [x.contiguous() for x in d.unbind(0)]; t.append(time.perf_counter())
    for _ in range(3):
        losses, _, sims = T.tta_step(eb.model, eb.opt, local, G, 0.03, 0.07, 1, 0, global_rows=eb.B)
    t.append(time.perf_counter())
    idd = idx.pin_memory().to(dev, non_blocking=True); t.append(time.perf_counter())
    score[idd] = sims; t.append(time.perf_counter())
    if pending is not None:
        pending.tolist()
    pending = losses; t.append(time.perf_counter())
    names = ["tokenize", "stack", "pin", "h2d", "slices", "3 steps (enqueue)", "idx h2d", "index_put", "log prev"]
    print(f"batch {bi:2d}: " + ", ".join(f"{n} {1e3 * (b - a):.1f}" for n, a, b in zip(names, t, t[1:])), flush=True)
