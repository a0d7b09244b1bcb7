"""Coarse alignment + TTA loss (forward and analytic gradient) at config 2 (B = Bl = 256, D = 512, gallery 5 000) and at
config 3's per-rank shape (global B = 2 048, Bl = 256 rows of this rank, D = 768): us per call, all launches of the call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def nrm(*shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=g, device=dev), dim=-1)
def med(fn, n=20):
    ts = []
    for r in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for name, B, Bl, D, N in (("config 2: B=256 (one GPU)", 256, 256, 512, 5000), ("config 3: global B=2048, this rank's 256 rows", 2048, 256, 768, 5000)):
    G = nrm(N, D)
    t_all, c_all = nrm(B, D), nrm(B, D)
    row0 = B - Bl
    t, p, v, q = t_all[row0:row0 + Bl].contiguous(), nrm(Bl, D), nrm(Bl, D), nrm(Bl, D)
    tc = med(lambda: ops.coarse_align(t, p, q, G))
    tl = med(lambda: ops.tta_loss_fwd_bwd(t, p, v, t_all, c_all, 0.03, 0.07, row0=row0))
    print(f"{name}: coarse_align {tc:7.1f} us ({6.0*Bl*N*D/tc/1e6:6.1f} TFLOP/s fp32)   loss fwd+bwd {tl:7.1f} us   sum {tc+tl:7.1f} us", flush=True)
