"""LayerNorm kernels at the step's sizes (ViT-B-32 text, B = 256): us per launch and algorithmic HBM rate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
C = int(os.environ.get("C", 512))
def med(fn, n=20):
    ts = []
    for r in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
Mf, Mb = 4 * 256 * 77, 3 * 256 * 77
x = torch.randn(Mf, C, generator=g, device=dev)
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
y, _, mu, rs = ops.ln_fwd(x, gam, bet)
t = med(lambda: ops.ln_fwd(x, gam, bet))
print(f"ln_fwd {Mf} x {C}: {t:6.1f} us  {Mf*C*6/t/1e6:5.2f} TB/s", flush=True)
dy = torch.randn(Mb, C, generator=g, device=dev).bfloat16()
dres = torch.randn(Mb, C, generator=g, device=dev)
part = torch.empty(ops.ln_bwd_blocks(Mb), 2, C, device=dev)
t = med(lambda: ops.ln_bwd(dy, x[:Mb], gam, mu[:Mb], rs[:Mb], dres=dres, inplace_res=True, partial=part))
print(f"ln_bwd {Mb} x {C}: {t:6.1f} us  {Mb*C*16/t/1e6:5.2f} TB/s", flush=True)
