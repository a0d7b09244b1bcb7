"""Attention kernels at the step's size (ViT-B-32 text, B = 256: 1024 sequences forward, 768 backward, 8 heads, L = 77):
us per launch and algorithmic HBM rate (fwd: qkv in + out; bwd: qkv + dO in, dqkv out)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops

dev = "cuda"
L, heads, C = 77, 8, 512
SCALE = 1
if len(sys.argv) > 1 and sys.argv[1] == "contig":     # upper bound of a [sequence][head][token][q|k|v] layout: one head per "sequence"
    heads, C, SCALE = 1, 64, 8
g = torch.Generator(device=dev).manual_seed(0)
def med(fn, n=20):
    ts = []
    for r in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
for nf, nb in ((1024 * SCALE, 768 * SCALE),):
    qkv = (torch.randn(nf * L, 3 * C, generator=g, device=dev) * 0.5).bfloat16()
    do = (torch.randn(nb * L, C, generator=g, device=dev) * 0.1).bfloat16()
    o, lse = ops.attn_fwd(qkv, nf, L, heads, True)
    t, tmin = med(lambda: ops.attn_fwd(qkv, nf, L, heads, True))
    by = nf * L * C * 2 * 4
    print(f"attn fwd  {nf} seq: med {t:7.1f} us  min {tmin:7.1f}  {by/t/1e6:5.2f} TB/s of 8 = {by/t/8e6:.3f}", flush=True)
    t, tmin = med(lambda: ops.attn_bwd(qkv, do, lse, nb, L, heads, True))
    by = nb * L * C * 2 * 8
    print(f"attn bwd  {nb} seq: med {t:7.1f} us  min {tmin:7.1f}  {by/t/1e6:5.2f} TB/s of 8 = {by/t/8e6:.3f}", flush=True)
