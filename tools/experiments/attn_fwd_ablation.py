"""Ablations of attn_fwd3_kernel (tools/experiments/attn_fwd_ablation.patch applied to a copy of attn.hip, built into a second library, NEAT_HIP_LIB=that library): bit 4 = staging only (no
compute, no store), bit 5 = no staging (compute + store on whatever the LDS holds)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neat_amd import lib, ops
dev = "cuda"
L, heads, C, nf = 77, 8, 512, 1024
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(nf * L, 3 * C, generator=g, device=dev) * 0.5).bfloat16()
out = torch.empty(nf * L, C, dtype=torch.bfloat16, device=dev)
lse = torch.empty(nf, heads, L, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(flags):
    lib.call("neat_attn_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), None, nf, L, heads, flags, st)
def med(fn, n=20):
    ts = []
    for r in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
cases = [("full", 1)]
for amt in (1, 2, 4, 8):
    cases.append((f"stagger by dispatch wave, {amt} x 1 us", 1 | 0x100 | (amt << 10)))
for amt in (2, 4):
    cases.append((f"stagger by blockIdx & 3, {amt} x 1 us", 1 | 0x200 | (amt << 10)))
cases.append(("full", 1))
for name, fl in cases:
    print(f"{name:40s} {med(lambda: run(fl)):7.1f} us", flush=True)
