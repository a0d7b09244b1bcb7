"""Vendor-library yardstick: torch.mm (hipBLASLt / rocBLAS) on the eight GEMM shapes of one text layer, plain bf16 out.
Not part of the product path; prints us per launch."""
import os
import torch
dev = "cuda"
M4, M3 = int(os.environ.get("M4", 78848)), int(os.environ.get("M3", 59136))     # forward / backward rows (packed: 14848 / 11136)
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", M4, 1536, 512), ("out", M4, 512, 512), ("fc", M4, 2048, 512), ("proj", M4, 512, 2048),
          ("dproj", M3, 2048, 512), ("dfc", M3, 512, 2048), ("do", M3, 512, 512), ("dqkv", M3, 512, 1536)]
tot = 0.0
for name, M, N, K in shapes:
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for r in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.mm(A, W.t(), out=out)
        b.record()
        torch.cuda.synchronize()
        if r >= 4:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    tot += med
    print(f"{name:6s} M={M} N={N} K={K}  med {med:7.1f} us  min {ts[0]:7.1f}  {2.0*M*N*K/med/1e6:7.1f} TF/s", flush=True)
print(f"sum {tot:.1f} us")
